#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_cql_gpu.py tests/test_dist_gpu.py tests/test_ops.py -x -q -m gpu > gpurun_out/r02/cql3_tests.log 2>&1
python bench.py --workload cql --steps 1000 --warmup 50 > gpurun_out/r02/cql3_bench.json 2> gpurun_out/r02/cql3_bench.err
python bench.py --workload cql > gpurun_out/r02/cql3_bench_default.json 2>> gpurun_out/r02/cql3_bench.err
python - > gpurun_out/r02/cql3_ab.log 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from porl_amd import engine as E
from porl_amd.train.cql_trainer import CQLTrainer
from porl_amd.util.synth import make_discrete_transitions
dev = torch.device("cuda", 0)
S, A, B, N = 60, 10, 4096, 100_000
for two in (1, 0):
    E.tune_set("qnet_two_groups", two)
    torch.manual_seed(0)
    t = CQLTrainer(state_size=S, action_size=A, gamma=0.99, device=dev, batch_size=B)
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=0)
    rb = t.replay_buffer
    rb.states[:N], rb.actions[:N], rb.rewards[:N], rb.next_states[:N], rb.dones[:N] = st, ac, rw, ns, dn
    rb.size, rb.position = N, 0
    t.async_losses = True
    for _ in range(50): t.learn_device_sampled()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): t.learn_device_sampled()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("two_groups=%d: %.0f updates/s (%.1f us)" % (two, 1000 / el, 1e3 * el))
E.tune_set("qnet_two_groups", 1)
PY
