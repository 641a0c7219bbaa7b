#!/usr/bin/env python3
"""Headline benchmark: POR gradient-steps/sec (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full POR update (agent/por.py:73-112 of the reference) INCLUDING minibatch
acquisition: draw B distinct rows of the device-resident replay shard, gather them, run value step +
EMA + policy step.  Workload at N=1 is BASELINE config 2 (S=60, A=2, H=1024, L=2, B=1024, 1 M-row
buffer, fp32); at N>1 every rank keeps its own 1.25 M-row shard (N=8 -> the 10 M-row buffer of config 4)
and draws B=1024 local rows — weak scaling, gradients all-reduced with RCCL.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : dominant kernel (the 128x128 fp32-MFMA grouped GEMM) timed with HIP events on its
                 launch stream in a second, instrumented pass over the same K steps
  cpu_baseline : the numpy oracle (oracle/por_oracle.py, BLAS threads stated) timed on the host cores
                 on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E
S, A, H, L, B = 60, 2, 1024, 2, 1024


def por_flops_per_sample():
    """14 440 448 MAC per sample."""
    mac_v = S * H + H * H + H
    mac_p = S * H + H * H + H * S
    fwd = 6 * mac_v + mac_p
    bwd = 2 * (mac_v + H * H + H) + (mac_p + H * H + H * S)
    return 2 * (fwd + bwd)


def cpu_baseline(budget_s=12.0):
    """CPU steps/s on the host, same shapes, bounded by wall time: the stronger of (a) eager PyTorch-CPU (MKL,
    autograd, torch.optim.Adam — how the reference itself runs; oracle/torch_cpu.py) and (b) the numpy oracle
    (oracle/por_oracle.py), each with its thread count tuned by a few trial steps so that oversubscription does not
    handicap it.  Both are this repository's restatements ("port"); the reference's files never travel to this box."""
    import numpy as np
    import torch
    from oracle.por_oracle import PorOracle
    from oracle.torch_cpu import PorTorchCpu
    from porl_amd.util.init import build_por_state_dict
    from porl_amd.util.synth import make_rows, split_rows
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        max_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threadpool_limits, max_threads = None, os.cpu_count() or 1
    sd = build_por_state_dict(S, H, L, seed=0)
    rows = make_rows(8 * B, S, A, seed=0)
    batches = [split_rows(rows[k * B:(k + 1) * B], S, A)[:4] for k in range(8)]

    def run(step, budget):
        step(0)                                                  # warm-up (thread pools, page faults)
        n, t0 = 0, time.perf_counter()
        while True:
            step(n)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget and n >= 3:
                return n, el

    # (a) eager PyTorch on the CPU
    tb = [tuple(torch.from_numpy(np.ascontiguousarray(x)) for x in (s, sp, r, d)) for s, r, sp, d in batches]
    ncpu = os.cpu_count() or 1
    best_tt, best_rate = torch.get_num_threads(), 0.0
    for tt in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(tt)
        m = PorTorchCpu(sd, S, H, L)
        n, el = run(lambda i: m.update(*tb[i % 8]), 0.6)
        if n / el > best_rate:
            best_tt, best_rate = tt, n / el
    torch.set_num_threads(best_tt)
    m = PorTorchCpu(sd, S, H, L)
    nt, elt = run(lambda i: m.update(*tb[i % 8]), budget_s / 2)
    cand = [dict(value=nt / elt, cores=int(best_tt), impl="eager PyTorch-CPU (oracle/torch_cpu.py)", n=nt, el=elt)]
    # (b) numpy oracle
    o = PorOracle(build_por_state_dict(S, H, L, seed=0), S, H, L)
    nb = [(s, sp, r, d) for s, r, sp, d in batches]
    best_t, best_rate = max_threads, 0.0
    if threadpool_limits is not None:
        for t in sorted({t for t in (8, 16, 32, 64, max_threads) if t <= max_threads}):
            with threadpool_limits(limits=t):
                n, el = run(lambda i: o.por_residual_update(*nb[i % 8]), 0.4)
            if n / el > best_rate:
                best_t, best_rate = t, n / el
    ctx = threadpool_limits(limits=best_t) if threadpool_limits is not None else None
    try:
        nn_, eln = run(lambda i: o.por_residual_update(*nb[i % 8]), budget_s / 2)
    finally:
        if ctx is not None and hasattr(ctx, "restore_original_limits"):
            ctx.restore_original_limits()
    cand.append(dict(value=nn_ / eln, cores=int(best_t), impl="numpy oracle (oracle/por_oracle.py)", n=nn_, el=eln))
    best = max(cand, key=lambda c: c["value"])
    other = min(cand, key=lambda c: c["value"])
    return dict(value=best["value"], unit="gradient-steps/sec", cores=best["cores"], kind="port",
                sample=f"{best['n']} POR updates (B={B}, H={H}, S={S}) of {best['impl']}, {best['cores']} threads (tuned), "
                       f"in {best['el']:.1f} s; the other port, {other['impl']} on {other['cores']} threads: "
                       f"{other['value']:.1f} steps/s")


def bench_cql(a):
    """Secondary workload (BASELINE config 3): CQL(H) learn() at B=4096, S=60, A=10, Q-net 64-128-64, 100 k-row
    buffer resident on the device, indices drawn on the device.  Not the headline metric."""
    import numpy as np
    import torch
    from porl_amd.train.cql_trainer import CQLTrainer
    from porl_amd.util.synth import make_discrete_transitions
    from oracle.por_oracle import CqlOracle
    dev = torch.device("cuda", 0)
    Sq, Aq, Bq, Nq = 60, 10, 4096, 100_000
    torch.manual_seed(0)
    t = CQLTrainer(state_size=Sq, action_size=Aq, gamma=0.99, device=dev, batch_size=Bq)
    st, ac, rw, ns, dn = make_discrete_transitions(Nq, Sq, Aq, seed=0)
    rb = t.replay_buffer
    rb.states[:Nq], rb.actions[:Nq], rb.rewards[:Nq], rb.next_states[:Nq], rb.dones[:Nq] = st, ac, rw, ns, dn
    rb.size, rb.position = Nq, 0
    t.async_losses = True
    for i in range(a.warmup):
        t.learn_device_sampled()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        t.learn_device_sampled()
        if i % 10 == 0:
            t.sync_target()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # instrumented second pass: HIP events around the step kernel on its launch stream
    from porl_amd import engine as E
    E.prof_enable(True)
    for i in range(a.steps):
        t.learn_device_sampled()
    prof = E.prof_read()
    E.prof_enable(False)
    roof = None
    dom = [p for p in prof if p["name"] in ("qnet_fused_kernel", "qnet_resident_kernel") and p["launches"]]
    if dom:
        # the step is HBM/latency-bound; algorithmic bytes = gathered rows + parameters/Adam state
        n_par = sum(p.numel() for p in t.q_network.parameters())
        alg_bytes = Bq * (2 * Sq + 3) * 4 + Bq * 8 + n_par * 28
        avg_us = 1e3 * dom[0]["total_ms"] / dom[0]["launches"]
        ach = alg_bytes / (avg_us * 1e-6) / 1e9
        roof = dict(bound="hbm", kernel=dom[0]["name"], achieved=ach, peak=8000.0, unit="GB/s", frac=ach / 8000.0,
                    traffic=None, avg_launch_us=avg_us, launches=dom[0]["launches"], algorithmic_bytes_per_launch=alg_bytes,
                    # the other roof (138 368 MAC per sample = 3 forward + 2 backward passes of the
                    # 20 864-MAC network): neither binds — the step is a chain of dependent 32-row layer stages
                    mfma_tflops=2.0 * 138368 * Bq / (avg_us * 1e-6) / 1e12 if (Sq, Aq) == (60, 10) else None,
                    mfma_frac=(2.0 * 138368 * Bq / (avg_us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if (Sq, Aq) == (60, 10) else None,
                    note="latency-bound: 2.6 MB of compulsory traffic per step, one block per 32 rows walks every layer",
                    all_kernels_us_per_step={p["name"]: 1e3 * p["total_ms"] / a.steps for p in prof if p["launches"]})
    # CPU baseline: eager PyTorch-CPU restatement (oracle/torch_cpu.py: autograd + torch.optim.Adam, like the reference
    # runs) incl. numpy's O(N) sampling as buffer/replay_buffer.py:64 does it, threads tuned; the numpy oracle beside it
    from oracle.torch_cpu import CqlTorchCpu
    qsd = {k: v.detach().cpu().numpy() for k, v in t.q_network.state_dict().items()}
    rng = np.random.default_rng(0)
    tst, tac, trw, tns, tdn = (torch.from_numpy(x) for x in (st, ac, rw, ns, dn))

    def torch_rate(budget, threads):
        torch.set_num_threads(threads)
        m = CqlTorchCpu(qsd, Aq)
        n, c0 = 0, time.perf_counter()
        while time.perf_counter() - c0 < budget:
            idx = torch.from_numpy(rng.choice(Nq, Bq, replace=False))
            m.learn(tst[idx], tac[idx], trw[idx], tns[idx], tdn[idx])
            n += 1
        return n / (time.perf_counter() - c0), n

    ncpu = os.cpu_count() or 1
    best_tt = max(sorted({x for x in (1, 4, 8, 16, 32) if x <= ncpu}), key=lambda x: torch_rate(0.5, x)[0])
    cpu, n = torch_rate(4.0, best_tt)
    o = CqlOracle(qsd, Aq)
    n2, c0 = 0, time.perf_counter()
    while time.perf_counter() - c0 < 2.0:
        idx = rng.choice(Nq, Bq, replace=False)
        o.learn(st[idx], ac[idx], rw[idx], ns[idx], dn[idx])
        n2 += 1
    cpu_np = n2 / (time.perf_counter() - c0)
    out = {"metric": "gradient-steps/sec (CQL learn, batch=4096)", "value": a.steps / el,
           "unit": "gradient-steps/sec", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "CQL S=60 A=10 B=4096 Q-net 64-128-64, 100k-row device-resident buffer, indices drawn "
                                  "and rows gathered on the device"},
           "cpu_baseline": {"value": cpu, "unit": "gradient-steps/sec", "kind": "port", "cores": int(best_tt),
                            "sample": f"{n} learn() steps of the eager PyTorch-CPU restatement (oracle/torch_cpu.py) incl. numpy "
                                      f"sampling, {best_tt} threads (tuned) in 4 s; numpy oracle: {cpu_np:.1f} steps/s"}}
    if roof:
        out["roofline"] = roof
    print(json.dumps(out), flush=True)


def bench_sorl_enc(a):
    """Secondary workload (BASELINE config 5): SORL.update with the FasterNet costmap encoder as backbone, B=512,
    S=362 (360 beams + goal), feature_dim=256, H=512, fp32 (the parity bar is 1e-5, so no bf16).  One step =
    encode s, encode s' (train-mode BatchNorm, DropPath), value + policy update.  Reports the per-kernel time
    split of one step from an instrumented second pass."""
    import numpy as np
    import torch
    from types import SimpleNamespace
    from porl_amd.agent.fasternet import FasterNet
    from porl_amd.agent.sorl import SORL
    from porl_amd import engine as E
    dev = torch.device("cuda", 0)
    Bq, F, Hq, Aq = a.batch or 512, 256, 512, 2
    torch.manual_seed(0)
    n_ang, n_dist = a.angle_bins, a.dist_bins             # 360 x 256 = the reference's image; 84 x 84 = BASELINE's wording
    backbone = FasterNet(3, F, max_batch=Bq, angle_bins=n_ang, dist_bins=n_dist, compute_dtype=a.enc_dtype)
    args = SimpleNamespace(state_size=n_ang + 2, feature_dim=F, hidden_dim=Hq, n_hidden=2, layer_norm=False, action_size=Aq,
                           max_batch=Bq)
    agent = SORL(args, max_steps=1000, tau=0.9, alpha=3.0, device=dev, backbone=backbone)
    agent.async_losses = True
    rng = np.random.default_rng(0)
    nb = 4
    st = np.empty((nb, 2, Bq, n_ang + 2), dtype=np.float32)
    st[..., :n_ang] = rng.uniform(0.15, 3.9, size=(nb, 2, Bq, n_ang))
    st[..., n_ang:] = rng.uniform(-3, 3, size=(nb, 2, Bq, 2))
    st = torch.from_numpy(st).to(dev)
    act = torch.from_numpy(rng.uniform(-1, 1, size=(nb, Bq, Aq)).astype(np.float32)).to(dev)
    rew = torch.from_numpy(rng.normal(size=(nb, Bq)).astype(np.float32)).to(dev)
    done = torch.from_numpy((rng.uniform(size=(nb, Bq)) < 0.1).astype(np.float32)).to(dev)

    def one_step(i):
        k = i % nb
        agent.update(st[k, 0], act[k], rew[k], st[k, 1], done[k])

    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    losses = agent._engine.stats[:2].cpu().numpy()
    if not np.isfinite(losses).all():
        raise SystemExit("non-finite loss in the benchmark run")
    E.prof_enable(True)
    for i in range(a.steps):
        one_step(i)
    prof = E.prof_read()
    E.prof_enable(False)
    split = {p["name"]: p["total_ms"] / a.steps for p in prof if p["launches"]}
    gemms = [p for p in prof if p["name"].startswith(("gemm_f32_kernel", "gemm_bf16_kernel")) and p["launches"]]
    dom = max(gemms, key=lambda p: p["total_ms"])
    ach = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
    hbm = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9
    enc_flops = 2 * 2 * Bq * 0.86e9 * (n_ang * n_dist) / (360.0 * 256.0)
    out = {"metric": "gradient-steps/sec (SORL update + FasterNet encoder, batch=512)", "value": a.steps / el,
           "unit": "gradient-steps/sec", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if a.enc_dtype == "fp32" else "bf16 operands / f32 accumulate (encoder GEMMs), f32 elsewhere",
           "data": "synthetic",
           "config": {"workload": f"SORL S={n_ang + 2} F={F} H={Hq} B={Bq} + FasterNet(3,{F}) encoder on {n_ang}x{n_dist} costmaps, "
                                  "2 encoder forwards (train-mode BN, DropPath) + value/policy update per step"},
           "algorithmic_tflops": enc_flops * a.steps / el / 1e12,
           "roofline": (dict(bound="mfma", kernel=dom["name"], achieved=ach, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                             frac=ach / PEAK_FP32_MFMA_TFLOPS) if a.enc_dtype == "fp32" else
                        dict(bound="hbm", kernel=dom["name"], achieved=hbm, peak=8000.0, unit="GB/s", frac=hbm / 8000.0,
                             algorithmic_bytes_note="rows x (K + N [+ N residual]) x 4 B per product: operands and "
                                                    "results stay fp32 in HBM, only the multiply is bf16")) |
                       dict(traffic=None, launches=dom["launches"],
                            avg_launch_us=1e3 * dom["total_ms"] / dom["launches"], all_kernels_ms_per_step=split)}
    if not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import fasternet_oracle as FO
        sd = {k: v.cpu().numpy() for k, v in backbone.state_dict().items()}
        stats = {k: v.copy() for k, v in sd.items() if "running" in k}
        bs = 8
        x = st[0, 0, :bs].cpu().numpy()
        c0 = time.perf_counter()
        FO.forward(sd, stats, x.copy(), True, np.ones((3, bs), np.float32), dtype=np.float32, angle_bins=n_ang, dist_bins=n_dist)
        FO.forward(sd, stats, x.copy(), True, np.ones((3, bs), np.float32), dtype=np.float32, angle_bins=n_ang, dist_bins=n_dist)
        dt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": (bs / Bq) / dt, "unit": "gradient-steps/sec", "kind": "port", "cores": os.cpu_count(),
                               "sample": f"2 encoder forwards of oracle/fasternet_oracle.py (numpy fp32) on {bs} of the "
                                         f"{Bq} samples in {dt:.1f} s, scaled by {bs}/{Bq}; heads' update excluded (<1 %)"}
    print(json.dumps(out), flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without torchrun: start N rank processes (one GPU each) from a parent that makes
    no HIP call at all — children are fresh interpreters (subprocess, no fork of GPU state, no re-exec) — relay
    rank 0's JSON line and exit with the worst child code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):    # a dead rank would leave the others in a collective
            for p in procs:
                if p.poll() is None:
                    p.kill()                                  # exactly the PIDs started above
        time.sleep(0.2)
    reader.join(timeout=10)
    codes = [p.returncode for p in procs]
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"rank exit codes {codes}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="por", choices=["por", "cql", "sorl_enc"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows-per-gpu", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--hidden", type=int, default=0,
                    help="por: hidden width other than the headline's 1024 (e.g. 256, the reference's class default, "
                         "value_functions.py:32); with --batch for the small-network figures of DESIGN.md §7")
    ap.add_argument("--angle-bins", type=int, default=360, help="sorl_enc: costmap rows (360 = the reference's image)")
    ap.add_argument("--dist-bins", type=int, default=256, help="sorl_enc: costmap columns")
    ap.add_argument("--enc-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="sorl_enc: operand type of the encoder's 1x1 / merge convolutions (fp32 = reference parity)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="issue every update back to back on one stream (profiling: kernels never overlap)")
    a = ap.parse_args()
    if a.workload == "cql":
        return bench_cql(a)
    if a.workload == "sorl_enc":
        return bench_sorl_enc(a)
    global H, B
    if a.hidden:
        H = a.hidden                        # not the headline configuration: config.workload says so
    if a.batch:
        B = a.batch
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)          # bare `python bench.py --gpus N`: this process never touches a GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    from types import SimpleNamespace

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # PORL_BENCH_BACKEND=gloo rehearses the N>1 code path with several ranks on ONE GPU (no RCCL, no timing claim)
    backend = os.environ.get("PORL_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from porl_amd.agent.por import POR
    from porl_amd.buffer.replay_buffer import PackedReplay
    from porl_amd.util.synth import make_rows
    from porl_amd import engine as E

    rows_per_gpu = a.rows_per_gpu or (1_000_000 if world == 1 else 1_250_000)
    # every rank generates only its own shard (same generator family, rank-keyed seed)
    shard = make_rows(rows_per_gpu, S, A, seed=1000 + rank)
    replay = PackedReplay(shard, S, A, dev, rank=0, world=1, seed=rank)
    del shard

    args = SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=A, max_batch=B)
    torch.manual_seed(0)                                       # identical replicas on every rank
    agent = POR(args, max_steps=1000, tau=0.9, alpha=10.0, device=dev)
    agent.async_losses = True                                  # no host sync inside the loop
    agent.pipeline = not a.no_pipeline
    if os.environ.get("PORL_IQL_FOLD") == "0":                 # A/B: slab combines as separate launches
        E.tune_set("iql_fold", 0)
    if os.environ.get("PORL_L0_KERNEL") == "0":                # A/B: input layers through the grouped GEMM
        E.tune_set("l0_kernel", 0)
    for i, v in enumerate(os.environ.get("PORL_TILE_MAP", "").split(",")):      # A/B: e.g. "0,1,2,3" = round-1 choice
        if v.strip():
            E.tune_set("tile_map%d" % i, int(v))
            E.tune_set("tile_map_short%d" % i, int(v))
    if os.environ.get("PORL_IQL_PAD"):                         # A/B: "value,policy[,min_blocks]" LDS pads of the pipelined update
        v = [int(x) for x in os.environ["PORL_IQL_PAD"].split(",")]
        E.tune_set("iql_pad_value", v[0]); E.tune_set("iql_pad_policy", v[1])
        if len(v) > 2:
            E.tune_set("iql_pad_min_blocks", v[2])
        if len(v) > 3:
            E.tune_set("iql_pad_min_k", v[3])
    if os.environ.get("PORL_GEMM_LDS_PAD"):                    # A/B: fewer co-resident GEMM blocks per CU (placement knob)
        E.tune_set("gemm_lds_pad", int(os.environ["PORL_GEMM_LDS_PAD"]))
    if os.environ.get("PORL_VBWD_TILE"):                       # A/B: tile of the value backward in pipelined mode
        E.tune_set("vbwd_tile_short", int(os.environ["PORL_VBWD_TILE"]))
    if os.environ.get("PORL_L0_TILE"):                         # A/B: tile of the K = 60 forward layers
        E.tune_set("l0_tile", int(os.environ["PORL_L0_TILE"]))
    losses = torch.zeros(a.steps + a.warmup, 8, device=dev)    # device-side loss history, one row per update

    def one_step(i):
        # draw B distinct rows of the resident shard + gather + split (one kernel), then the update;
        # the three loss statistics of update i land in losses[i] without copies
        agent._engine.set_stats(losses[i])
        agent.update_from_replay(replay, B)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, not warm-up: ~50 ms of fp32-MFMA work on scratch tensors so that a short run (the driver times 20 steps
    # after 5 warm-up steps: 9 ms in all) does not measure the card's clock / power ramp from idle.  It touches no
    # agent state; the W warm-up updates and the K timed updates below are exactly as asked.  PORL_BENCH_SPINUP_MS=0
    # turns it off.
    spin_ms = float(os.environ.get("PORL_BENCH_SPINUP_MS", "50"))
    if spin_ms > 0:
        sa, sb = torch.randn(4096, 1024, device=dev), torch.randn(1024, 1024, device=dev)
        sc = torch.empty(4096, 1024, device=dev)
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < spin_ms:
            for _ in range(20):
                E.gemm_f32(0, sa, sb, 4096, 1024, 1024, 1024, 1024, sc, 1024)
            torch.cuda.synchronize()
        del sa, sb, sc

    for i in range(a.warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step(a.warmup + i)
    agent.flush()                                              # a deferred policy step belongs to the timed work
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if world > 1:
        # async mode keeps per-rank SHARES of the batch means in the history: one reduction after the timed
        # loop makes them the global-batch statistics (sum of shares; minimum of the per-rank minima)
        sums = losses[:, 0:2].contiguous()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        losses[:, 0:2] = sums
        mins = losses[:, 2].contiguous()
        dist.all_reduce(mins, op=dist.ReduceOp.MIN)
        losses[:, 2] = mins
    lh = losses[:, :3].cpu().numpy()
    if not np.isfinite(lh).all():
        raise SystemExit("non-finite loss in the benchmark run")

    # Not `value`: the same loop again for 1 000 updates.  After any idle the card needs ~10 ms of THIS workload to
    # settle its clocks (scripts/bench_ramp.py: the first ~20 updates after a sync run 10 % slower, whatever GEMM or
    # streaming spin-up precedes them), and a K = 20 run also pays the pipeline's fill and drain once; a training job
    # runs millions of updates, so the sustained rate is reported beside the contract's K-step figure.
    sustained = None
    if world == 1 and a.steps < 1000 and os.environ.get("PORL_BENCH_SUSTAINED", "1") != "0":
        n_s = 1000
        barrier()
        ts = time.perf_counter()
        for i in range(n_s):
            one_step(a.warmup + i % a.steps)
        agent.flush()
        barrier()
        sustained = n_s / (time.perf_counter() - ts)

    roof = None
    if not a.no_roofline:
        # second pass, instrumented: HIP events around every kernel launch on the launch stream.  The timed loop above
        # overlaps the policy phase with the next value phase on two streams; a kernel's roofline is quoted with the
        # kernel ALONE on the chip, so this pass runs the same updates back to back on one stream.
        agent.flush()
        agent.pipeline = False
        psteps = max(a.steps, 100)                 # enough launches for stable per-kernel averages whatever K is
        E.prof_enable(True)
        for i in range(psteps):
            one_step(a.warmup + i % a.steps)
        prof = E.prof_read()
        E.prof_enable(False)
        agent.pipeline = not a.no_pipeline
        # profile labels are "<launch of the step>:<kernel>".  The roofline is quoted for the DOMINANT LAUNCH of the
        # update (the labelled launch with the largest total time: the 4-net hidden-layer forward) — one kernel
        # instantiation can serve launches of very different shapes, which a per-instantiation average would mix
        gemms = [p for p in prof if p["launches"] and "gemm_f32_kernel" in p["name"]]
        if gemms:
            dom = max(gemms, key=lambda p: p["total_ms"])
            name = dom["name"].split(":")[-1]
            avg_ms = dom["total_ms"] / dom["launches"]
            flops_per_launch = dom["flops"] / dom["launches"]
            ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
            # HBM bytes per launch are NOT measured by this run: they come from the committed rocprofv3 --pmc passes
            # over this same command (separate passes, as the microarch guide prescribes); the source file is named
            traffic, traffic_src = None, None
            for fname in ("r02_hbm_traffic.json", "r01_hbm_traffic.json"):
                try:
                    tj = json.load(open(os.path.join(REPO, "profiles", fname)))
                    traffic = tj["kernels"][name]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/" + fname
                    break
                except Exception:
                    pass
            roof = dict(bound="mfma", kernel=name, launch=dom["name"], achieved=ach, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=ach / PEAK_FP32_MFMA_TFLOPS, traffic=traffic, traffic_source=traffic_src,
                        avg_launch_us=avg_ms * 1e3, launches=dom["launches"],
                        flop_per_launch=flops_per_launch,
                        launches_per_step=sum(p["launches"] for p in prof) / psteps,
                        instrumented_pass="%d updates back to back on one stream (no overlap), HIP events around "
                                          "every launch (~2.5 us of event overhead inside each figure)" % psteps,
                        # every launch of one update, HIP-event timed on the launch stream (instrumented pass)
                        step_launches_us={p["name"]: round(1e3 * p["total_ms"] / psteps, 2) for p in prof if p["launches"]})
            # SURVEY.md §8(d) asks for both fractions: the HBM-bound launch of the update is the value group's
            # Adam + Polyak sweep (36 B per parameter + the folded slab combines, algorithmic bytes from the launch site)
            sweeps = [p for p in prof if p["launches"] and "adam_ema_kernel" in p["name"] and p["bytes"] > 0]
            if sweeps:
                sw = max(sweeps, key=lambda p: p["bytes"])
                sw_ms = sw["total_ms"] / sw["launches"]
                gbs = sw["bytes"] / sw["launches"] / (sw_ms * 1e-3) / 1e9
                roof["hbm_bound_launch"] = dict(bound="hbm", launch=sw["name"], achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s",
                                                frac=gbs / PEAK_HBM_GBS, avg_launch_us=sw_ms * 1e3,
                                                bytes_per_launch=sw["bytes"] / sw["launches"])

    if rank == 0:
        steps_per_s = a.steps / elapsed
        # weak scaling: every rank pushes one batch-1024 through the update per step, so the whole-job
        # figure counts batch-1024 units of all ranks; one optimizer step consumes `world` of them
        # (its gradient is the mean over the world*1024 rows).  At N=1 the two numbers coincide.
        units_per_s = steps_per_s * world
        out = {
            "metric": "gradient-steps/sec (POR update, batch=%d)" % B,
            "value": units_per_s, "unit": "gradient-steps/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"POR S={S} A={A} H={H} L={L} B={B}/GPU (global {B * world}), "
                                   f"{rows_per_gpu * world} -row replay ({rows_per_gpu}/GPU) resident in HBM, "
                                   "device sampler + gather + full update per step",
                       "parallelism": f"dp{world}", "global_batch": B * world},
            "optimizer_steps_per_sec": steps_per_s,
            "samples_per_sec": steps_per_s * B * world,
            "algorithmic_tflops": steps_per_s * B * world * por_flops_per_sample() / 1e12,
            "final_losses": {"v_loss": float(lh[-1, 0]), "g_loss": float(lh[-1, 1]), "min_nll": float(lh[-1, 2])},
            "sustained_1000_updates_per_sec": sustained,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() if world > 1 else "none"),
        }
        if roof:
            out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu_baseline"] = units_per_s / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
