"""Create / use / drop cycles of every engine kind: device memory, signal memory and host memory must come back
(handles of the C ABI own host-side plans; the pipelined update owns counters in signal memory and a side stream)."""
import gc
import resource
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _free_bytes():
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]


def _cycle_por():
    from porl_amd.agent.por import POR
    from porl_amd.buffer.replay_buffer import PackedReplay
    from porl_amd.util.synth import make_rows
    S, A, H, B = 24, 2, 128, 64
    replay = PackedReplay(make_rows(4096, S, A, seed=3), S, A, DEV, rank=0, world=1, seed=0)
    agent = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=2, layer_norm=False, action_size=A, max_batch=B),
                max_steps=100, tau=0.9, alpha=10.0, device=DEV)
    agent.async_losses, agent.pipeline = True, True
    for _ in range(4):
        agent.update_from_replay(replay, B)         # pipelined: counters in signal memory, side stream, staging slots
    agent.flush()
    sd = agent.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values() if torch.is_tensor(v) and v.is_floating_point())


def _cycle_cql():
    from porl_amd.train.cql_trainer import CQLTrainer
    from porl_amd.util.synth import make_discrete_transitions
    S, A, B, N = 12, 4, 64, 512
    t = CQLTrainer(state_size=S, action_size=A, gamma=0.99, device=DEV, batch_size=B)
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=1)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    np.random.seed(0)
    assert np.isfinite([t.learn() for _ in range(3)]).all()


def _cycle_iqn():
    from porl_amd.train.iqn_trainer import IQNTrainer
    from porl_amd.util.synth import make_discrete_transitions
    S, A, B, N = 12, 4, 32, 256
    t = IQNTrainer(S, A, gamma=0.99, device=DEV, batch_size=B, hidden_size=64, embedding_dim=16)
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=2)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    np.random.seed(0)
    assert np.isfinite([t.learn() for _ in range(3)]).all()




@pytest.mark.parametrize("cycle", [_cycle_por, _cycle_cql, _cycle_iqn])
def test_engines_give_their_memory_back(cycle):
    # first uses: lazy one-time allocations.  Every pipelined agent takes its side stream from torch's pool of 32 streams
    # per device, round-robin, and a pool stream's FIRST kernel launch makes the runtime create its hardware queue
    # (~1.2 MiB of device memory, scripts/dbg/leak_probe.py): 34 cycles touch every pool stream once.
    for _ in range(34 if cycle is _cycle_por else 2):
        cycle()
    free0, rss0 = _free_bytes(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for _ in range(25):
        cycle()
    free1, rss1 = _free_bytes(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 25 create/use/drop cycles"
    assert rss1 - rss0 < 64 << 10, f"host peak RSS grew by {(rss1 - rss0) >> 10} MiB over 25 cycles"     # ru_maxrss is in KiB
