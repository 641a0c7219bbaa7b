"""Host-side cost of issuing one POR update (no GPU wait): how many microseconds of CPU per step go into the Python
wrapper, the ctypes calls and the HIP launches.  The queue is drained before every measurement and only a few steps
are issued, so the host never blocks on a full queue."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from porl_amd.agent.por import POR
from porl_amd.buffer.replay_buffer import PackedReplay
from porl_amd.util.synth import make_rows

S, A, H, L, B = 60, 2, 1024, 2, 1024
dev = torch.device("cuda", 0)
rows = make_rows(200_000, S, A, seed=1)
rep = PackedReplay(rows, S, A, dev, seed=1)
torch.manual_seed(0)
ag = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=A, max_batch=B),
         1000, 0.9, 10.0, device=dev)
ag.async_losses = True
eng = ag._engine
for _ in range(30):
    ag.update_from_replay(rep, B)


def host_us(fn, n=8, reps=20):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t0) / n)
    torch.cuda.synchronize()
    return best * 1e6


hp = ag._hyper(B, ag.v_optimizer, ag.goal_policy_optimizer)
print(f"update_from_replay (python + 2 native calls): {host_us(lambda: ag.update_from_replay(rep, B)):7.1f} us")
print(f"eng.step(hp) alone (one native call, all launches of the step): {host_us(lambda: eng.step(hp)):7.1f} us")
print(f"eng.load_batch_sampled alone (one launch): {host_us(lambda: eng.load_batch_sampled(rep.rows, B, 1, 1, A, False)):7.1f} us")
print(f"ag._hyper(...) (python only): {host_us(lambda: ag._hyper(B, ag.v_optimizer, ag.goal_policy_optimizer)):7.1f} us")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    ag.update_from_replay(rep, B)
torch.cuda.synchronize()
print(f"GPU-bound rate: {(time.perf_counter() - t0) / 300 * 1e6:7.1f} us/step")
