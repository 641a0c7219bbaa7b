"""How much of the POR step's idle-CU time can a second, independent chain of launches fill?
Two independent POR agents (own engines, own replay) are stepped alternately on two HIP streams; the aggregate
rate against one agent alone bounds what cross-step pipelining of the value / policy chains can win."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from porl_amd.agent.por import POR
from porl_amd.buffer.replay_buffer import PackedReplay
from porl_amd.util.synth import make_rows

S, A, H, L, B = 60, 2, 1024, 2, 1024
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def make(seed):
    rows = make_rows(200_000, S, A, seed=seed)
    rep = PackedReplay(rows, S, A, dev, seed=seed)
    torch.manual_seed(0)
    ag = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=A, max_batch=B),
             1000, 0.9, 10.0, device=dev)
    ag.async_losses = True
    return ag, rep


def run(agents, streams, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for (ag, rep), st in zip(agents, streams):
            with torch.cuda.stream(st):
                ag.update_from_replay(rep, B)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


a1, a2 = make(1), make(2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for n in (20, 200):
    one = run([a1], [s1], n)
two_same = run([a1, a2], [s1, s1], 200)
two = run([a1, a2], [s1, s2], 200)
# host-only cost of issuing one step: enqueue 100 steps right after a sync, stop the clock before waiting
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    a1[0].update_from_replay(a1[1], B)
host = (time.perf_counter() - t0) / 100
torch.cuda.synchronize()
print(f"host enqueue cost:    {host * 1e6:8.1f} us/step")
print(f"one agent:            {one * 1e6:8.1f} us/step  ({1 / one:7.0f} steps/s)")
print(f"two agents, 1 stream: {two_same * 1e6:8.1f} us per pair ({2 / two_same:7.0f} steps/s aggregate)")
print(f"two agents, 2 streams:{two * 1e6:8.1f} us per pair ({2 / two:7.0f} steps/s aggregate)  -> x{2 * one / two:.2f}")
