"""Pipelined vs one-stream update rate for small networks (launch-bound configurations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from porl_amd.agent.por import POR
from porl_amd.buffer.replay_buffer import PackedReplay
from porl_amd.util.synth import make_rows
dev = torch.device("cuda", 0)
S, A = 60, 2
replay = PackedReplay(make_rows(200_000, S, A, seed=1), S, A, dev, rank=0, world=1, seed=0)
for H, B in ((256, 256), (256, 1024), (512, 1024), (1024, 256), (1024, 1024)):
    for pipe in (False, True):
        args = SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=2, layer_norm=False, action_size=A, max_batch=B)
        torch.manual_seed(0)
        agent = POR(args, max_steps=1000, tau=0.9, alpha=10.0, device=dev)
        agent.async_losses = True
        agent.pipeline = pipe
        for _ in range(50): agent.update_from_replay(replay, B)
        agent.flush(); torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 1000
        for _ in range(n): agent.update_from_replay(replay, B)
        agent.flush(); torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(f"H={H} B={B} pipeline={int(pipe)}: {n / el:8.0f} updates/s ({1e6 * el / n:.1f} us)", flush=True)
