"""Latency of SORL.select_action(obs) for one observation (the rollout path of test.py:28-30), host sync included."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
from porl_amd.agent.sorl import SORL
dev = torch.device("cuda", 0)
for S, H in ((362, 512), (60, 1024)):
    torch.manual_seed(0)
    agent = SORL(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=2, layer_norm=False, action_size=2, max_batch=64,
                                 feature_dim=256), 1000, 0.9, 3.0, device=dev)
    x = torch.randn(1, S, device=dev)
    for _ in range(20): agent.select_action(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 500
    for _ in range(n): a = agent.select_action(x)
    el = time.perf_counter() - t0
    print(f"S={S} H={H}: select_action (B=1, returns numpy) {1e6 * el / n:.1f} us per call")
    xn = x.cpu().numpy()
    for _ in range(20): agent.select_action(xn)
    t0 = time.perf_counter()
    for _ in range(n): a = agent.select_action(xn)
    el = time.perf_counter() - t0
    print(f"    from an ndarray observation: {1e6 * el / n:.1f} us per call")
    from porl_amd import engine as E
    E.prof_enable(True)
    for _ in range(200): agent.select_action(x)
    prof = E.prof_read()
    E.prof_enable(False)
    dev_us = sum(p["total_ms"] for p in prof) * 1e3 / 200
    print(f"    device time per call {dev_us:.1f} us in {sum(p['launches'] for p in prof) / 200:.0f} launches: "
          + ", ".join(f"{p['name']} x{p['launches'] // 200}" for p in prof if p["launches"]))
