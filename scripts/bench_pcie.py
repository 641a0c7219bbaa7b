"""POR update fed from HOST memory every step (the reference's `.to(device)` hand-over, por_train.py:71-82) — the
PCIe-inclusive rate DESIGN.md quotes beside the HBM-resident headline."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
from porl_amd.agent.por import POR
from porl_amd.util.synth import make_rows
S, A, H, L, B = 60, 2, 1024, 2, 1024
dev = torch.device("cuda", 0)
torch.manual_seed(0)
agent = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=A, max_batch=B), 1000, 0.9, 10.0, device=dev)
agent.async_losses = True
rows = torch.from_numpy(make_rows(64 * B, S, A, seed=0))
for pinned in (False, True):
    host = rows.pin_memory() if pinned else rows
    def step(i):
        data = host[(i % 64) * B:(i % 64 + 1) * B].to(dev, non_blocking=pinned)
        agent.por_residual_update(data[:, :S], data[:, S + 1:-A - 1], data[:, S], data[:, -A - 1])
    for i in range(20): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    for i in range(n): step(i)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"host batches ({'pinned' if pinned else 'pageable'}), one H2D copy of {B * rows.shape[1] * 4 / 1e6:.2f} MB per update: "
          f"{n / el:.0f} updates/s ({1e3 * el / n:.3f} ms)")
