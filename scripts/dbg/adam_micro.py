"""Stand-alone timing of the Adam(+EMA) sweep at the two group sizes of the headline POR agent (no folded combines)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from porl_amd import engine as E
dev = torch.device("cuda")
for n, tgt in ((1173624, False), (2226180, True), (4 * 1173624, False)):
    n = (n + 3) // 4 * 4
    p, g, m, v, t = (torch.randn(n, device=dev) for _ in range(5))
    v.abs_()
    for _ in range(20):
        E.adam_ema(p, g, m, v, t if tgt else None, 1e-4, 3, ema_beta=0.005 if tgt else 0.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        E.adam_ema(p, g, m, v, t if tgt else None, 1e-4, 3, ema_beta=0.005 if tgt else 0.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    byts = n * (36 if tgt else 28)
    print(f"n={n} target={tgt}: {us:.2f} us per launch (back to back), {byts / us / 1e6:.2f} TB/s")
