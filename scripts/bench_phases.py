"""Per-kernel timing of one POR update (HIP events via the library's profiling hooks) under tuning knobs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from porl_amd.agent.por import POR
from porl_amd import engine as E
from porl_amd.util.synth import make_rows, split_rows

S, H, L, B = 60, 1024, 2, 1024
dev = torch.device("cuda")
torch.manual_seed(0)
agent = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=2, max_batch=B), 1000, 0.9, 10.0, device=dev)
agent.async_losses = True
rows = torch.from_numpy(make_rows(B, S, 2, seed=1)).to(dev)
s, r, sp, d, _ = split_rows(rows, S, 2)

def run(tag, steps=30):
    for _ in range(5):
        agent.por_residual_update(s, sp, r, d)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        agent.por_residual_update(s, sp, r, d)
    t1.record(); torch.cuda.synchronize()
    total = t0.elapsed_time(t1) / steps
    E.prof_enable(True)
    for _ in range(steps):
        agent.por_residual_update(s, sp, r, d)
    prof = E.prof_read()
    E.prof_enable(False)
    print(f"== {tag}: {total*1e3:.1f} us/step (no sampling)")
    for p in sorted(prof, key=lambda p: -p["total_ms"]):
        if p["launches"]:
            us = p["total_ms"] * 1e3 / p["launches"]
            tf = p["flops"] / p["launches"] / (us * 1e-6) / 1e12 if p["flops"] else 0
            print(f"   {p['name']:48s} n/step={p['launches']/steps:4.1f} avg={us:7.1f}us  {tf:6.1f} TF")

for pad in [int(x) for x in (sys.argv[1:] or ["0"])]:
    E.tune_set("gemm_lds_pad", pad)
    run(f"lds_pad={pad}")
