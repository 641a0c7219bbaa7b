"""Diagnostic (GPU, run by hand: python tests/diag_param_error.py; not collected by pytest): where do HIP-vs-oracle
parameter differences at H=1024 come from?
Compares engine fp32, oracle fp32 and oracle fp64 after K updates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
import oracle.por_oracle as O
from porl_amd.agent.por import POR
from porl_amd.util.synth import make_rows, split_rows

S, H, L, B, K = 60, 1024, 2, 1024, 3
dev = torch.device("cuda")
torch.manual_seed(0)
agent = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=2, max_batch=B), 1000, 0.9, 10.0, device=dev)
init = {k: v.cpu().numpy() for k, v in agent.state_dict().items()}
rows = make_rows(K * B, S, 2, seed=1)
o32 = O.PorOracle(init, S, H, L)
for k in range(K):
    s, r, sp, d, _ = split_rows(rows[k*B:(k+1)*B], S, 2)
    o32.por_residual_update(s, sp, r, d)
O.F32 = np.float64
o64 = O.PorOracle(init, S, H, L)
for k in range(K):
    s, r, sp, d, _ = split_rows(rows[k*B:(k+1)*B].astype(np.float64), S, 2)
    o64.por_residual_update(s, sp, r, d)
O.F32 = np.float32
drows = torch.from_numpy(rows).to(dev)
for k in range(K):
    s, r, sp, d, _ = split_rows(drows[k*B:(k+1)*B], S, 2)
    print(agent.por_residual_update(s, sp, r, d))
eng = {k: v.cpu().numpy() for k, v in agent.state_dict().items()}
print(f"{'tensor':28s} {'|e-o32|':>10s} {'|e-o64|':>10s} {'|o32-o64|':>10s}  n(e-o64>1e-6) n(o32-o64>1e-6)")
for k in eng:
    e, a, b = eng[k].astype(np.float64), o32.P[k].astype(np.float64), o64.P[k]
    d1, d2, d3 = np.abs(e-a), np.abs(e-b), np.abs(a-b)
    print(f"{k:28s} {d1.max():10.3e} {d2.max():10.3e} {d3.max():10.3e}  {(d2>1e-6).sum():8d} {(d3>1e-6).sum():8d}  n={e.size}")
