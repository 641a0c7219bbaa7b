import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from porl_amd import engine as E
from porl_amd.engine import IqlEngine
from porl_amd.agent.por import POR
from porl_amd.util.synth import make_rows, split_rows
DEV = torch.device("cuda")
S, H, L, B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (60, 512, 3, 2048)))
rows = torch.from_numpy(make_rows(3 * B, S, 2, seed=7)).to(DEV)
torch.manual_seed(0)
agent = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=2, max_batch=B), 1000, 0.9, 10.0, device=DEV)
eng = agent._engine
for k in range(3):
    s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, 2)
    Bk = eng.load_batch(s, sp, r, d, sp)
    agent.v_optimizer.step_count += 1; agent.goal_policy_optimizer.step_count += 1
    hp = agent._hyper(Bk, agent.v_optimizer, agent.goal_policy_optimizer)
    g = {}
    for skinny in (0, 1, 0, 1):
        E.tune_set("skinny", skinny)
        eng.value_backward(hp)
        gv = [x.clone() for x in IqlEngine.views(eng.grads_vf, eng.tensor_table(0))]
        g.setdefault(("v", skinny), []).append(gv)
    eng.value_apply(hp)
    for skinny in (0, 1, 0, 1):
        E.tune_set("skinny", skinny)
        eng.policy_backward(hp)
        gp = [x.clone() for x in IqlEngine.views(eng.grads_pol, eng.tensor_table(1))]
        g.setdefault(("p", skinny), []).append(gp)
    eng.policy_apply(hp)
    agent.goal_lr_schedule.step()
    for ph in ("v", "p"):
        for i in range(len(g[(ph, 0)][0])):
            a0, a1, b0, b1 = g[(ph, 0)][0][i], g[(ph, 0)][1][i], g[(ph, 1)][0][i], g[(ph, 1)][1][i]
            sc = float(a0.abs().max())
            print(f"step{k} {ph}[{i}] {tuple(a0.shape)} max {sc:.2e}  gemm-vs-gemm {float((a0-a1).abs().max())/sc:.1e}  skinny-vs-skinny {float((b0-b1).abs().max())/sc:.1e}  gemm-vs-skinny {float((a0-b0).abs().max())/sc:.1e}")
