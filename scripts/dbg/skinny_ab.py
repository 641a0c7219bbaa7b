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
res = {}
for skinny in (0, 1):
    E.tune_set("skinny", skinny)
    torch.manual_seed(0)
    agent = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=2, max_batch=B), 1000, 0.9, 10.0, device=DEV)
    eng = agent._engine
    out = []
    for k in range(3):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, 2)
        Bk = eng.load_batch(s, sp, r, d, sp)
        agent.v_optimizer.step_count += 1; agent.goal_policy_optimizer.step_count += 1
        hp = agent._hyper(Bk, agent.v_optimizer, agent.goal_policy_optimizer)
        eng.value_backward(hp)
        gv = [g.clone() for g in IqlEngine.views(eng.grads_vf, eng.tensor_table(0))]
        eng.value_apply(hp)
        eng.policy_backward(hp)
        gp = [g.clone() for g in IqlEngine.views(eng.grads_pol, eng.tensor_table(1))]
        eng.policy_apply(hp)
        agent.goal_lr_schedule.step()
        out.append((gv, gp, eng.stats[:3].clone()))
    res[skinny] = out
E.tune_set("skinny", 1)
for k in range(3):
    for which, nm in ((0, "vf"), (1, "pol")):
        for i, (a, b) in enumerate(zip(res[0][k][which], res[1][k][which])):
            sc = float(a.abs().max())
            err = float((a - b).abs().max())
            print(f"step{k} {nm}[{i}] shape {tuple(a.shape)} max|g| {sc:.3e} max diff {err:.3e} rel {err / max(sc, 1e-30):.2e}")
    print("stats", res[0][k][2].tolist(), res[1][k][2].tolist())
