"""`torch.ops.porl_hip.*`: the C-ABI entry points registered as PyTorch custom operators (porl_amd/ops.py)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from porl_amd import _native as N
from porl_amd import ops as O
from porl_amd.util.synth import make_rows, split_rows


def test_operators_are_registered_with_their_schemas():
    for name in O.SCHEMAS:
        op = getattr(torch.ops.porl_hip, name)
        assert op.default._schema.name == f"porl_hip::{name}"
    # in-place arguments are declared as such (alias annotations), so functionalization cannot reorder them
    s = str(torch.ops.porl_hip.adam_ema_sweep.default._schema)
    assert "Tensor(a!) p" in s and "Tensor(b!) m" in s and "Tensor(c!) v" in s


def test_no_cpu_kernel():
    p = torch.zeros(8)
    with pytest.raises(N.NativeError):
        torch.ops.porl_hip.adam_ema_sweep(p, p.clone(), p.clone(), p.clone(), None, 1e-3, 1, 0.9, 0.999, 1e-8, 0.0)
    with pytest.raises(N.NativeError):
        torch.ops.porl_hip.replay_gather(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        torch.ops.porl_hip.mlp_forward(12345, torch.zeros(1, 4), 0)      # unknown handle (CPU tensor: no kernel either)


@pytest.mark.gpu
def test_building_block_operators_match_the_engine_calls():
    from porl_amd import engine as E
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    p, gr, m, v, t = (torch.randn(1024, generator=g).to(dev) for _ in range(5))
    v = v.abs()
    p2, m2, v2, t2 = p.clone(), m.clone(), v.clone(), t.clone()
    torch.ops.porl_hip.adam_ema_sweep(p, gr, m, v, t, 1e-3, 3, 0.9, 0.999, 1e-8, 0.005)
    E.adam_ema(p2, gr, m2, v2, t2, 1e-3, 3, 0.9, 0.999, 1e-8, 0.005)
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2) and torch.equal(t, t2)
    rows = torch.randn(100, 124, generator=g).to(dev)
    idx = torch.randint(0, 100, (17,), generator=g).to(dev)
    assert torch.equal(torch.ops.porl_hip.replay_gather(rows, idx), rows[idx])
    a, b, bias = torch.randn(70, 96, generator=g).to(dev), torch.randn(50, 96, generator=g).to(dev), torch.randn(50, generator=g).to(dev)
    got = torch.ops.porl_hip.gemm_f32(a, b, bias, 1)
    want = torch.relu(a.double() @ b.double().T + bias.double())
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), atol=2e-5)
    idx = torch.ops.porl_hip.sample_indices(1000, 1000, 5, 0, rows)
    assert sorted(idx.cpu().tolist()) == list(range(1000))


@pytest.mark.gpu
def test_por_step_operator_equals_the_agent_update():
    from porl_amd.agent.por import POR
    dev = torch.device("cuda")
    S, H, B = 60, 64, 32
    args = SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=2, layer_norm=False, action_size=2, max_batch=B)
    torch.manual_seed(0)
    a = POR(args, 1000, 0.9, 10.0, device=dev)
    torch.manual_seed(0)
    b = POR(args, 1000, 0.9, 10.0, device=dev)
    h = O.register_engine(b._engine)
    rows = torch.from_numpy(make_rows(2 * B, S, 2, seed=2)).to(dev)
    for k in range(2):
        s, r, sp, d, _ = split_rows(rows[k * B:(k + 1) * B], S, 2)
        want = a.por_residual_update(s, sp, r, d)
        lr = a.goal_lr_schedule._lr(k)          # the schedule steps AFTER the policy Adam (por.py:109-110)
        got = torch.ops.porl_hip.por_step(h, s, sp, r, d, None, 0.9, 0.99, 10.0, 0.005, 1e-4, lr, k + 1, k + 1)
        assert tuple(got[:2].tolist()) == want
    for (k1, v1), v2 in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(v1, v2), k1
    # the two halves as separate operators
    s, r, sp, d, _ = split_rows(rows[:B], S, 2)
    want = a.por_residual_update(s, sp, r, d)
    v_loss = torch.ops.porl_hip.iql_value_step(h, s, sp, r, d, 0.9, 0.99, 0.005, 1e-4, 3)
    g = torch.ops.porl_hip.awr_policy_step(h, s, 10.0, a.goal_lr_schedule._lr(2), 3)
    assert (float(v_loss[0]), float(g[0])) == want
    mean = torch.ops.porl_hip.mlp_forward(h, s[:4], 0)
    assert torch.equal(mean, a.goal_policy(s[:4]).mean)
    O.release_engine(h)
