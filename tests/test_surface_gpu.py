"""The reference's stand-alone surface classes that no training script calls — TwinQ, ValueFunction
(agent/value_functions.py:6-28), DeterministicPolicy (agent/policy.py:62-73) — and GaussianPolicy.act(enable_grad=True)
(agent/policy.py:30-33): forwards and gradients on the device (porl_amd/util/hip_mlp.py: fp32-MFMA products with their
epilogues) against the numpy oracle's mlp_forward / mlp_backward (oracle/por_oracle.py, pinned to the reference by
tests/test_oracle_golden.py).  Tolerance: 2e-6 of the largest magnitude (fp32, different summation orders)."""
import numpy as np
import pytest
import torch

import oracle.por_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _np(mod, prefix):
    return {f"{prefix}.{k}": v.detach().cpu().numpy() for k, v in mod.state_dict().items()}


def _close(got, want, what):
    want = np.asarray(want, dtype=np.float64)
    scale = max(float(np.abs(want).max()), 1e-6)
    err = float(np.abs(np.asarray(got.detach().cpu() if isinstance(got, torch.Tensor) else got, dtype=np.float64) - want).max())
    assert err <= 2e-6 * scale + 1e-9, (what, err, scale)


@pytest.mark.parametrize("B,S,A,H,L", [(37, 17, 5, 48, 2), (300, 60, 2, 256, 2), (4, 8, 3, 200, 3)])
def test_twinq_valuefunction_deterministic_policy_forward_and_gradients(B, S, A, H, L):
    from porl_amd.agent.policy import DeterministicPolicy
    from porl_amd.agent.value_functions import TwinQ, ValueFunction
    g = torch.Generator().manual_seed(B)
    s, a = torch.randn(B, S, generator=g), torch.rand(B, A, generator=g) * 2 - 1
    torch.manual_seed(1)
    tq, vf, dp = TwinQ(S, A, H, L).to(DEV), ValueFunction(S, H, L).to(DEV), DeterministicPolicy(S, A, H, L).to(DEV)
    sa = np.concatenate([s.numpy(), a.numpy()], 1)
    # ---- forwards ------------------------------------------------------------------------------------------------
    q1, q2 = tq.both(s.to(DEV), a.to(DEV))
    P = {**_np(tq.q1, "q1"), **_np(tq.q2, "q2"), **_np(vf.v, "v"), **_np(dp.net, "net")}
    r1, c1 = O.mlp_forward(P, "q1", sa, L)
    r2, _ = O.mlp_forward(P, "q2", sa, L)
    assert q1.shape == (B,) and q2.shape == (B,)
    _close(q1, r1[:, 0], "q1"); _close(q2, r2[:, 0], "q2")
    _close(tq(s.to(DEV), a.to(DEV)), np.minimum(r1, r2)[:, 0], "min(q1, q2)")
    rv, cv = O.mlp_forward(P, "v", s.numpy(), L)
    v = vf(s.to(DEV))
    assert v.shape == (B,)
    _close(v, rv[:, 0], "v")
    rp, cp = O.mlp_forward(P, "net", s.numpy(), L, out_act="tanh")
    act = dp(s.to(DEV))
    assert act.shape == (B, A) and float(act.detach().abs().max()) <= 1.0
    _close(act, rp, "deterministic policy")
    assert torch.equal(dp.act(s.to(DEV)), act) and not dp.act(s.to(DEV)).requires_grad
    # ---- gradients: d(sum of w * output)/d(parameters) and d/d(input) -------------------------------------------------
    wq = torch.randn(B, generator=g)
    (q1 * wq.to(DEV)).sum().backward()
    G = O.mlp_backward(P, "q1", c1, wq.numpy()[:, None], L)
    for k, p in tq.q1.named_parameters():
        _close(p.grad, G["q1." + k], "q1." + k)
    assert all(p.grad is None for p in tq.q2.parameters())
    wa = torch.randn(B, A, generator=g)
    x = s.to(DEV).requires_grad_(True)
    (dp.act(x, enable_grad=True) * wa.to(DEV)).sum().backward()
    G = O.mlp_backward(P, "net", cp, wa.numpy(), L, out_act="tanh")
    for k, p in dp.net.named_parameters():
        _close(p.grad, G["net." + k], "net." + k)
    # input gradient: through the oracle's first layer
    d = wa.numpy() * (1 - cp["out"] ** 2)
    dh = d @ P[f"net.{2 * L}.weight"]
    for i in reversed(range(L)):
        dz = dh * (cp["h"][i] > 0)
        dh = dz @ P[f"net.{2 * i}.weight"]
    _close(x.grad, dh, "d/d(obs)")


def test_gaussian_policy_act_with_grad_matches_the_engine_forward_and_differentiates():
    from types import SimpleNamespace
    from porl_amd.agent.por import POR
    S, H, B = 60, 64, 33
    torch.manual_seed(0)
    agent = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=2, layer_norm=False, action_size=2, max_batch=64),
                1000, 0.9, 10.0, device=DEV)
    pol = agent.goal_policy
    obs = torch.randn(B, S, generator=torch.Generator().manual_seed(2)).to(DEV)
    mean_engine = pol.act(obs, deterministic=True)                       # fused engine forward, no grad
    mean_grad = pol.act(obs, deterministic=True, enable_grad=True)
    assert mean_grad.requires_grad and not mean_engine.requires_grad
    np.testing.assert_allclose(mean_grad.detach().cpu().numpy(), mean_engine.cpu().numpy(), atol=2e-6)
    w = torch.randn(B, S, generator=torch.Generator().manual_seed(3))
    (mean_grad * w.to(DEV)).sum().backward()
    P = {"net." + k: v.detach().cpu().numpy() for k, v in pol.net.state_dict().items()}
    _, c = O.mlp_forward(P, "net", obs.cpu().numpy(), 2)
    G = O.mlp_backward(P, "net", c, w.numpy(), 2)
    for k, p in pol.net.named_parameters():
        _close(p.grad, G["net." + k], "net." + k)
    sample = pol.act(obs, enable_grad=True)
    assert sample.shape == (B, S) and sample.requires_grad


def test_surface_forwards_have_no_cpu_path():
    from porl_amd import _native as N
    from porl_amd.agent.value_functions import ValueFunction
    with pytest.raises(N.NativeError):
        ValueFunction(8)(torch.zeros(2, 8))
