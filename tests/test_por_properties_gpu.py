"""Size-independent properties of the POR update at BASELINE config 2's full size (S=60, H=1024, B=1024), where the
CPU oracle is too slow to be the checker for every case: invariances the arithmetic of agent/por.py:73-112 implies,
checked on the HIP path alone."""
import numpy as np
import pytest
import torch

from porl_amd.util.synth import make_rows, split_rows
from test_por_gpu import DEV, _make_por

pytestmark = pytest.mark.gpu
S, A, H, B = 60, 2, 1024, 1024


def _batch(seed, n=B):
    rows = torch.from_numpy(make_rows(n, S, A, seed=seed)).to(DEV)
    return split_rows(rows, S, A)


def _same_step(a, b):
    """Parameters after ONE update from identical initial values.  The first Adam step is lr * g / (|g| + 1e-8): every
    weight moves by ~lr whatever the size of its gradient, so for the handful of weights whose gradient is ~1e-8 the
    order of an fp32 sum decides the sign and the two results differ by up to 2 lr there; everything
    else agrees to summation-order noise."""
    for (k, x), y in zip(a.state_dict().items(), b.state_dict().values()):
        diff = (x - y).abs()
        assert float(diff.max()) <= 2.1e-4, k
        assert float((diff > 2e-6).float().mean()) <= 1e-3, k


def test_row_order_of_the_minibatch_does_not_matter():
    """Every loss term is a mean over rows (por.py:87,106): a permuted minibatch changes only the order of fp32 sums."""
    s, r, sp, d, _ = _batch(1)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(DEV)
    a, b = _make_por(S, H, 2, B), _make_por(S, H, 2, B)
    la = a.por_residual_update(s, sp, r, d)
    lb = b.por_residual_update(s[perm].contiguous(), sp[perm].contiguous(), r[perm].contiguous(), d[perm].contiguous())
    np.testing.assert_allclose(la, lb, rtol=2e-6)
    _same_step(a, b)


def test_terminal_rows_ignore_the_target_network():
    """target = r + (1 - d) * gamma * next_v (por.py:84): with d = 1 everywhere next_v is multiplied by zero, so a
    perturbed target network changes nothing but the target network itself."""
    s, r, sp, _, _ = _batch(2)
    d = torch.ones(B, device=DEV)
    a, b = _make_por(S, H, 2, B), _make_por(S, H, 2, B)
    with torch.no_grad():
        for p in b.v_target.parameters():
            p.mul_(1.5)
    la, lb = a.por_residual_update(s, sp, r, d), b.por_residual_update(s, sp, r, d)
    assert la == lb
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if not k.startswith("v_target."):
            assert torch.equal(sa[k], sb[k]), k


def test_zero_learning_rates_leave_the_online_networks_untouched():
    """Adam with lr = 0 is the identity on the parameters; the target still moves by the Polyak rule (util.py:54-56)."""
    s, r, sp, d, _ = _batch(3)
    a = _make_por(S, H, 2, B, value_lr=0.0, policy_lr=0.0)
    before = {k: v.clone() for k, v in a.state_dict().items()}
    v_loss, g_loss = a.por_residual_update(s, sp, r, d)
    assert np.isfinite([v_loss, g_loss]).all()
    after = a.state_dict()
    for k in before:
        if k.startswith("v_target."):
            want = before[k] * (1.0 - a.beta) + a.beta * before[k.replace("v_target.", "vf.", 1)]
            assert float((after[k] - want).abs().max()) <= 1e-7, k
        else:
            assert torch.equal(after[k], before[k]), k


def test_a_minibatch_repeated_twice_gives_the_same_update():
    """Means over 2B rows that are B rows twice equal the means over the B rows: same losses, same first step."""
    s, r, sp, d, _ = _batch(4)
    a, b = _make_por(S, H, 2, B), _make_por(S, H, 2, 2 * B)
    la = a.por_residual_update(s, sp, r, d)
    lb = b.por_residual_update(torch.cat([s, s]), torch.cat([sp, sp]), torch.cat([r, r]), torch.cat([d, d]))
    np.testing.assert_allclose(la, lb, rtol=2e-6)
    _same_step(a, b)
