"""Size-independent properties of the CQL learn step at BASELINE config 3's full size (S=60, A=10, B=4096, Q-net
64-128-64), checked on the HIP path alone (src/porl/train/cql_trainer.py:88-124)."""
import numpy as np
import pytest
import torch

from porl_amd.util.synth import make_discrete_transitions
from test_cql_gpu import DEV, _trainer

pytestmark = pytest.mark.gpu
S, A, B, N = 60, 10, 4096, 20000


def _batch(seed):
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed)
    idx = np.random.default_rng(seed).choice(N, B, replace=False)
    return [torch.from_numpy(x[idx]).to(DEV) for x in (st, ac, rw, ns, dn)]


def _same_step(a, b):
    # first Adam step = lr * g / (|g| + 1e-8): weights whose gradient is ~1e-8 may flip sign with the summation order
    for (k, x), y in zip(a.q_network.state_dict().items(), b.q_network.state_dict().values()):
        diff = (x - y).abs()
        assert float(diff.max()) <= 2.1 * 5e-4, k
        assert float((diff > 2e-6).float().mean()) <= 2e-3, k


def test_row_order_of_the_minibatch_does_not_matter():
    """TD term and penalty are batch means (cql_trainer.py:104,83): a permuted minibatch is the same update."""
    st, ac, rw, ns, dn = _batch(1)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(DEV)
    a, b = _trainer(S, A, B, 3), _trainer(S, A, B, 3)
    la = a.learn_on(st, ac, rw, ns, dn)
    lb = b.learn_on(st[perm].contiguous(), ac[perm].contiguous(), rw[perm].contiguous(), ns[perm].contiguous(),
                    dn[perm].contiguous())
    np.testing.assert_allclose(la, lb, rtol=2e-6)
    _same_step(a, b)


def test_done_rows_ignore_the_target_network():
    """y = r + gamma * max_a Q_tgt(s', a) * (1 - done) (cql_trainer.py:99-102): with done = 1 everywhere a different
    target network gives the same update, bit for bit."""
    st, ac, rw, ns, _ = _batch(2)
    dn = torch.ones(B, device=DEV)
    a, b = _trainer(S, A, B, 3), _trainer(S, A, B, 3)
    with torch.no_grad():
        for p in b.target_network.parameters():
            p.mul_(-2.0)
    assert a.learn_on(st, ac, rw, ns, dn) == b.learn_on(st, ac, rw, ns, dn)
    for (k, x), y in zip(a.q_network.state_dict().items(), b.q_network.state_dict().values()):
        assert torch.equal(x, y), k


def test_alpha_zero_is_the_plain_td_update_and_the_penalty_is_reported():
    """loss = td + alpha * penalty (cql_trainer.py:111): with alpha = 0 the reported loss is the TD term alone while the
    penalty (logsumexp - ln A - Q[a] >= -ln A) is still measured; a uniform shift of the rewards changes the TD term only."""
    st, ac, rw, ns, dn = _batch(3)
    a, b = _trainer(S, A, B, 3, alpha=0.0), _trainer(S, A, B, 3, alpha=0.0)
    la, lb = a.learn_on(st, ac, rw, ns, dn), b.learn_on(st, ac, rw + 1.0, ns, dn)
    assert la == pytest.approx(a.last_td_loss, rel=1e-6) and lb == pytest.approx(b.last_td_loss, rel=1e-6)
    assert a.last_cql_penalty == b.last_cql_penalty and a.last_cql_penalty >= -np.log(A) - 1e-6
    assert la != lb
