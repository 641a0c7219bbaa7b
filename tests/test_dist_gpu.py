"""Distributional DQN variants against reference-generated goldens (oracle/gen_golden.py: gen_qr,
gen_c51 — five `learn()` steps under the reference's numpy index stream, from its initial online / target
parameters): QR-DQN quantile-Huber loss and C51 projection + cross-entropy, loss heads in csrc/dist_losses.hpp."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from porl_amd.util.synth import make_discrete_transitions

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _np_sd(m):
    return {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


def _fill(t, N, S, A, seed_data, reward_scale=1.0):
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    t.replay_buffer = type(t.replay_buffer)(N, (S,), DEV)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]) * reward_scale, ns[i], bool(dn[i]))


def _load(t, z):
    t.q_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init/").items()})
    t.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init_target/").items()})


def test_qr_dqn_learn_matches_reference_golden():
    from porl_amd.train.qr_dqn_trainer import QRDQNTrainer
    z, _ = load_golden("qrdqn_s9_a5_n12")
    S, A, NQ, B, K, N, seed_model, seed_data, seed_np, h0, h1 = (int(v) for v in z["meta"])
    t = QRDQNTrainer(S, A, float(z["gamma"]), device=DEV, network_hidden_sizes=[h0, h1], num_quantiles=NQ,
                     kappa=float(z["kappa"]), batch_size=B)
    assert list(t.q_network.state_dict().keys()) == list(sub(z, "init/").keys())
    _load(t, z)
    _fill(t, N, S, A, seed_data)
    np.testing.assert_allclose(t.tau.cpu().numpy(), ((2 * np.arange(NQ) + 1) / (2 * NQ))[None], rtol=2e-7)
    np.random.seed(seed_np)
    for k in range(K):
        np.testing.assert_allclose(t.learn(), z["loss"][k], rtol=2e-5)
    got = _np_sd(t.q_network)
    for k, v in sub(z, "final/").items():
        np.testing.assert_allclose(got[k], v, atol=1e-5, err_msg=k)
    x = torch.from_numpy(z["probe_x"]).to(DEV)
    np.testing.assert_allclose(t.q_network.get_mean_q_values(x).cpu().numpy(), z["probe_mean_q"], atol=2e-6)
    assert t.q_network(x).shape == (4, A, NQ)


def test_c51_learn_matches_reference_golden():
    from porl_amd.train.c51_trainer import C51Trainer
    z, _ = load_golden("c51_s9_a5_n21")
    S, A, NA, B, K, N, seed_model, seed_data, seed_np, h0, h1 = (int(v) for v in z["meta"])
    t = C51Trainer(S, A, float(z["gamma"]), device=DEV, atom_size=NA, v_min=float(z["v_min"]), v_max=float(z["v_max"]),
                   network_hidden_sizes=[h0, h1], batch_size=B)
    assert list(t.q_network.state_dict().keys()) == list(sub(z, "init/").keys())
    _load(t, z)
    _fill(t, N, S, A, seed_data, reward_scale=2.0)
    np.random.seed(seed_np)
    for k in range(K):
        np.testing.assert_allclose(t.learn(), z["loss"][k], rtol=2e-5)
    got = _np_sd(t.q_network)
    for k, v in sub(z, "final/").items():
        np.testing.assert_allclose(got[k], v, atol=1e-5, err_msg=k)
    x = torch.from_numpy(z["probe_x"]).to(DEV)
    np.testing.assert_allclose(t.q_network(x).cpu().numpy(), z["probe_logp"], atol=5e-6)
    np.testing.assert_allclose(t.q_network.get_q_values(x).cpu().numpy(), z["probe_q"], atol=5e-6)


def test_c51_projection_conserves_probability_mass():
    """Size-independent property of the categorical projection (c51_trainer.py:78-137): whatever the rewards, the
    projected distribution sums to 1, so for a uniform online distribution the loss is exactly log(atoms)."""
    from porl_amd import _native as N
    B, A, NA = 257, 3, 51
    g = torch.Generator().manual_seed(0)
    lt = torch.randn(B, A * NA, generator=g).to(DEV)
    lc = torch.zeros(B, A * NA, device=DEV)                         # uniform log-probabilities: -log(NA) everywhere
    act = torch.randint(0, A, (B,), generator=g).to(DEV)
    rew = (5 * torch.randn(B, generator=g)).to(DEV)                 # far beyond [v_min, v_max]: both clamps
    done = (torch.rand(B, generator=g) < 0.3).float().to(DEV)
    sup = torch.linspace(-10, 10, NA).to(DEV)
    dl, rl = torch.empty_like(lc), torch.empty(B, device=DEV)
    N.check(N.lib().porl_c51_loss(N.ptr(lc), N.ptr(lt), A * NA, N.ptr(act), N.ptr(rew), N.ptr(done), N.ptr(sup), B, A, NA,
                                  0.99, -10.0, 10.0, N.ptr(dl), N.ptr(rl), N.current_stream_ptr(lc)), "porl_c51_loss")
    np.testing.assert_allclose(rl.cpu().numpy(), np.log(NA), rtol=2e-6)
    d = dl.view(B, A, NA).cpu().numpy()
    np.testing.assert_allclose(d.sum(axis=2), 0.0, atol=1e-7)       # softmax gradient rows sum to zero
    taken = act.cpu().numpy()
    for b in range(0, B, 37):
        others = [a for a in range(A) if a != taken[b]]
        assert not d[b, others].any()


def test_iqn_quantile_huber_loss_head_matches_reference_golden():
    """IQNTrainer.quantile_huber_loss (iqn_trainer.py:136-149) and its autograd gradient.  Only this head is pinned:
    upstream's IQN learn() cannot run (IQNTrainer and IQNNetwork disagree on the constructor and on get_q_values)."""
    from porl_amd import _native as N
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "iqn_quantile_huber.npz"))
    cur, tgt, taus = (torch.from_numpy(z[k]).to(DEV) for k in ("cur", "target", "taus"))
    B, NP = cur.shape
    dcur, rl, out = torch.empty_like(cur), torch.empty(B, device=DEV), torch.zeros(1, device=DEV)
    N.check(N.lib().porl_iqn_quantile_huber(N.ptr(cur), N.ptr(tgt), N.ptr(taus), B, NP, tgt.shape[1], float(z["kappa"]),
                                            N.ptr(dcur), N.ptr(rl), N.current_stream_ptr(cur)), "porl_iqn_quantile_huber")
    N.check(N.lib().porl_reduce_mean(N.ptr(rl), B, N.ptr(out), N.current_stream_ptr(cur)), "porl_reduce_mean")
    np.testing.assert_allclose(float(out), float(z["loss"]), rtol=2e-6)
    np.testing.assert_allclose(dcur.cpu().numpy(), z["dcur"], atol=2e-8, rtol=2e-5)


@pytest.mark.parametrize("head", ["qr", "c51"])
def test_out_of_range_action_is_never_used_as_an_index(head):
    """Round-2 advisor finding: an action outside [0, A) from replay must not become a device address.  The loss heads
    give that row a NaN loss term and a zero gradient (and touch nothing outside the row); the trainers turn it into
    the IndexError the reference's `gather` raises (qr_dqn_trainer.py:147, c51_trainer.py:155)."""
    from porl_amd import _native as N
    B, A, NQ = 9, 3, 8
    g = torch.Generator().manual_seed(1)
    guard = torch.full((B + 2, A * NQ), 7.0, device=DEV)           # one sentinel row before and after the gradient rows
    dz = guard[1:B + 1]
    zc, zo, zt = (torch.randn(B, A * NQ, generator=g).to(DEV) for _ in range(3))
    act = torch.randint(0, A, (B,), generator=g)
    act[2], act[5] = A, -1                                          # both sides of the valid range
    act = act.to(DEV)
    rew, done = torch.randn(B, generator=g).to(DEV), torch.zeros(B, device=DEV)
    rl = torch.empty(B, device=DEV)
    if head == "qr":
        N.check(N.lib().porl_qr_loss(N.ptr(zc), N.ptr(zo), N.ptr(zt), A * NQ, N.ptr(act), N.ptr(rew), N.ptr(done), B, A, NQ,
                                     0.99, 1.0, N.ptr(dz), N.ptr(rl), N.current_stream_ptr(zc)), "porl_qr_loss")
    else:
        sup = torch.linspace(-2, 2, NQ).to(DEV)
        N.check(N.lib().porl_c51_loss(N.ptr(zc), N.ptr(zt), A * NQ, N.ptr(act), N.ptr(rew), N.ptr(done), N.ptr(sup), B, A, NQ,
                                      0.99, -2.0, 2.0, N.ptr(dz), N.ptr(rl), N.current_stream_ptr(zc)), "porl_c51_loss")
    rl, d = rl.cpu().numpy(), dz.cpu().numpy()
    assert np.isnan(rl[[2, 5]]).all() and np.isfinite(np.delete(rl, [2, 5])).all()
    assert not d[[2, 5]].any() and np.isfinite(d).all()
    assert (guard[0] == 7.0).all() and (guard[B + 1] == 7.0).all()


def test_trainer_raises_index_error_on_out_of_range_action():
    from porl_amd.train.qr_dqn_trainer import QRDQNTrainer
    S, A = 6, 3
    torch.manual_seed(0)
    t = QRDQNTrainer(S, A, 0.99, device=DEV, network_hidden_sizes=[32, 32], num_quantiles=8, batch_size=16)
    g = torch.Generator().manual_seed(0)
    st, ns = torch.randn(16, S, generator=g).to(DEV), torch.randn(16, S, generator=g).to(DEV)
    ac = torch.randint(0, A, (16,), generator=g)
    rw, dn = torch.randn(16, generator=g).to(DEV), torch.zeros(16, device=DEV)
    assert np.isfinite(t.learn_on(st, ac.to(DEV), rw, ns, dn))
    ac[3] = A + 4
    with pytest.raises(IndexError):
        t.learn_on(st, ac.to(DEV), rw, ns, dn)
