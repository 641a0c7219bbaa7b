"""Implicit Quantile Network (last open item): network forward and IQNTrainer.learn on the HIP path
against the reference-generated golden (oracle/gen_golden.py:gen_iqn — upstream's live IQNNetwork and upstream's own
learn(), constructor bypassed, `get_q_values` bound to forward) and against the fp64 oracle (oracle/iqn_oracle.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, sub
from oracle import iqn_oracle as IO
from porl_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _golden():
    z = np.load(os.path.join(GOLDEN, "iqn_s9_a5.npz"), allow_pickle=False)
    return z, tuple(int(v) for v in z["meta"][:8])


def _trainer(z, S, A, E, H, B, NP, NPP, **kw):
    from porl_amd.train.iqn_trainer import IQNTrainer
    t = IQNTrainer(S, A, gamma=float(z["gamma"]), device=DEV, learning_rate=float(z["lr"]), batch_size=B,
                   kappa=float(z["kappa"]), embedding_dim=E, hidden_size=H, num_quantiles_n_prime_loss=NP,
                   num_quantiles_n_double_prime_loss=NPP, **kw)
    t.q_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init/").items()})
    t.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init_target/").items()})
    return t


def _np_sd(m):
    return {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


def test_network_forward_matches_reference_golden():
    z, (S, A, E, H, B, K, NP, NPP) = _golden()
    t = _trainer(z, S, A, E, H, B, NP, NPP)
    x, taus = torch.from_numpy(z["probe_x"]).to(DEV), torch.from_numpy(z["probe_taus"]).to(DEV)
    with torch.no_grad():
        got = t.q_network(x, taus)
        emb = t.q_network.get_quantile_embedding(taus)
    assert got.shape == (7, 5, A) and emb.shape == (7, 5, E)
    np.testing.assert_allclose(emb.cpu().numpy(), z["probe_embed"], atol=2e-5)     # cos of arguments up to pi * E: fp32 cos vs cos
    np.testing.assert_allclose(got.cpu().numpy(), z["probe_z"], atol=1e-5)
    with torch.no_grad():
        np.testing.assert_array_equal(t.q_network.get_q_values(x, taus).cpu().numpy(), got.cpu().numpy())


def test_seeded_construction_draws_the_reference_initial_weights():
    """Same module construction order as upstream: the golden's initial state_dict comes back from torch.manual_seed."""
    z, (S, A, E, H, *_rest) = _golden()
    from porl_amd.net.iqn_network import IQNNetwork
    torch.manual_seed(int(z["meta"][8]))
    net = IQNNetwork(S, A, E, H)
    for k, v in sub(z, "init/").items():
        np.testing.assert_array_equal(net.state_dict()[k].numpy(), v, err_msg=k)


def test_learn_matches_reference_golden():
    z, (S, A, E, H, B, K, NP, NPP) = _golden()
    t = _trainer(z, S, A, E, H, B, NP, NPP)
    for k in range(K):
        i = slice(k * B, (k + 1) * B)
        loss = t.learn_on(*(torch.from_numpy(z[n][i]) for n in ("states", "actions", "rewards", "next_states", "dones")),
                          taus_prime=torch.from_numpy(z["taus_prime"][k]), taus_double_prime=torch.from_numpy(z["taus_double_prime"][k]))
        np.testing.assert_allclose(loss, z["loss"][k], rtol=1e-5)
    for k, ref in sub(z, "final/").items():
        np.testing.assert_allclose(_np_sd(t.q_network)[k], ref, atol=1e-5, err_msg=k)
    # the target network is untouched by learn() and follows sync_target()
    for k, ref in sub(z, "init_target/").items():
        np.testing.assert_array_equal(_np_sd(t.target_network)[k], ref)
    t.sync_target()
    for k, v in _np_sd(t.q_network).items():
        np.testing.assert_array_equal(_np_sd(t.target_network)[k], v)


@pytest.mark.parametrize("B,H,E,NP,NPP,A", [(1024, 512, 64, 8, 8, 6), (37, 40, 10, 3, 5, 2), (1, 24, 8, 1, 1, 3)])
def test_gradients_and_loss_match_the_fp64_oracle(B, H, E, NP, NPP, A):
    """One step at the reference's default sizes (hidden 512, 64 cosine features) and at ragged ones: loss, the total
    gradient norm and every gradient against the oracle; then the parameters after the Adam step."""
    from porl_amd.train.iqn_trainer import IQNTrainer
    S = 13
    rng = np.random.default_rng(B + H)
    torch.manual_seed(5)
    t = IQNTrainer(S, A, gamma=0.95, device=DEV, learning_rate=1e-3, batch_size=B, kappa=0.5, embedding_dim=E, hidden_size=H,
                   num_quantiles_n_prime_loss=NP, num_quantiles_n_double_prime_loss=NPP)
    with torch.no_grad():
        t._target.flat.add_(0.05 * torch.randn_like(t._target.flat))
    P0, T0 = _np_sd(t.q_network), _np_sd(t.target_network)
    st, ns = rng.standard_normal((B, S)).astype(np.float32), rng.standard_normal((B, S)).astype(np.float32)
    ac = rng.integers(0, A, B)
    rw = (2.0 * rng.standard_normal(B)).astype(np.float32)
    dn = (rng.random(B) < 0.2).astype(np.float32)
    tp, tpp = rng.random((B, NP)).astype(np.float32), rng.random((B, NPP)).astype(np.float32)
    o = IO.IqnOracle(P0, T0, gamma=0.95, kappa=0.5, lr=1e-3)
    want = o.learn(st, ac, rw, ns, dn, tp, tpp)
    got = t.learn_on(*(torch.from_numpy(a) for a in (st, ac, rw, ns, dn)), taus_prime=torch.from_numpy(tp),
                     taus_double_prime=torch.from_numpy(tpp))
    np.testing.assert_allclose(got, want, rtol=1e-5)
    np.testing.assert_allclose(float(t.optimizer._clip[0]), o.grad_norm, rtol=1e-5)
    # gradients (as clipped; they stay in the flat buffer after the step): 1e-5 of each tensor's largest entry at the small
    # sizes.  At B = 1024, H = 512 the 4.7 M hidden activations include a dozen whose pre-activation is within fp32
    # rounding of zero: their ReLU masks differ between an fp32 and an fp64 evaluation, and each such flip moves every
    # entry of the layers below by ~1e-4 of the tensor's largest entry (measured: 5.6e-8 against 1.8e-5) — there the
    # bound is on the relative 2-norm of the difference.
    for p_, (k, _) in zip(t.q_network.parameters(), t.q_network.named_parameters()):
        g, want_g = p_.grad.cpu().numpy().astype(np.float64), o.G[k]
        if B * NP * H < 100_000:
            assert np.abs(g - want_g).max() <= 1e-5 * np.abs(want_g).max() + 1e-12, (k, np.abs(g - want_g).max(), np.abs(want_g).max())
        else:
            assert np.linalg.norm(g - want_g) <= 5e-3 * np.linalg.norm(want_g), (k, np.linalg.norm(g - want_g), np.linalg.norm(want_g))
    # parameters after the FIRST Adam step: delta = lr * g / (|g| + 1e-8), so entries with |g| of the order of 1e-8 turn
    # rounding noise of g into a visible difference (sensitivity lr / 4e-8 per unit of g) — bounded by 2 lr, rare elsewhere
    P1 = _np_sd(t.q_network)
    lr = 1e-3
    for k in IO.NAMES:
        d = np.abs(P1[k] - o.P[k])
        sensitive = np.abs(o.G[k]) < (1e-6 if B * NP * H < 100_000 else 2e-5)
        assert d.max() <= 2 * lr * 1.001 and (d[~sensitive] <= 2e-6).all(), (k, d.max(), d[~sensitive].max())


def test_grad_clip_is_torchs_clip_grad_norm():
    g = torch.Generator().manual_seed(3)
    for n, max_norm in ((100_003, 10.0), (100_003, 1e6), (5, 0.3), (1, 2.0)):
        x = torch.randn(n, generator=g) * 0.7
        p = torch.nn.Parameter(torch.zeros(n))
        p.grad = x.clone()
        total = torch.nn.utils.clip_grad_norm_([p], max_norm)
        gd = x.to(DEV)
        nc, ws = torch.zeros(2, device=DEV), torch.zeros(256, dtype=torch.float64, device=DEV)
        N.check(N.lib().porl_grad_clip(N.ptr(gd), n, max_norm, N.ptr(nc), N.ptr(ws), N.current_stream_ptr(gd)), "porl_grad_clip")
        np.testing.assert_allclose(float(nc[0]), float(total), rtol=2e-6)
        np.testing.assert_allclose(gd.cpu().numpy(), p.grad.numpy(), rtol=3e-6, atol=0)
        if float(total) <= max_norm:
            assert float(nc[1]) == 1.0 and torch.equal(gd.cpu(), x)


def test_glue_kernels_against_torch_expressions():
    """The three autograd pieces (csrc/iqn.hpp) against the torch expressions of iqn_network.py:58-62 and
    iqn_trainer.py:101-103 evaluated on the same device, forward and backward."""
    from porl_amd.net.iqn_network import _Hadamard, SelectAction, cos_embed
    g = torch.Generator().manual_seed(11)
    for B, n_tau, H, A in ((33, 5, 48, 4), (8, 1, 30, 7), (2, 9, 4, 2)):
        feat = torch.randn(B, H, generator=g).to(DEV).requires_grad_(True)
        emb = torch.randn(B * n_tau, H, generator=g).to(DEV).requires_grad_(True)
        w = torch.randn(B * n_tau, H, generator=g).to(DEV)
        out = _Hadamard.apply(feat, emb, n_tau)
        ref = (feat.unsqueeze(1).expand(-1, n_tau, -1) * emb.view(B, n_tau, H)).reshape(B * n_tau, H)
        assert torch.equal(out, ref)
        gf, ge = torch.autograd.grad(out, (feat, emb), w)
        rf, re = torch.autograd.grad(ref, (feat, emb), w)
        assert torch.equal(ge, re)
        np.testing.assert_allclose(gf.cpu().numpy(), rf.cpu().numpy(), rtol=1e-5, atol=1e-6)
        z = torch.randn(B, n_tau, A, generator=g).to(DEV).requires_grad_(True)
        act = torch.randint(0, A, (B,), generator=g).to(DEV)
        sel = SelectAction.apply(z, act)
        ref = z.gather(2, act.view(-1, 1, 1).expand(-1, n_tau, 1)).squeeze(2)
        assert torch.equal(sel, ref)
        w2 = torch.randn(B, n_tau, generator=g).to(DEV)
        assert torch.equal(torch.autograd.grad(sel, z, w2)[0], torch.autograd.grad(ref, z, w2)[0])
        taus = torch.rand(B, n_tau, generator=g)
        ce = cos_embed(taus.to(DEV), 16).cpu().numpy()
        np.testing.assert_allclose(ce, IO.cos_embed(taus.numpy().astype(np.float64), 16).reshape(-1, 16), atol=2e-5)


def test_target_kernel_picks_the_first_maximum_and_masks_terminal_rows():
    B, n_tau, A = 6, 4, 3
    zo = torch.zeros(B, n_tau, A)
    zo[0, :, 2] = 1.0
    zo[1, :, 1] = 1.0
    zo[1, :, 2] = 1.0          # tie between actions 1 and 2: torch.argmax takes the first
    zo[2, 0, 0], zo[2, 1, 1] = 4.0, 5.0          # the choice is on the tau-MEAN, not on any single fraction
    zt = torch.arange(B * n_tau * A, dtype=torch.float32).view(B, n_tau, A)
    r = torch.arange(B, dtype=torch.float32)
    d = torch.tensor([0, 0, 0, 1, 0, 1], dtype=torch.float32)
    td, na = torch.empty(B, n_tau, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    args = [a.to(DEV).contiguous() for a in (zo, zt, r, d)]
    N.check(N.lib().porl_iqn_target(*(N.ptr(a) for a in args), 0.5, B, n_tau, A, N.ptr(td), N.ptr(na),
                                    N.current_stream_ptr(td)), "porl_iqn_target")
    want_a = zo.mean(1).argmax(1)
    assert torch.equal(na.cpu(), want_a) and want_a.tolist()[:3] == [2, 1, 1]
    want = r[:, None] + 0.5 * zt[torch.arange(B), :, want_a] * (1 - d[:, None])
    assert torch.equal(td.cpu(), want)


def test_bad_action_raises_like_the_reference_gather():
    z, (S, A, E, H, B, K, NP, NPP) = _golden()
    t = _trainer(z, S, A, E, H, B, NP, NPP)
    before = _np_sd(t.q_network)
    bad = torch.from_numpy(z["actions"][:B].copy())
    bad[3] = A
    with pytest.raises(IndexError):
        t.learn_on(torch.from_numpy(z["states"][:B]), bad, torch.from_numpy(z["rewards"][:B]),
                   torch.from_numpy(z["next_states"][:B]), torch.from_numpy(z["dones"][:B]),
                   taus_prime=torch.from_numpy(z["taus_prime"][0]), taus_double_prime=torch.from_numpy(z["taus_double_prime"][0]))


def test_learn_from_the_replay_buffer_and_act():
    z, (S, A, E, H, B, K, NP, NPP) = _golden()
    t = _trainer(z, S, A, E, H, B, NP, NPP)
    for i in range(3 * B):
        t.replay_buffer.push(z["states"][i], int(z["actions"][i]), float(z["rewards"][i]), z["next_states"][i], bool(z["dones"][i]))
    np.random.seed(0)
    torch.manual_seed(0)
    losses = t.train_offline(num_iterations=6)
    assert len(losses) == 6 and all(np.isfinite(losses)) and t.optimizer.step_count == 6
    t.epsilon = 0.0
    a = t.select_action(z["states"][0])
    assert 0 <= a < A
    # CPU device is refused (no fallback)
    from porl_amd.train.iqn_trainer import IQNTrainer
    with pytest.raises(N.NativeError):
        IQNTrainer(S, A, gamma=0.9, device=torch.device("cpu"))
