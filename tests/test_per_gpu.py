"""Prioritized replay with the sum tree on the device against the reference's golden run
(tests/golden/per_cap300.npz from /root/reference/src/porl/buffer/prioritized_replay_buffer.py via
oracle/gen_golden.py): same tree indices under the same `random` stream, same importance weights, same tree."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _filled(z):
    from porl_amd.buffer.prioritized_replay_buffer import PrioritizedReplayBuffer
    cap, N, S, B, seed = (int(v) for v in z["meta"])
    buf = PrioritizedReplayBuffer(cap, alpha=0.6, beta_start=0.4, beta_frames=1000, device=DEV)
    for i in range(N):
        buf.add(z["td"][i], z["st"][i], int(z["ac"][i]), float(z["rw"][i]), z["ns"][i], float(z["dn"][i]))
    return buf, cap, N, S, B, seed


def test_sampling_weights_and_write_back_match_the_reference():
    z, _ = load_golden("per_cap300")
    buf, cap, N, S, B, seed = _filled(z)
    assert len(buf) == cap
    random.seed(seed)
    for k in range(2):
        s, a, r, n2, d, w, idxs = buf.sample(B)
        assert np.array_equal(idxs.cpu().numpy(), z[f"idx{k}"])
        np.testing.assert_allclose(w.cpu().numpy(), z[f"w{k}"], rtol=2e-7)
        assert np.array_equal(s.cpu().numpy(), z[f"s{k}"]) and np.array_equal(a.cpu().numpy(), z[f"a{k}"])
        assert np.array_equal(r.cpu().numpy(), z[f"r{k}"])
    buf.update_priorities(list(z["upd_idx"]), z["upd_td"])          # contains one leaf twice: the later value wins
    s, a, r, n2, d, w, idxs = buf.sample(B)
    assert np.array_equal(idxs.cpu().numpy(), z["idx2"])
    np.testing.assert_allclose(w.cpu().numpy(), z["w2"], rtol=2e-7)
    assert np.array_equal(s.cpu().numpy(), z["s2"])
    np.testing.assert_allclose(buf.tree.cpu().numpy(), z["tree_after"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(buf.total_priority(), float(z["total"]), rtol=1e-13)
    assert abs(float(buf.beta) - float(z["beta"])) < 1e-15


def test_tree_invariants_at_scale():
    """capacity 100 000 (the reference trainer's size): every internal node equals the sum of its children bit for
    bit, sampled leaves carry positive priority, and frequencies follow the priorities."""
    from porl_amd.buffer.prioritized_replay_buffer import PrioritizedReplayBuffer
    cap, S = 100_000, 4
    buf = PrioritizedReplayBuffer(cap, device=DEV, state_shape=(S,))
    rng = np.random.default_rng(0)
    buf._alloc(np.zeros(S, np.float32))
    buf.n_entries = cap
    td = torch.from_numpy(rng.uniform(0.01, 1.0, size=cap)).to(DEV)
    td[:10] = 50.0                                                    # ten heavy leaves
    idx = torch.arange(cap, device=DEV) + (cap - 1)
    for a in range(0, cap, 4096):
        buf._update(idx[a:a + 4096].contiguous(), td[a:a + 4096].contiguous())
    t = buf.tree.cpu().numpy()
    inner = np.arange(cap - 1)
    assert np.array_equal(t[inner], t[2 * inner + 1] + t[2 * inner + 2])
    random.seed(1)
    hits = np.zeros(cap, dtype=np.int64)
    for _ in range(20):
        *_, w, idxs = buf.sample(4096)
        slots = (idxs - (cap - 1)).cpu().numpy()
        assert slots.min() >= 0 and slots.max() < cap
        np.add.at(hits, slots, 1)
        assert float(w.max()) == 1.0 and float(w.min()) > 0
    p = t[cap - 1:] / t[0]
    expect_heavy = 20 * 4096 * p[:10].sum()
    assert abs(hits[:10].sum() - expect_heavy) < 6 * np.sqrt(expect_heavy)


@pytest.fixture
def qnet_path(request):
    from porl_amd import engine as E
    E.tune_set("qnet_fused", 1 if request.param == "fused" else 0)
    yield request.param
    E.tune_set("qnet_fused", 1)


@pytest.mark.parametrize("qnet_path", ["fused", "general"], indirect=True)
def test_per_trainer_learn_matches_reference_golden(qnet_path):
    """PERTrainer.learn (dqn_per_trainer.py:67-123): Double-DQN target, the reference's (B,1)x(B,) weighted loss,
    Adam, priority write-back — four steps under the same `random` stream; on the one-launch kernel and on the
    multi-launch path wide networks take."""
    from conftest import sub
    from porl_amd.train.dqn_per_trainer import PERTrainer
    from porl_amd.util.synth import make_discrete_transitions
    z, _ = load_golden("per_trainer_s12_a5")
    S, A, B, K, N, cap, seed_model, seed_data, seed_rand = (int(v) for v in z["meta"])
    t = PERTrainer(S, A, float(z["gamma"]), device=DEV, batch_size=B, capacity=cap)
    t.memory.beta_frames = 1000
    t.q_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init/").items()})
    t.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init_target/").items()})
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    for i in range(N):
        t.memory.add(1.0, st[i], int(ac[i]), float(rw[i]), ns[i], float(dn[i]))
    random.seed(seed_rand)
    for k in range(K):
        np.testing.assert_allclose(t.learn(), z["loss"][k], rtol=2e-5)
    got = {k: v.detach().cpu().numpy() for k, v in t.q_network.state_dict().items()}
    for k, v in sub(z, "final/").items():
        np.testing.assert_allclose(got[k], v, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(t.memory.tree.cpu().numpy(), z["tree_after"], rtol=2e-5, atol=5e-6)   # (|td|+eps)^0.6 of fp32 errors near 0


def test_per_sample_weighting_changes_the_step():
    from porl_amd.train.dqn_per_trainer import PERTrainer
    from porl_amd.util.synth import make_discrete_transitions
    S, A, B, N = 12, 5, 64, 300
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=1)
    losses = []
    for flag in (False, True):
        torch.manual_seed(0)
        t = PERTrainer(S, A, 0.99, device=DEV, batch_size=B, capacity=512, per_sample_weights=flag)
        for i in range(N):
            t.memory.add(0.1 + (i % 7), st[i], int(ac[i]), float(rw[i]), ns[i], float(dn[i]))
        random.seed(3)
        losses.append([t.learn() for _ in range(2)])
    assert np.isfinite(losses).all() and abs(losses[0][0] - losses[1][0]) > 1e-6
