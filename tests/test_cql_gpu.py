"""CQL device step (through CQLTrainer and the C ABI) against the reference's goldens and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from oracle.por_oracle import CqlOracle
from porl_amd.util.synth import make_discrete_transitions

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _trainer(S, A, B, seed, **kw):
    from porl_amd.train.cql_trainer import CQLTrainer
    torch.manual_seed(seed)
    return CQLTrainer(state_size=S, action_size=A, gamma=kw.pop("gamma", 0.99), device=DEV, batch_size=B, **kw)


def _np_sd(mod):
    return {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}


@pytest.mark.parametrize("name", ["cql_s60_a10_b64", "cql_s8_a4_b256"])
def test_cql_learn_matches_reference_golden(name):
    z, meta = load_golden(name)
    S, A, B, K, N = (int(meta[k]) for k in ("S", "A", "B", "K", "N"))
    t = _trainer(S, A, B, int(meta["seed_model"]), alpha=meta["alpha"])
    for k, v in sub(z, "init/").items():
        assert np.array_equal(_np_sd(t.q_network)[k], v), k
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=int(meta["seed_data"]))
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    np.random.seed(int(meta["seed_np"]))                      # the reference's index stream
    for k in range(K):
        loss = t.learn()
        np.testing.assert_allclose(loss, z["loss"][k], rtol=1e-5)
        if (k + 1) % int(meta["sync_every"]) == 0:
            t.sync_target()
    for k, v in sub(z, "final/").items():
        np.testing.assert_allclose(_np_sd(t.q_network)[k], v, atol=1e-5, err_msg=k)
    for k, v in sub(z, "final_target/").items():
        np.testing.assert_allclose(_np_sd(t.target_network)[k], v, atol=1e-5, err_msg=k)
    idx = z["indices"][0]
    pen = t.compute_cql_penalty(torch.from_numpy(st[idx]).to(DEV), torch.from_numpy(ac[idx]).to(DEV))
    np.testing.assert_allclose(float(pen), float(z["penalty_final_on_batch0"]), atol=2e-6)
    am = sub(z, "adam/")
    sd = t.optimizer.state_dict()
    names = [n for n, _ in t.q_network.named_parameters()]
    for i, n in enumerate(names):
        np.testing.assert_allclose(sd["state"][i]["exp_avg"].cpu().numpy(), am[n + ".exp_avg"], atol=1e-7, rtol=1e-4)


def test_cql_config3_vs_oracle():
    """BASELINE config 3 shapes: S=60, A=10, B=4096, Q-net 64-128-64."""
    S, A, B, N = 60, 10, 4096, 20000
    t = _trainer(S, A, B, 0)
    o = CqlOracle(_np_sd(t.q_network), A, lr=5e-4)
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=9)
    rng = np.random.default_rng(0)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    for k in range(4):
        idx = rng.choice(N, B, replace=False)
        loss = t.learn_on(dev(st[idx]), dev(ac[idx]), dev(rw[idx]), dev(ns[idx]), dev(dn[idx]))
        np.testing.assert_allclose(loss, o.learn(st[idx], ac[idx], rw[idx], ns[idx], dn[idx]), rtol=1e-5)
        if k == 1:
            t.sync_target(); o.sync_target()
    for k, v in o.Q.items():
        np.testing.assert_allclose(_np_sd(t.q_network)[k], v, atol=1e-5, err_msg=k)
    x = dev(st[:33])
    q_ref = o.Q
    from oracle.por_oracle import qnet_forward
    np.testing.assert_allclose(t.q_network(x).cpu().numpy(), qnet_forward(o.Q, "", st[:33], 4)[0], atol=2e-6)
    np.testing.assert_allclose(t.target_network(x).cpu().numpy(), qnet_forward(o.T, "", st[:33], 4)[0], atol=2e-6)
    assert t.get_action(st[0]) == int(np.argmax(qnet_forward(o.Q, "", st[:1], 4)[0]))


def test_replay_buffer_sample_matches_reference_stream():
    """Drop-in ReplayBuffer: ring semantics, numpy index stream and dtypes of the reference (golden)."""
    from porl_amd.buffer.replay_buffer import ReplayBuffer
    z, meta = load_golden("replay_ring")
    N, cap, S, B, K = (int(meta[k]) for k in ("N", "cap", "S", "B", "K"))
    rb = ReplayBuffer(cap, (S,), DEV)
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, 4, seed=5)
    for i in range(N):
        rb.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    assert len(rb) == int(z["size"]) and rb.position == int(z["position"])
    np.random.seed(int(meta["seed_np"]))
    for k in range(K):
        s, a, r, n, d = rb.sample(B)
        assert [str(t.dtype) for t in (s, a, r, n, d)] == [str(x) for x in z["dtype_names"]]
        for got, key in ((s, "s"), (a, "a"), (r, "r"), (n, "n"), (d, "d")):
            assert np.array_equal(got.cpu().numpy(), z[f"{key}{k}"]), key
    with pytest.raises(ValueError):
        rb.sample(len(rb) + 1)
    # pushes after the mirror exists are reflected
    rb.push(st[0] + 1, 3, 2.5, ns[0], True)
    s, a, r, n, d = rb.sample_at(np.array([(rb.position - 1) % cap]))
    assert np.array_equal(s.cpu().numpy()[0], st[0] + 1) and int(a) == 3 and float(r) == 2.5 and float(d) == 1.0


@pytest.mark.parametrize("S,A,B,hidden", [(60, 10, 4096, None), (8, 4, 50, None)])
def test_one_launch_step_matches_the_multi_launch_path(S, A, B, hidden):
    """csrc/qnet_fused.hpp (32 rows per block, everything in LDS) against the grouped-GEMM path on the same
    minibatches, incl. a ragged batch and odd widths: same losses (rtol 1e-6) and parameters (2e-6) after 3 steps."""
    from porl_amd import engine as E
    kw = {} if hidden is None else {"hidden_layers": hidden}
    st, ac, rw, ns, dn = make_discrete_transitions(3 * B, S, A, seed=21)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    results = []
    for fused in (1, 0):
        try:
            E.tune_set("qnet_fused", fused)
            t = _trainer(S, A, B, 4, **kw)
            assert t._engine.fused == bool(fused)
            losses = []
            for k in range(3):
                sl = slice(k * B, (k + 1) * B)
                losses.append(t.learn_on(dev(st[sl]), dev(ac[sl]), dev(rw[sl]), dev(ns[sl]), dev(dn[sl])))
            results.append((losses, _np_sd(t.q_network)))
        finally:
            E.tune_set("qnet_fused", 1)
    np.testing.assert_allclose(results[0][0], results[1][0], rtol=2e-6)
    for k, v in results[1][1].items():
        np.testing.assert_allclose(results[0][1][k], v, atol=2e-6, err_msg=k)


def test_learn_device_sampled_gathers_inside_the_step_kernel():
    """`learn_device_sampled` (indices drawn on the device, rows gathered by the step kernel itself) equals
    `learn_on` applied to the gathered minibatch, bit for bit."""
    from porl_amd import engine as E
    S, A, B, N = 60, 10, 512, 5000
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=3)
    ta, tb = _trainer(S, A, B, 7), _trainer(S, A, B, 7)
    for t in (ta, tb):
        rb = t.replay_buffer
        rb.states[:N], rb.actions[:N], rb.rewards[:N], rb.next_states[:N], rb.dones[:N] = st, ac, rw, ns, dn
        rb.size, rb.position = N, 0
    for k in range(3):
        la = ta.learn_device_sampled(seed=5)
        tb.replay_buffer._sync_mirror()
        idx = E.sample_indices(N, B, 5, k, device=DEV)
        lb = tb.learn_on(*tb.replay_buffer.gather_device(idx))
        assert la == lb
    for (k, a), (_, b) in zip(ta.q_network.state_dict().items(), tb.q_network.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("S,A,B,hidden", [(17, 3, 33, (32, 96)), (64, 32, 96, (128, 128, 64)), (5, 2, 1, (7,))])
def test_one_launch_gradients_on_odd_shapes(S, A, B, hidden):
    """Engine level: gradients and loss statistics of the one-launch kernel against the grouped-GEMM path for
    widths that are not multiples of 32, the widest supported network, and a single-row batch."""
    from porl_amd import engine as E
    from porl_amd.train.cql_trainer import QnetEngine
    rng = np.random.default_rng(S * 1000 + A)
    eng = QnetEngine(S, A, hidden, max(B, 64), DEV)
    assert eng.fused
    for flat in (eng.params, eng.params_tgt):           # through the parameter views: the images' padding stays zero
        for v in eng.views(flat):
            v.copy_(torch.from_numpy(rng.uniform(-0.3, 0.3, tuple(v.shape)).astype(np.float32)))
    st, ac, rw, ns, dn = make_discrete_transitions(B, S, A, seed=8)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    hp = eng.hyper(0.99, 0.7, 1.0 / B, 1, 5e-4)
    out = []
    for fused in (1, 0):
        try:
            E.tune_set("qnet_fused", fused)
            eng.load_batch(dev(st), dev(ac), dev(rw), dev(ns), dev(dn))
            eng.grads.zero_()
            eng.cql_backward(hp)
            out.append((eng.grads.cpu().numpy().copy(), eng.stats[:3].cpu().numpy().copy()))
        finally:
            E.tune_set("qnet_fused", 1)
    scale = np.abs(out[1][0]).max()
    assert np.abs(out[0][0] - out[1][0]).max() <= 2e-6 * max(scale, 1.0)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=2e-6, atol=5e-8)   # (penalty = lse - ln A - q: O(1) terms cancel)


@pytest.mark.parametrize("S,A,B,hidden", [(60, 10, 4096, (64, 128, 64)), (17, 3, 33, (32, 96)), (5, 2, 1, (7,)), (12, 5, 100, (64,))])
@pytest.mark.parametrize("variant", ["cql", "double_dqn", "bcq_mask"])
def test_two_group_step_kernel_is_bit_identical_to_the_one_group_kernel(S, A, B, hidden, variant):
    """csrc/qnet_fused.hpp: the 512-thread kernel (target net || online net, dZ chain || dW tiles on two wave groups)
    performs the same arithmetic in the same summation orders as the 256-thread kernel: three learn steps on indexed
    replay rows give identical parameters, Adam moments, loss statistics and |TD| write-backs.  Its 16-row form
    (v_mfma_f32_16x16x4_f32, twice as many blocks, other partial-sum grouping) agrees to rounding."""
    from porl_amd import _native as NN
    from porl_amd import engine as E
    from porl_amd.train.cql_trainer import QnetEngine
    rng = np.random.default_rng(S * 100 + A)
    N = 3 * B + 7
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=31)
    dev = lambda x: torch.from_numpy(x).to(DEV)
    st_d, ac_d, rw_d, ns_d, dn_d = dev(st), dev(ac).long(), dev(rw), dev(ns), dev(dn)
    init = [rng.uniform(-0.3, 0.3, 200_000).astype(np.float32) for _ in range(2)]
    mask = torch.from_numpy((rng.uniform(size=(B, A)) < 0.6).astype(np.float32)).to(DEV)
    idxs = [torch.from_numpy(rng.permutation(N)[:B].astype(np.int64)).to(DEV) for _ in range(3)]
    out = []
    for two, rows16 in ((1, 0), (0, 0), (1, 1)):
        try:
            E.tune_set("qnet_two_groups", two)
            E.tune_set("qnet_rows16", rows16)
            eng = QnetEngine(S, A, hidden, max(B, 64), DEV)
            assert eng.fused
            for flat, src in zip((eng.params, eng.params_tgt), init):
                o = 0
                for v in eng.views(flat):
                    v.copy_(torch.from_numpy(src[o:o + v.numel()].reshape(tuple(v.shape))))
                    o += v.numel()
            td_abs = torch.zeros(B, device=DEV)
            var = None
            if variant == "double_dqn":
                var = NN.QnetVariant(1, None, None, NN.ptr(td_abs), None, 0)
            elif variant == "bcq_mask":
                var = NN.QnetVariant(0, None, None, NN.ptr(td_abs), NN.ptr(mask), 0)
            for k in range(3):
                hp = eng.hyper(0.99, 0.7, 1.0 / B, k + 1, 5e-4)
                eng.learn_indexed(hp, st_d, ac_d, rw_d, ns_d, dn_d, idxs[k], variant=var)
            torch.cuda.synchronize()
            out.append([x.clone() for x in (eng.params, eng.adam_m, eng.adam_v, eng.grads, eng.stats[:3], td_abs)])
        finally:
            E.tune_set("qnet_two_groups", 1)
            E.tune_set("qnet_rows16", 1)                       # the default since round 3
    names = ("params", "adam_m", "adam_v", "grads", "stats", "td_abs")
    for a, b, what in zip(out[0], out[1], names):
        assert torch.equal(a, b), what
    for a, c, what in zip(out[0], out[2], names):
        scale = max(float(a.abs().max()), 1e-3)
        assert float((a - c).abs().max()) <= 3e-6 * max(scale, 1.0) + 2e-6 * scale, what


def test_wide_networks_keep_the_multi_launch_path():
    from porl_amd.train.cql_trainer import QnetEngine
    assert not QnetEngine(60, 10, (64, 256, 64), 64, DEV).fused


@pytest.fixture
def qnet_path(request):
    """"fused" = the one-launch step kernel; "general" = the multi-launch path every network wider than 128 takes
    (gather launch + grouped GEMMs + the variant-aware loss head), forced here on the SAME small networks so that it is
    pinned by the same reference goldens."""
    from porl_amd import engine as E
    E.tune_set("qnet_fused", 1 if request.param == "fused" else 0)
    yield request.param
    E.tune_set("qnet_fused", 1)


@pytest.mark.parametrize("qnet_path", ["fused", "general"], indirect=True)
@pytest.mark.parametrize("name", ["dqn_s10_a6", "ddqn_s10_a6"])
def test_dqn_and_double_dqn_learn_match_reference_golden(name, qnet_path):
    """DQNTrainer.learn (dqn_trainer.py:93-118) and DDQNTrainer.learn (ddqn_trainer.py:58-99): five steps under the
    reference's numpy index stream, from the reference's initial online/target parameters."""
    from porl_amd.train.dqn_trainer import DDQNTrainer, DQNTrainer
    z, _ = load_golden(name)
    S, A, B, K, N, seed_model, seed_data, seed_np, double = (int(v) for v in z["meta"])
    cls = DDQNTrainer if double else DQNTrainer
    t = cls(S, A, float(z["gamma"]), device=DEV, batch_size=B)
    assert t._engine.fused == (qnet_path == "fused")
    t.q_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init/").items()})
    t.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init_target/").items()})
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    t.replay_buffer = type(t.replay_buffer)(N, (S,), DEV)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    np.random.seed(seed_np)
    for k in range(K):
        np.testing.assert_allclose(t.learn(), z["loss"][k], rtol=2e-5)
    got = _np_sd(t.q_network)
    for k, v in sub(z, "final/").items():
        np.testing.assert_allclose(got[k], v, atol=1e-5, err_msg=k)


def test_double_dqn_on_a_wide_network_equals_the_oracle_step():
    """network_hidden_sizes beyond the one-launch kernel's 128 columns (dqn_trainer.py:66-69 lets the caller pick any):
    DDQNTrainer takes the multi-launch path by itself; one step against a direct fp64 evaluation of ddqn_trainer.py:58-99
    from the same parameters (loss and the updated output-layer bias, whose gradient is the column sum of dL/dQ)."""
    from porl_amd.net.q_network import QNetwork
    from porl_amd.train.dqn_trainer import DDQNTrainer
    S, A, B = 12, 7, 96
    torch.manual_seed(3)
    t = DDQNTrainer(S, A, 0.97, device=DEV, batch_size=B, network=lambda s, a: QNetwork(s, a, [48, 200, 160]))
    assert not t._engine.fused
    g = torch.Generator().manual_seed(5)
    st, ns = torch.randn(B, S, generator=g), torch.randn(B, S, generator=g)
    ac = torch.randint(0, A, (B,), generator=g)
    rw, dn = torch.randn(B, generator=g), (torch.rand(B, generator=g) < 0.2).float()
    P = {k: v.detach().cpu().double() for k, v in t.q_network.state_dict().items()}
    T = {k: v.detach().cpu().double() for k, v in t.target_network.state_dict().items()}

    def fwd(p, x):
        h = x.double()
        keys = sorted({k.rsplit(".", 1)[0] for k in p}, key=lambda s: int(s.split(".")[-1]))
        for i, k in enumerate(keys):
            h = h @ p[k + ".weight"].T + p[k + ".bias"]
            if i < len(keys) - 1:
                h = torch.relu(h)
        return h
    q, qn_t, qn_o = fwd(P, st), fwd(T, ns), fwd(P, ns)
    qa = q[torch.arange(B), ac]
    y = rw.double() + 0.97 * qn_t[torch.arange(B), qn_o.argmax(1)] * (1 - dn.double())
    want = float(((qa - y) ** 2).mean())
    loss = t.learn_on(st.to(DEV), ac.to(DEV), rw.to(DEV), ns.to(DEV), dn.to(DEV))
    np.testing.assert_allclose(loss, want, rtol=2e-5)
    # first Adam step: every parameter moves by lr * sign(grad) (up to eps); check the sign pattern of the output bias
    gb = torch.zeros(A, dtype=torch.float64).index_add_(0, ac, 2 * (qa - y) / B)
    last = sorted(P, key=lambda k: int(k.split(".")[-2]))[-1].rsplit(".", 1)[0] + ".bias"
    moved = t.q_network.state_dict()[last].cpu().double() - P[last]
    big = gb.abs() > 1e-6
    assert torch.equal(torch.sign(moved[big]), -torch.sign(gb[big]))
    np.testing.assert_allclose(moved[big].abs().numpy(), 5e-4, rtol=1e-3)


@pytest.mark.parametrize("qnet_path", ["fused", "general"], indirect=True)
def test_bcq_pretrain_and_learn_match_reference_golden(qnet_path):
    """Discrete BCQ (src/porl/policy/bcq.py): six cross-entropy epochs of the behaviour policy (:23-47), then five
    bcq_learn steps (:50-86) whose bootstrap action is the masked argmax of the target network — under the reference's
    numpy index stream, from the reference's initial Q / target / behaviour parameters."""
    from porl_amd.policy.bcq import bcq_behavior_pretrain, bcq_learn
    from porl_amd.train.bcq_trainer import BCQTrainer
    z, _ = load_golden("bcq_s10_a6")
    S, A, B, K, N, seed_model, seed_data, seed_np, KP = (int(v) for v in z["meta"])
    torch.manual_seed(seed_model)
    t = BCQTrainer(S, A, float(z["gamma"]), device=DEV, batch_size=B, num_epochs=KP, threshold=float(z["threshold"]))
    assert list(t.behavior_policy.state_dict().keys()) == list(sub(z, "init_behavior/").keys())
    t.behavior_policy.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init_behavior/").items()})
    t.q_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init/").items()})
    t.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init_target/").items()})
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    t.replay_buffer = type(t.replay_buffer)(N, (S,), DEV)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    np.random.seed(seed_np)
    np.testing.assert_allclose(bcq_behavior_pretrain(t), z["ce_loss"], rtol=2e-5)
    got = _np_sd(t.behavior_policy)
    for k, v in sub(z, "behavior_after/").items():
        np.testing.assert_allclose(got[k], v, atol=1e-5, err_msg=k)
    # mask / probabilities of the first learn batch (peek without consuming numpy's stream)
    state = np.random.get_state()
    nxt = t.replay_buffer.sample(B)[3]
    np.random.set_state(state)
    np.testing.assert_allclose(t.behavior_policy(nxt).cpu().numpy(), z["probs0"], atol=2e-6)
    np.testing.assert_array_equal(t.behavior_policy.sample(nxt, t.threshold).cpu().numpy(), z["mask0"])
    assert 0.2 < z["mask0"].mean() < 0.8                      # the mask really selects
    for k in range(K):
        np.testing.assert_allclose(bcq_learn(t), z["loss"][k], rtol=2e-5)
    got = _np_sd(t.q_network)
    for k, v in sub(z, "final/").items():
        np.testing.assert_allclose(got[k], v, atol=1e-5, err_msg=k)


def test_bcq_mask_with_no_allowed_action_falls_back_to_the_first_action():
    """(mask - 1) * 1e10 drowns every Q value when no action passes the threshold: all entries tie at -1e10 and
    torch.argmax returns index 0 (bcq.py:68-73).  threshold = 1.0 forces that case."""
    from porl_amd.policy.bcq import bcq_learn
    from porl_amd.train.bcq_trainer import BCQTrainer
    from porl_amd.train.dqn_trainer import DQNTrainer
    S, A, B, N = 8, 5, 32, 64
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=2)
    torch.manual_seed(0)
    t = BCQTrainer(S, A, 0.9, device=DEV, batch_size=B, threshold=1.0)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    np.random.seed(0)
    idx = np.random.choice(N, B, replace=False)
    np.random.seed(0)
    q0 = _np_sd(t.q_network)
    tq = t.target_network(torch.from_numpy(ns[idx]).to(DEV)).cpu().numpy()
    q = t.q_network(torch.from_numpy(st[idx]).to(DEV)).cpu().numpy()
    loss = bcq_learn(t)
    y = rw[idx] + 0.9 * tq[:, 0] * (1 - dn[idx])
    want = np.mean((q[np.arange(B), ac[idx]] - y) ** 2)
    np.testing.assert_allclose(loss, want, rtol=2e-5)


@pytest.mark.parametrize("qnet_path", ["fused", "general"], indirect=True)
def test_dueling_double_dqn_learn_matches_reference_golden(qnet_path):
    """DDDQNTrainer.learn (dddqn_trainer.py:59-103) with DuelingQNetwork pairs (q_network.py:33-68): five steps under the
    reference's numpy index stream from the reference's initial online / target parameters.  The heads ride on the engine
    as a composed output layer; value.* / advantage.* / model.* must all follow the reference."""
    from porl_amd.net.q_network import DuelingQNetwork
    from porl_amd.train.dddqn_trainer import DDDQNTrainer
    z, _ = load_golden("dddqn_s10_a6")
    S, A, B, K, N, seed_model, seed_data, seed_np = (int(v) for v in z["meta"])
    torch.manual_seed(seed_model)
    t = DDDQNTrainer(S, A, float(z["gamma"]), device=DEV, batch_size=B)
    assert isinstance(t.q_network, DuelingQNetwork) and list(t.q_network.state_dict().keys()) == list(sub(z, "init/").keys())
    for k, v in sub(z, "init/").items():                          # same seed: the reference's own initialisation
        np.testing.assert_array_equal(t.q_network.state_dict()[k].cpu().numpy(), v, err_msg=k)
    t.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in sub(z, "init_target/").items()})
    x = torch.from_numpy(z["probe_x"]).to(DEV)
    np.testing.assert_allclose(t.q_network(x).cpu().numpy(), z["probe_q0"], atol=2e-6)
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    t.replay_buffer = type(t.replay_buffer)(N, (S,), DEV)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    np.random.seed(seed_np)
    for k in range(K):
        np.testing.assert_allclose(t.learn(), z["loss"][k], rtol=2e-5)
    got = _np_sd(t.q_network)
    for k, v in sub(z, "final/").items():
        np.testing.assert_allclose(got[k], v, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(t.q_network(x).cpu().numpy(), z["probe_q"], atol=5e-6)
    # optimizer state in torch.optim.Adam's format, one entry per q_network parameter in its order
    osd = t.optimizer.state_dict()
    assert len(osd["state"]) == len(list(t.q_network.parameters())) == 10
    for i, p_ in enumerate(t.q_network.parameters()):
        assert tuple(osd["state"][i]["exp_avg"].shape) == tuple(p_.shape) and float(osd["state"][i]["step"]) == K
    # hard target sync copies heads and composed layer
    t.sync_target()
    np.testing.assert_array_equal(t.target_network(x).cpu().numpy(), t.q_network(x).cpu().numpy())
    for k, v in t.q_network.state_dict().items():
        assert torch.equal(v, t.target_network.state_dict()[k]), k
