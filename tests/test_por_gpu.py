"""End-to-end parity of the HIP update step (through the agent classes and the C ABI) against the
numpy oracle and the reference-generated golden vectors.  GPU only.

Tolerances (fp32, BASELINE.json north_star "within 1e-5"): losses rtol 1e-5 (g_loss ~ 80, 1 ulp =
7.6e-6 absolute); parameters max-abs 1e-5 after K steps.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import checksum, load_golden, sub
from oracle.por_oracle import PorOracle, sorl_oracle
from porl_amd.util.synth import make_rows, split_rows

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
LOSS_RTOL, PARAM_ATOL = 1e-5, 1e-5


def _args(S, H, L, ln=False, A=2, B=1024):
    return SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=ln, feature_dim=256,
                           action_size=A, max_batch=B)


def _np_sd(agent):
    return {k: v.detach().cpu().numpy() for k, v in agent.state_dict().items()}


def _cmp_params(got, ref, atol=PARAM_ATOL):
    worst = ("", 0.0)
    for k in ref:
        err = float(np.abs(got[k].astype(np.float64) - ref[k]).max())
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] <= atol, f"max-abs param error {worst[1]:.3e} at {worst[0]}"


def _cmp_params_robust(got, truth, frac_tol=1e-3, elem_tol=2e-6, max_tol=PARAM_ATOL):
    """Large nets: compare with the fp64 oracle.  fp32 rounding can flip a ReLU mask bit (|z| ~ 1e-7) in
    a ~1e6-element activation matrix; a flipped unit moves one weight row by up to ~lr/10 through Adam.
    Any two fp32 implementations differ this way (the numpy fp32 oracle sits 1.2e-5 from fp64 on 731
    weights of BASELINE config 2 after 3 steps; the HIP path 4.6e-7).  So: all but a 1e-3 fraction of
    every tensor within 2e-6 of the exact result, and nothing further than the 1e-5 bar (only the B=8192 case
    passes a wider `max_tol`, for the reason stated there)."""
    for k, ref in truth.items():
        err = np.abs(got[k].astype(np.float64) - ref)
        frac = float((err > elem_tol).mean())
        assert frac <= frac_tol, f"{k}: {frac:.2e} of elements differ by more than {elem_tol}"
        assert float(err.max()) <= max_tol, f"{k}: max-abs {err.max():.3e}"


def _oracle64(init, S, H, L, ln=False, **kw):
    import oracle.por_oracle as O
    O.set_precision(np.float64)
    return PorOracle({k: np.asarray(v, np.float64) for k, v in init.items()}, S, H, L, ln, **kw)


@pytest.fixture(autouse=True)
def _restore_precision():
    yield
    import oracle.por_oracle as O
    O.set_precision(np.float32)


def _make_por(S, H, L, B, seed=0, ln=False, **kw):
    from porl_amd.agent.por import POR
    torch.manual_seed(seed)
    return POR(_args(S, H, L, ln=ln, B=B), kw.pop("max_steps", 1000), kw.pop("tau", 0.9), kw.pop("alpha", 10.0),
               device=DEV, **kw)


@pytest.mark.parametrize("name", ["por_s60_h64_b32", "por_s17_h48_l3_b50", "por_s60_h64_b32_ln"])
def test_por_matches_reference_golden(name):
    z, meta = load_golden(name)
    S, H, L, B, K, A = (int(meta[k]) for k in ("S", "H", "L", "B", "K", "A"))
    agent = _make_por(S, H, L, B, seed=int(meta["seed_model"]), ln=bool(meta["layer_norm"]), tau=meta["tau"],
                      alpha=meta["alpha"])
    init = sub(z, "init/")
    _cmp_params(_np_sd(agent), init, atol=0.0)         # same seed -> the reference's initial weights
    rows = torch.from_numpy(make_rows(K * B, S, A, seed=int(meta["seed_data"]))).to(DEV)
    for k in range(K):
        batch = rows[k * B:(k + 1) * B]                # strided column views, like por_train.py:74-78
        s, r, sp, d, a = split_rows(batch, S, A)
        vl, gl = agent.por_residual_update(s, sp, r, d)
        np.testing.assert_allclose(vl, z["v_loss"][k], rtol=LOSS_RTOL)
        np.testing.assert_allclose(gl, z["g_loss"][k], rtol=LOSS_RTOL)
        np.testing.assert_allclose(agent.goal_lr_schedule.get_last_lr()[0], z["goal_lr_after"][k], rtol=1e-12)
    _cmp_params(_np_sd(agent), sub(z, "final/"))
    # optimizer state interchange (torch.optim.Adam format)
    am = sub(z, "adam_v/")
    sd = agent.v_optimizer.state_dict()
    names = [n for n, _ in agent.vf.named_parameters(prefix="vf")]
    for i, n in enumerate(names):
        np.testing.assert_allclose(sd["state"][i]["exp_avg"].cpu().numpy(), am[n + ".exp_avg"], atol=1e-7, rtol=1e-4)
        assert float(sd["state"][i]["step"]) == float(am["__step__"])


def _sync_oracle(o, agent):
    """Copy the engine's parameters and Adam moments into the oracle so every step is compared on its own."""
    import oracle.por_oracle as O
    for k, v in agent.state_dict().items():
        o.P[k] = v.detach().cpu().numpy().astype(O.F32)
    for opt, st, mod, prefix in ((agent.v_optimizer, o.adam_v, agent.vf, "vf"),
                                 (agent.goal_policy_optimizer, o.adam_g, agent.goal_policy, "goal_policy")):
        ms, vs = opt._moments()
        for (n, _), m, v in zip(mod.named_parameters(prefix=prefix), ms, vs):
            st.m[n] = m.cpu().numpy().astype(O.F32)
            st.v[n] = v.cpu().numpy().astype(O.F32)
        st.step = opt.step_count
    o.sched_t = agent.goal_lr_schedule.last_epoch


def _cmp_grads(named, views, ref, tag):
    for (n, _), g in zip(named, views):
        r = ref[n]
        scale = max(1e-30, float(np.abs(r).max()))
        err = np.abs(g.cpu().numpy().astype(np.float64) - r) / scale
        # A flipped ReLU mask bit (fp32 vs fp64 pre-activation within ~1e-7 of zero) changes one row by a few % of the
        # largest gradient: allow a 1e-3 fraction beyond 2e-5.  When the flipped unit sits in an UPPER layer of a sample with
        # a large advantage weight (up to 100x), that one sample's change reaches every element of the lower layers'
        # gradients at the 1e-5..1e-4 level (seen at S=60 H=512 L=3 B=2048: 15 % of dW0 beyond 2e-5 on one of two
        # trajectories that differ only by rounding; from identical state the kernels agree with the round-2 path to
        # 1e-7, scripts/dbg/skinny_ab2.py).  That signature — median error at rounding level, errors beyond 1e-3 rare — is
        # accepted too; an indexing or summation bug moves the median or puts many elements beyond 1e-3.
        frac = float((err > 2e-5).mean())
        flip_signature = float(np.median(err)) <= 5e-6 and float((err > 1e-3).mean()) <= 1e-3
        assert frac <= 1e-3 or flip_signature, (f"{tag}: grad {n}: {frac:.2e} of elements beyond 2e-5, median "
                                                f"{np.median(err):.2e}, {(err > 1e-3).mean():.2e} beyond 1e-3")
        assert float(err.max()) < 5e-2, f"{tag}: grad {n} rel-to-max error {err.max():.2e}"


def _phase_check(agent, o, s, sp, r, d, tag):
    """One update, phase by phase, against the fp64 oracle started from the engine's current state."""
    from porl_amd.engine import IqlEngine
    eng = agent._engine
    _sync_oracle(o, agent)
    f64 = lambda t: np.ascontiguousarray(t.cpu().numpy().astype(np.float64))
    s_np, sp_np, r_np, d_np = f64(s), f64(sp), f64(r), f64(d)
    B = eng.load_batch(s, sp, r, d, sp)
    agent.v_optimizer.step_count += 1
    agent.goal_policy_optimizer.step_count += 1
    hp = agent._hyper(B, agent.v_optimizer, agent.goal_policy_optimizer)
    eng.value_backward(hp)
    v_loss_o, target_o = o.value_update(s_np, sp_np, r_np, d_np)
    _cmp_grads(list(agent.vf.named_parameters(prefix="vf")), IqlEngine.views(eng.grads_vf, eng.tensor_table(0)),
               o.last_vf_grads, tag)
    np.testing.assert_allclose(float(eng.stats[0]), v_loss_o, rtol=LOSS_RTOL)
    eng.value_apply(hp)
    _cmp_params_robust({k: v for k, v in _np_sd(agent).items() if not k.startswith("goal_policy")},
                       {k: v for k, v in o.P.items() if not k.startswith("goal_policy")})
    _sync_oracle(o, agent)                       # the policy phase starts from identical value nets
    o.adam_g.step -= 1
    o.sched_t = agent.goal_lr_schedule.last_epoch
    eng.policy_backward(hp)
    g_loss_o = o.policy_update(s_np, target_o, sp_np)
    _cmp_grads(list(agent.goal_policy.named_parameters(prefix="goal_policy")),
               IqlEngine.views(eng.grads_pol, eng.tensor_table(1)), o.last_pol_grads, tag)
    np.testing.assert_allclose(float(eng.stats[1]), g_loss_o, rtol=LOSS_RTOL)
    np.testing.assert_allclose(float(eng.stats[2]), o.last_min_nlp, rtol=LOSS_RTOL)
    eng.policy_apply(hp)
    agent.goal_lr_schedule.step()
    _cmp_params_robust(_np_sd(agent), o.P)


@pytest.mark.parametrize("S,H,L,B,ln", [(60, 64, 2, 32, False), (60, 256, 2, 256, False), (17, 48, 3, 50, False),
                                        (60, 128, 1, 100, False), (362, 64, 2, 16, False),
                                        (60, 1024, 2, 1024, False), (60, 512, 3, 2048, False),
                                        (60, 64, 2, 32, True), (17, 48, 3, 50, True), (60, 100, 1, 37, True),
                                        (60, 1024, 2, 1024, True)])
def test_por_phases_vs_oracle(S, H, L, B, ln):
    agent = _make_por(S, H, L, B, ln=ln)
    o = _oracle64(_np_sd(agent), S, H, L, ln)
    rows = torch.from_numpy(make_rows(3 * B, S, 2, seed=7)).to(DEV)
    for k in range(3):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, 2)
        _phase_check(agent, o, s, sp, r, d, f"step{k}")


@pytest.mark.parametrize("name", ["por_s60_h256_b256", "por_s60_h1024_b256", "por_s60_h1024_b1024",
                                  "por_s60_h1024_b1024_ln"])
def test_por_baseline_configs_vs_golden_losses(name):
    """BASELINE configs 1/2 (H=1024, B=256/1024): losses against the reference's recorded values; ALL 5.6 M
    parameters within 1e-5 max-abs (the north_star bar) of the fp64 run of the oracle; and, directly against the
    reference, the 16 sampled elements + sums per tensor it recorded in `final_cks` (oracle/gen_golden.py)."""
    z, meta = load_golden(name)
    S, H, L, B, K, A = (int(meta[k]) for k in ("S", "H", "L", "B", "K", "A"))
    ln = bool(meta["layer_norm"])
    agent = _make_por(S, H, L, B, seed=int(meta["seed_model"]), ln=ln)
    o = _oracle64(_np_sd(agent), S, H, L, ln)
    rows_np = make_rows(K * B, S, A, seed=int(meta["seed_data"])).astype(np.float64)
    rows = torch.from_numpy(rows_np.astype(np.float32)).to(DEV)
    for k in range(K):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, A)
        vl, gl = agent.por_residual_update(s, sp, r, d)
        np.testing.assert_allclose(vl, z["v_loss"][k], rtol=LOSS_RTOL)
        np.testing.assert_allclose(gl, z["g_loss"][k], rtol=LOSS_RTOL)
        sn, rn, spn, dn, _ = split_rows(rows_np[k * B:(k + 1) * B], S, A)
        o.por_residual_update(sn, spn, rn, dn)
    got = _np_sd(agent)
    _cmp_params(got, o.P, atol=PARAM_ATOL)                  # measured: 4.6e-7 at config 2
    keys = [str(k) for k in z["keys"]]
    assert list(got.keys()) == keys
    for i, k in enumerate(keys):
        mine, ref = checksum(got[k], i), z["final_cks"][i]
        n = got[k].size
        np.testing.assert_allclose(mine[2:], ref[2:], atol=PARAM_ATOL, rtol=0, err_msg=k)
        assert abs(mine[0] - ref[0]) <= PARAM_ATOL * max(1.0, np.sqrt(n)), k
        assert abs(mine[1] - ref[1]) <= PARAM_ATOL * max(1.0, np.sqrt(n)) * 4, k


def test_por_forward_api_and_state_dict_roundtrip():
    agent = _make_por(60, 64, 2, 64)
    o = PorOracle(_np_sd(agent), 60, 64, 2)
    x = torch.from_numpy(make_rows(40, 60, 2, seed=3)[:, :60].copy()).to(DEV)
    v1, v2 = agent.vf.both(x)
    from oracle.por_oracle import twin_forward, mlp_forward
    r1, r2, _, _ = twin_forward(o.P, "vf", x.cpu().numpy(), 2, False)
    np.testing.assert_allclose(v1.cpu().numpy(), r1, atol=2e-6)
    np.testing.assert_allclose(agent.vf(x).cpu().numpy(), np.minimum(r1, r2), atol=2e-6)
    t1, _ = agent.v_target.both(x)
    np.testing.assert_allclose(t1.cpu().numpy(), r1, atol=2e-6)      # target == vf at init
    dist = agent.goal_policy(x)
    mean_ref, _ = mlp_forward(o.P, "goal_policy.net", x.cpu().numpy(), 2)
    np.testing.assert_allclose(dist.mean.cpu().numpy(), mean_ref, atol=2e-6)
    # state_dict -> fresh agent -> identical update
    sd = {k: v.clone() for k, v in agent.state_dict().items()}
    other = _make_por(60, 64, 2, 64, seed=123)
    other.load_state_dict(sd)
    rows = torch.from_numpy(make_rows(64, 60, 2, seed=9)).to(DEV)
    s, r, sp, d, a = split_rows(rows, 60, 2)
    assert agent.por_residual_update(s, sp, r, d) == other.por_residual_update(s, sp, r, d)


@pytest.mark.parametrize("name", ["sorl_s60_h64_b32", "sorl_s362_h64_b16_a10"])
def test_sorl_matches_reference_golden(name):
    from porl_amd.agent.sorl import SORL
    z, meta = load_golden(name)
    S, H, L, B, K, A = (int(meta[k]) for k in ("S", "H", "L", "B", "K", "A"))
    torch.manual_seed(int(meta["seed_model"]))
    agent = SORL(_args(S, H, L, A=A, B=B), int(meta["max_steps"]), meta["tau"], meta["alpha"], device=DEV)
    _cmp_params(_np_sd(agent), sub(z, "init/"), atol=0.0)
    rows = torch.from_numpy(make_rows(K * B, S, A, seed=int(meta["seed_data"]))).to(DEV)
    for k in range(K):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, A)
        vl, gl = agent.update(s, a, r, sp, d)
        np.testing.assert_allclose(vl, z["v_loss"][k], rtol=LOSS_RTOL)
        np.testing.assert_allclose(gl, z["g_loss"][k], rtol=2e-5)
    _cmp_params(_np_sd(agent), sub(z, "final/"))
    np.testing.assert_allclose(agent.select_action(rows[:8, :S]), z["select_action"], atol=2e-6)


def test_sorl_vf_update_matches_reference_golden():
    from porl_amd.agent.sorl import SORL
    z, meta = load_golden("sorl_vf_s60_h64_b32")
    S, H, L, B, K, A = (int(meta[k]) for k in ("S", "H", "L", "B", "K", "A"))
    torch.manual_seed(int(meta["seed_model"]))
    agent = SORL(_args(S, H, L, A=A, B=B), 1000, meta["tau"], meta["alpha"], device=DEV)
    rows = torch.from_numpy(make_rows(K * B, S, A, seed=int(meta["seed_data"]))).to(DEV)
    for k in range(K):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, A)
        np.testing.assert_allclose(agent.vf_update(s, a, r, sp, d), z["v_loss"][k], rtol=LOSS_RTOL)
    _cmp_params(_np_sd(agent), sub(z, "final/"))


def test_error_conventions():
    agent = _make_por(60, 64, 2, 32)
    rows = torch.from_numpy(make_rows(64, 60, 2, seed=1)).to(DEV)
    s, r, sp, d, a = split_rows(rows, 60, 2)
    with pytest.raises(RuntimeError):                       # batch larger than the engine was sized for
        agent.por_residual_update(s, sp, r, d)
    with pytest.raises(RuntimeError):                       # wrong feature width
        agent.por_residual_update(s[:32, :59], sp[:32], r[:32], d[:32])
    with pytest.raises(RuntimeError):                       # wrong device
        agent.por_residual_update(s[:32].cpu(), sp[:32], r[:32], d[:32])


def test_update_from_replay_equals_update_on_the_drawn_rows():
    """Device sampler + gather + split (one kernel) feeds the same arithmetic as the tensor API."""
    from porl_amd.buffer.replay_buffer import PackedReplay
    S, A, B, N = 60, 2, 128, 5000
    rows = make_rows(N, S, A, seed=11)
    a1, a2 = _make_por(S, 64, 2, B), _make_por(S, 64, 2, B)
    rp = PackedReplay(rows, S, A, DEV, seed=3)
    for step in range(3):
        idx = torch.empty(B, dtype=torch.int64, device=DEV)
        a1._engine.load_batch_sampled(rp.rows, B, rp.seed, rp.draws, A, False, idx_out=idx)   # peek at the draw
        got = a1.update_from_replay(rp, B)
        ih = idx.cpu().numpy()
        assert len(set(ih.tolist())) == B and ih.min() >= 0 and ih.max() < N          # distinct, in range
        batch = torch.from_numpy(rows[ih]).to(DEV)
        s, r, sp, d, _ = split_rows(batch, S, A)
        want = a2.por_residual_update(s, sp, r, d)
        assert got == want
    for (k, v1), v2 in zip(a1.state_dict().items(), a2.state_dict().values()):
        assert torch.equal(v1, v2), k


def test_device_sampler_is_a_uniform_permutation_prefix():
    from porl_amd.engine import sample_indices
    n, B = 1000, 1000
    idx = sample_indices(n, B, seed=5, step=0, device=DEV).cpu().numpy()
    assert sorted(idx.tolist()) == list(range(n))                # batch == n -> a permutation of all rows
    # different steps / seeds give different draws; counts over many draws are flat
    counts = np.zeros(97, dtype=np.int64)
    for step in range(400):
        counts += np.bincount(sample_indices(97, 10, seed=1, step=step, device=DEV).cpu().numpy(), minlength=97)
    expect = 400 * 10 / 97
    assert counts.min() > 0.5 * expect and counts.max() < 1.6 * expect
    with pytest.raises(Exception):
        sample_indices(10, 11, seed=0, step=0, device=DEV)        # B > size: like numpy's ValueError


def test_stats_redirection_keeps_a_device_loss_history():
    agent = _make_por(60, 64, 2, 32)
    agent.async_losses = True
    hist = torch.zeros(4, 8, device=DEV)
    rows = torch.from_numpy(make_rows(4 * 32, 60, 2, seed=2)).to(DEV)
    ref = _make_por(60, 64, 2, 32)
    want = []
    for k in range(4):
        s, r, sp, d, a = split_rows(rows[k * 32:(k + 1) * 32], 60, 2)
        agent._engine.set_stats(hist[k])
        agent.por_residual_update(s, sp, r, d)
        want.append(ref.por_residual_update(s, sp, r, d))
    agent.flush()                                   # the last policy phase (side stream) writes g_loss of the last row
    np.testing.assert_array_equal(hist[:, :2].cpu().numpy(), np.array(want, dtype=np.float32))


def test_por_global_batch_of_config4_on_one_gpu():
    """B=8192 (the global batch of BASELINE config 4) in one engine: losses and all 5.6 M parameters after one
    update against the fp64 oracle; exercises workspace sizing, split-K choices and reductions at 8x the headline
    batch, and the 1/B scaling the data-parallel mode relies on."""
    S, H, L, B = 60, 1024, 2, 8192
    agent = _make_por(S, H, L, B)
    o = _oracle64(_np_sd(agent), S, H, L, False)
    rows_np = make_rows(B, S, 2, seed=41).astype(np.float64)
    rows = torch.from_numpy(rows_np.astype(np.float32)).to(DEV)
    s, r, sp, d, a = split_rows(rows, S, 2)
    vl, gl = agent.por_residual_update(s, sp, r, d)
    sn, rn, spn, dn, _ = split_rows(rows_np, S, 2)
    vo, go = o.por_residual_update(sn, spn, rn, dn)
    np.testing.assert_allclose([vl, gl], [vo, go], rtol=LOSS_RTOL)
    # the first Adam step is lr * g / (|g| + 1e-8): for the handful of weights whose gradient is ~1e-8 the fp32 noise
    # of an 8192-row sum decides the sign, so a single element may be off by up to 2 * lr = 2e-4
    _cmp_params_robust(_np_sd(agent), o.P, max_tol=2.1e-4)


@pytest.fixture(params=[0, 7], ids=["gemm-path", "skinny-path"])
def same_kernels_in_both_modes(request):
    """The pipelined update and the one-stream update pick their <= 64-wide products' kernels separately (csrc: g_skinny /
    g_skinny_pipelined — by default skinny.hpp on one stream, the grouped GEMM in the pipelined update, where it measured
    faster).  "Pipelining only reorders" is a statement about equal kernels, so these tests pin one selection for both
    modes — each of the two; across selections results agree to rounding (test_skinny_kernels_agree_...)."""
    from porl_amd import engine as E
    E.tune_set("skinny", request.param)
    E.tune_set("skinny_pipelined", request.param)
    yield request.param
    E.tune_set("skinny", 7)
    E.tune_set("skinny_pipelined", -1)          # the default: by hidden width (on up to 512, off above)


def test_pipelined_updates_are_bit_identical_to_back_to_back_updates(same_kernels_in_both_modes):
    """async_losses=True issues the policy phase of update t on the engine's side stream and lets update t+1's value
    phase run beside it (agent/_iql.py).  Same kernels on the same data in the same order per buffer: every loss,
    parameter and Adam moment must equal the single-stream run bit for bit; state read right after an update
    (optimizer state_dict, a policy forward) must already contain that update."""
    S, A, B, H, K = 60, 2, 256, 256, 6
    rows = torch.from_numpy(make_rows(K * B, S, A, seed=5)).to(DEV)
    a_sync, a_pipe = _make_por(S, H, 2, B), _make_por(S, H, 2, B)
    a_pipe.async_losses = True
    hist = torch.zeros(K, 8, device=DEV)
    want = []
    for k in range(K):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, A)
        want.append(a_sync.por_residual_update(s, sp, r, d))
        a_pipe._engine.set_stats(hist[k])
        a_pipe.por_residual_update(s, sp, r, d)
        if k == 2:                                  # mid-run reads see a consistent agent
            sd_p = a_pipe.goal_policy_optimizer.state_dict()
            sd_s = a_sync.goal_policy_optimizer.state_dict()
            for i in sd_s["state"]:
                assert torch.equal(sd_p["state"][i]["exp_avg"], sd_s["state"][i]["exp_avg"])
                assert float(sd_p["state"][i]["step"]) == float(sd_s["state"][i]["step"]) == 3.0
            assert torch.equal(a_pipe.goal_policy(s).mean, a_sync.goal_policy(s).mean)
    assert a_pipe._engine._policy_done is not None
    for (k1, v1), v2 in zip(a_pipe.state_dict().items(), a_sync.state_dict().values()):   # state_dict() flushes
        assert torch.equal(v1, v2), k1
    np.testing.assert_array_equal(hist[:, :2].cpu().numpy(), np.array(want, dtype=np.float32))


@pytest.mark.parametrize("sync,ln", [("signal", False), ("event", False), ("signal", True), ("signal/phase-calls", False)])
def test_pipelined_updates_at_the_headline_size_are_bit_identical(sync, ln, monkeypatch, same_kernels_in_both_modes):
    """The same property at BASELINE config 2 (H = 1024, B = 1024, device sampler), where the two streams really
    overlap: 64x64 short blocks, the value phase's 1 024-block launches at two blocks per CU, three staging slots, the
    streams ordered by signal counters (or events).  40 pipelined updates against 40 one-stream updates: every
    parameter, target parameter and Adam moment equal bit for bit, and so is the loss history."""
    import porl_amd.agent._iql as iql
    from porl_amd.buffer.replay_buffer import PackedReplay
    # "signal" issues the whole update from one native call (porl_iql_update_pipelined); ".../phase-calls" keeps the
    # phase-by-phase sequence from Python
    monkeypatch.setattr(iql, "_PIPE_SYNC", sync.split("/")[0])
    monkeypatch.setattr(iql, "_PIPE_ONECALL", "/" not in sync)
    S, A, B, H, K = 60, 2, 1024, 1024, 40
    rows = make_rows(50_000, S, A, seed=11)
    out = []
    for pipe in (False, True):
        agent = _make_por(S, H, 2, B, ln=ln)
        agent.async_losses = True
        agent.pipeline = pipe
        rp = PackedReplay(rows, S, A, DEV, seed=3)
        hist = torch.zeros(K, 8, device=DEV)
        for k in range(K):
            agent._engine.set_stats(hist[k])
            agent.update_from_replay(rp, B)
        agent.flush()
        torch.cuda.synchronize()
        eng = agent._engine
        out.append([t.clone() for t in (eng.params_vf, eng.params_tgt, eng.params_pol, eng.adam_m_vf, eng.adam_v_vf,
                                        eng.adam_m_pol, eng.adam_v_pol, hist[:, :3])])
    for a, b, what in zip(out[0], out[1], ("vf", "target", "policy", "m_vf", "v_vf", "m_pol", "v_pol", "losses")):
        assert torch.equal(a, b), what
    assert torch.isfinite(out[0][-1]).all()


def test_engine_on_a_non_current_device_guard():
    """The C entry points launch on the device that owns the engine's buffers, and the stream handed over is torch's
    current stream OF THAT device — whatever the thread's current device is (_native.py:202)."""
    if torch.cuda.device_count() < 2:
        # one-GPU box: the guard's own logic still runs (device_of(workspace) == current device); check it is recorded
        agent = _make_por(60, 64, 2, 32)
        rows = torch.from_numpy(make_rows(32, 60, 2, seed=1)).to(DEV)
        s, r, sp, d, a = split_rows(rows, 60, 2)
        agent.por_residual_update(s, sp, r, d)
        assert agent._engine.device.index == torch.cuda.current_device()
        return
    from porl_amd.agent.por import POR
    dev1 = torch.device("cuda", 1)
    torch.manual_seed(0)
    a1 = POR(_args(60, 64, 2, B=32), 1000, 0.9, 10.0, device=dev1)
    a0 = _make_por(60, 64, 2, 32)
    rows = make_rows(32, 60, 2, seed=1)
    s, r, sp, d, a = split_rows(torch.from_numpy(rows).to(DEV), 60, 2)
    s1, r1, sp1, d1, _ = split_rows(torch.from_numpy(rows).to(dev1), 60, 2)
    assert torch.cuda.current_device() == 0
    assert a1.por_residual_update(s1, sp1, r1, d1) == a0.por_residual_update(s, sp, r, d)


def test_checkpoint_roundtrip_through_torch_save(tmp_path):
    """(f)2 checkpoint interchange: agent + BOTH optimizers + the LR scheduler go through torch.save / torch.load
    (weights_only) into a freshly constructed agent, which must continue bit-identically with the original."""
    S, A, B, H = 60, 2, 64, 128
    rows = torch.from_numpy(make_rows(6 * B, S, A, seed=13)).to(DEV)
    a = _make_por(S, H, 2, B)
    for k in range(3):
        s, r, sp, d, _ = split_rows(rows[k * B:(k + 1) * B], S, A)
        a.por_residual_update(s, sp, r, d)
    path = tmp_path / "ckpt.pt"
    torch.save({"agent": a.state_dict(), "v_opt": a.v_optimizer.state_dict(),
                "g_opt": a.goal_policy_optimizer.state_dict(), "sched": a.goal_lr_schedule.state_dict()}, path)
    ck = torch.load(path, map_location=DEV, weights_only=True)
    b = _make_por(S, H, 2, B, seed=77)                         # different init: everything must come from the file
    b.load_state_dict(ck["agent"])
    b.v_optimizer.load_state_dict(ck["v_opt"])
    b.goal_policy_optimizer.load_state_dict(ck["g_opt"])
    b.goal_lr_schedule.load_state_dict(ck["sched"])
    assert b.goal_lr_schedule.get_last_lr() == a.goal_lr_schedule.get_last_lr()
    for k in range(3, 6):
        s, r, sp, d, _ = split_rows(rows[k * B:(k + 1) * B], S, A)
        assert a.por_residual_update(s, sp, r, d) == b.por_residual_update(s, sp, r, d)
    for (k1, v1), v2 in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(v1, v2), k1
    sa, sb = a.v_optimizer.state_dict(), b.v_optimizer.state_dict()
    for i in sa["state"]:
        assert torch.equal(sa["state"][i]["exp_avg_sq"], sb["state"][i]["exp_avg_sq"])
    # the optimizer file is torch.optim.Adam's format: a stock Adam over same-shaped parameters accepts it
    stock = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(p)) for p in a.vf.parameters()], lr=1e-4)
    stock.load_state_dict(ck["v_opt"])
    assert stock.state_dict()["state"][0]["step"] == 3


@pytest.mark.parametrize("B", [1, 3, 8])
def test_small_batch_inference_path_matches_the_batched_kernels(B):
    """(f)2 low-latency forward: at B <= 8 the policy mean and the twin values come from 3 GEMV launches
    (small_fwd_kernel); same numbers as the training-size path (B = 9 rows go through the grouped GEMMs) and as the
    oracle."""
    from oracle.por_oracle import mlp_forward, twin_forward
    agent = _make_por(60, 256, 2, 64)
    o = PorOracle(_np_sd(agent), 60, 256, 2)
    x9 = torch.from_numpy(make_rows(9, 60, 2, seed=3)[:, :60].copy()).to(DEV)
    big = agent.goal_policy(x9).mean[:B]
    small = agent.goal_policy(x9[:B]).mean
    assert small.shape == (B, 60)
    np.testing.assert_allclose(small.cpu().numpy(), big.cpu().numpy(), atol=2e-6)
    ref, _ = mlp_forward(o.P, "goal_policy.net", x9[:B].cpu().numpy(), 2)
    np.testing.assert_allclose(small.cpu().numpy(), ref, atol=2e-6)
    v1, v2 = agent.vf.both(x9[:B])
    r1, r2, _, _ = twin_forward(o.P, "vf", x9[:B].cpu().numpy(), 2, False)
    np.testing.assert_allclose(v1.cpu().numpy(), r1, atol=2e-6)
    np.testing.assert_allclose(v2.cpu().numpy(), r2, atol=2e-6)
    # the rollout path: mean as numpy through pinned host memory, from a device tensor, a CPU tensor or an ndarray
    want = small.cpu().numpy()
    for src in (x9[:B], x9[:B].cpu(), x9[:B].cpu().numpy()):
        got = agent.goal_policy.mean_numpy(src)
        assert isinstance(got, np.ndarray) and got.dtype == np.float32
        np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(agent.goal_policy.mean_numpy(x9[0].cpu().numpy()), want[0])
    np.testing.assert_allclose(agent.goal_policy.mean_numpy(x9.cpu().numpy()), agent.goal_policy(x9).mean.cpu().numpy(), atol=0)
    # strided input rows (a column slice of a packed batch) are read in place
    packed = torch.from_numpy(make_rows(B, 60, 2, seed=4)).to(DEV)
    np.testing.assert_allclose(agent.goal_policy(packed[:, :60]).mean.cpu().numpy(),
                               agent.goal_policy(packed[:, :60].contiguous()).mean.cpu().numpy(), atol=0)


@pytest.mark.parametrize("S,H,B", [(60, 1024, 1024), (20, 48, 50), (64, 200, 130), (8, 132, 64)])
def test_input_layer_kernel_is_bit_identical_to_the_grouped_gemm(S, H, B):
    """csrc/l0_fwd.hpp: the K = obs_dim <= 64 input layers run on a one-round kernel of their own; it reproduces the
    grouped GEMM's MFMA chain (same k order, zero padding, bias, ReLU), so every loss and parameter of a run equals the
    run with porl_tune_set("l0_kernel", 0) bit for bit — at the headline shape and at ragged ones (rows and columns
    that do not fill a 64 x 128 tile, K below and at the 64 limit)."""
    from porl_amd import _native as N
    rows = torch.from_numpy(make_rows(3 * B, S, 2, seed=21)).to(DEV)
    out = []
    try:
        for use in (1, 0):
            N.check(N.lib().porl_tune_set(b"l0_kernel", use), "porl_tune_set")
            agent = _make_por(S, H, 2, B)
            losses = []
            for k in range(3):
                s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, 2)
                losses.append(agent.por_residual_update(s, sp, r, d))
            both = agent.vf.both(rows[:5, :S].contiguous())
            out.append((losses, [v.clone() for v in agent.state_dict().values()], [b.clone() for b in both]))
    finally:
        N.check(N.lib().porl_tune_set(b"l0_kernel", 1), "porl_tune_set")
    assert out[0][0] == out[1][0]
    for x, y in zip(out[0][1], out[1][1]):
        assert torch.equal(x, y)
    for x, y in zip(out[0][2], out[1][2]):
        assert torch.equal(x, y)


def test_one_call_pipelined_update_rejects_bad_arguments_without_side_effects():
    """porl_iql_update_pipelined (the small-network fast path of update_from_replay) validates the minibatch before it
    enqueues anything: an oversized batch raises, the step counters stay where they were, and the agent goes on to
    produce the same results as one that never saw the bad call."""
    from porl_amd import _native as N
    from porl_amd.buffer.replay_buffer import PackedReplay
    S, A, B, H = 60, 2, 64, 64
    rows = make_rows(5_000, S, A, seed=9)
    a, b = _make_por(S, H, 2, B), _make_por(S, H, 2, B)
    for ag in (a, b):
        ag.async_losses = True
    ra, rb = PackedReplay(rows, S, A, DEV, seed=4), PackedReplay(rows, S, A, DEV, seed=4)
    a.update_from_replay(ra, B)
    b.update_from_replay(rb, B)
    with pytest.raises(N.NativeError):
        a.update_from_replay(ra, B + 1)                      # max_batch is B
    assert a.v_optimizer.step_count == 1 and ra.draws == rb.draws
    a.update_from_replay(ra, B)
    b.update_from_replay(rb, B)
    for (k, x), y in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(x, y), k


def test_agent_built_on_the_cpu_moves_to_the_device_and_updates():
    """Round-2 advisor finding: POR/SORL default to device=cpu like the reference (por.py:21, por_train.py:57 then moves
    nothing — but `.to('cuda')` / `.cuda()` / load_state_dict-on-cpu-then-move are the usual patterns).  The moved agent
    must equal an agent built on the device from the same seed."""
    from porl_amd.agent.por import POR
    S, H, B = 60, 64, 32
    torch.manual_seed(0)
    a = POR(_args(S, H, 2, B=B), 1000, 0.9, 10.0)                 # device=cpu default
    assert a._engine.device.type == "cpu"
    sd_cpu = {k: v.clone() for k, v in a.state_dict().items()}
    a = a.to("cuda")
    assert a._engine.device.type == "cuda" and all(p.is_cuda for p in a.parameters())
    b = _make_por(S, H, 2, B)
    for k, v in b.state_dict().items():
        assert torch.equal(v.cpu(), sd_cpu[k]), k
    rows = torch.from_numpy(make_rows(2 * B, S, 2, seed=4)).to(DEV)
    for k in range(2):
        s, r, sp, d, _ = split_rows(rows[k * B:(k + 1) * B], S, 2)
        assert a.por_residual_update(s, sp, r, d) == b.por_residual_update(s, sp, r, d)
    for (k1, v1), v2 in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(v1, v2), k1
    # and back: a CPU copy holds the same numbers and refuses to compute (no CPU path)
    c = a.cpu()
    assert c._engine.device.type == "cpu"
    for (k1, v1), v2 in zip(c.state_dict().items(), b.state_dict().values()):
        assert v1.device.type == "cpu" and torch.equal(v1, v2.cpu()), k1
    with pytest.raises(Exception):
        c.por_residual_update(*[t.cpu() for t in split_rows(rows[:B], S, 2)[:1]] * 2, rows[:B, 0].cpu(), rows[:B, 0].cpu())
    c = c.cuda()
    s, r, sp, d, _ = split_rows(rows[:B], S, 2)
    assert c.por_residual_update(s, sp, r, d) == b.por_residual_update(s, sp, r, d)


def test_adam_sweep_over_an_empty_range_is_a_no_op():
    """porl_adam_ema(n = 0) is a public ABI entry: an empty sweep, not a division by zero (round-2 advisor finding)."""
    from porl_amd import _native as N
    buf = torch.full((4, 4), 3.0, device=DEV)                     # valid, aligned pointers; length 0 (torch hands out a
    p, g, m, v = (N.ptr(buf[i]) for i in range(4))                # null data_ptr for empty tensors, so go through the ABI)
    N.check(N.lib().porl_adam_ema(p, g, m, v, None, 0, 1e-3, 1, 0.9, 0.999, 1e-8, 0.0, N.current_stream_ptr(buf)), "porl_adam_ema")
    torch.cuda.synchronize()
    assert (buf == 3.0).all()


@pytest.mark.parametrize("S,H,L,B,sorl", [(60, 1024, 2, 1024, False), (17, 48, 3, 50, False), (60, 256, 2, 130, True),
                                          (64, 192, 1, 2100, False), (8, 72, 2, 64, True), (60, 512, 3, 2048, False)])
def test_skinny_kernels_agree_with_the_grouped_gemm_path(S, H, L, B, sorl):
    """The <= 64-wide products of the step (input-layer weight gradients, policy mean, policy output-layer backward) run
    on csrc/skinny.hpp; porl_tune_set("skinny", 0) sends them through the grouped GEMM as in round 2.  Same partial sums
    per 64-chunk in another order, so from IDENTICAL state (three updates; both paths' backward passes run on every
    update before the parameters move) gradients and losses agree to fp32 rounding.  Covers ragged shapes, more than 16
    row tiles (several tiles per slab) and SORL's 2-wide output layer; both paths are pinned to the reference by the
    golden tests above.  (Whole trajectories are NOT compared: Adam's first steps turn a 1e-10 gradient difference on
    a |g| ~ 1e-8 entry into a 1e-6 parameter difference, and ReLU-mask flips do the rest.)"""
    from porl_amd import engine as E
    from porl_amd.agent.sorl import SORL
    from porl_amd.engine import IqlEngine
    A = 2
    rows = torch.from_numpy(make_rows(3 * B, S, A, seed=31)).to(DEV)
    if sorl:
        torch.manual_seed(0)
        agent = SORL(_args(S, H, L, A=A, B=B), 1000, 0.9, 3.0, device=DEV)
        v_opt, p_opt, sched = agent.v_optimizer, agent.policy_optimizer, agent.lr_schedule
    else:
        agent = _make_por(S, H, L, B)
        v_opt, p_opt, sched = agent.v_optimizer, agent.goal_policy_optimizer, agent.goal_lr_schedule
    eng = agent._engine
    try:
        for k in range(3):
            s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, A)
            Bk = eng.load_batch(s, sp, r, d, a if sorl else sp)
            v_opt.step_count += 1
            p_opt.step_count += 1
            hp = agent._hyper(Bk, v_opt, p_opt)
            for phase, group, apply in (("value_backward", 0, "value_apply"), ("policy_backward", 1, "policy_apply")):
                got = {}
                for skinny in (0, 1):
                    E.tune_set("skinny", skinny)
                    getattr(eng, phase)(hp)
                    flat = eng.grads_vf if group == 0 else eng.grads_pol
                    got[skinny] = ([g.clone() for g in IqlEngine.views(flat, eng.tensor_table(group))], eng.stats[:3].clone())
                for i, (g0, g1) in enumerate(zip(got[0][0], got[1][0])):
                    scale = float(g0.abs().max())
                    assert float((g0 - g1).abs().max()) <= 1e-6 * scale + 1e-12, (k, phase, i, scale)
                np.testing.assert_allclose(got[1][1].cpu().numpy(), got[0][1].cpu().numpy(), rtol=1e-6)
                getattr(eng, apply)(hp)
            sched.step()
    finally:
        E.tune_set("skinny", 1)
