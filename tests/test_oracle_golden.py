"""Pins oracle/por_oracle.py against golden vectors recorded from the reference itself
(oracle/gen_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO, load_golden, sub, checksum
from oracle.por_oracle import PorOracle, sorl_oracle, CqlOracle
from porl_amd.util.synth import make_rows, split_rows, make_discrete_transitions

LOSS_RTOL = 1e-5
PARAM_ATOL = 1e-5


def _assert_params(P, final, atol=PARAM_ATOL):
    for k, ref in final.items():
        got = P[k]
        assert got.shape == ref.shape, k
        err = np.abs(got.astype(np.float64) - ref).max() if ref.size else 0.0
        assert err <= atol, f"{k}: max-abs {err:.3e}"


def _run_por(z, meta, init):
    S, H, L, B, K, A = (int(meta[k]) for k in ("S", "H", "L", "B", "K", "A"))
    o = PorOracle(init, S, H, L, bool(meta["layer_norm"]), tau=meta["tau"], alpha=meta["alpha"],
                  max_steps=int(meta["max_steps"]))
    rows = make_rows(K * B, S, A, seed=int(meta["seed_data"]))
    vl, gl = [], []
    for k in range(K):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, A)
        v, g = o.por_residual_update(s, sp, r, d)
        vl.append(v)
        gl.append(g)
    np.testing.assert_allclose(vl, z["v_loss"], rtol=LOSS_RTOL)
    np.testing.assert_allclose(gl, z["g_loss"], rtol=LOSS_RTOL)
    return o


@pytest.mark.parametrize("name", ["por_s60_h64_b32", "por_s60_h64_b32_ln", "por_s17_h48_l3_b50"])
def test_por_small_full(name):
    z, meta = load_golden(name)
    o = _run_por(z, meta, sub(z, "init/"))
    _assert_params(o.P, sub(z, "final/"))
    am = sub(z, "adam_v/")
    assert o.adam_v.step == int(am["__step__"])
    for n in o.vf_names:
        np.testing.assert_allclose(o.adam_v.m[n], am[n + ".exp_avg"], atol=1e-7, rtol=1e-4)
        np.testing.assert_allclose(o.adam_v.v[n], am[n + ".exp_avg_sq"], atol=1e-9, rtol=1e-4)
    ag = sub(z, "adam_g/")
    for n in o.pol_names:
        np.testing.assert_allclose(o.adam_g.m[n], ag[n + ".exp_avg"], atol=1e-6, rtol=1e-4)
    # cosine schedule value after each update (por.py:110)
    from oracle.por_oracle import cosine_lr
    want = [cosine_lr(meta["policy_lr"], t + 1, int(meta["max_steps"])) for t in range(int(meta["K"]))]
    np.testing.assert_allclose(want, z["goal_lr_after"], rtol=1e-12)


def _seeded_init(meta, sorl=False):
    """Large cases store only checksums; the initial weights are rebuilt with torch's seeded
    default init in the reference's construction order and checked."""
    torch = pytest.importorskip("torch")
    from porl_amd.util.init import build_por_state_dict
    return build_por_state_dict(int(meta["S"]), int(meta["H"]), int(meta["L"]), bool(meta["layer_norm"]),
                                seed=int(meta["seed_model"]))


@pytest.mark.parametrize("name", ["por_s60_h256_b256", "por_s60_h1024_b256", "por_s60_h1024_b1024",
                                  "por_s60_h1024_b1024_ln"])
def test_por_large_checksums(name):
    z, meta = load_golden(name)
    init = _seeded_init(meta)
    keys = [str(k) for k in z["keys"]]
    assert list(init.keys()) == keys
    for i, k in enumerate(keys):
        np.testing.assert_allclose(checksum(init[k], i), z["init_cks"][i], rtol=0, atol=0)
    o = _run_por(z, meta, init)
    for i, k in enumerate(keys):
        got, ref = checksum(o.P[k], i), z["final_cks"][i]
        n = o.P[k].size
        # sampled elements: plain tolerance; sums: tolerance scaled by element count
        np.testing.assert_allclose(got[2:], ref[2:], atol=PARAM_ATOL, rtol=0, err_msg=k)
        assert abs(got[0] - ref[0]) <= PARAM_ATOL * max(1.0, np.sqrt(n)), k
        assert abs(got[1] - ref[1]) <= PARAM_ATOL * max(1.0, np.sqrt(n)) * 4, k


@pytest.mark.parametrize("name", ["sorl_s60_h64_b32", "sorl_s362_h64_b16_a10"])
def test_sorl_update(name):
    z, meta = load_golden(name)
    S, H, L, B, K, A = (int(meta[k]) for k in ("S", "H", "L", "B", "K", "A"))
    o = sorl_oracle(sub(z, "init/"), S, H, L, bool(meta["layer_norm"]), tau=meta["tau"],
                    alpha=meta["alpha"], max_steps=int(meta["max_steps"]))
    rows = make_rows(K * B, S, A, seed=int(meta["seed_data"]))
    for k in range(K):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, A)
        v, g = o.sorl_update(s, a, r, sp, d)
        np.testing.assert_allclose(v, z["v_loss"][k], rtol=LOSS_RTOL)
        np.testing.assert_allclose(g, z["g_loss"][k], rtol=2e-5)
    _assert_params(o.P, sub(z, "final/"))
    np.testing.assert_allclose(o.select_action(rows[:8, :S]), z["select_action"], atol=1e-6)


def test_sorl_vf_update():
    z, meta = load_golden("sorl_vf_s60_h64_b32")
    S, H, L, B, K, A = (int(meta[k]) for k in ("S", "H", "L", "B", "K", "A"))
    o = sorl_oracle(sub(z, "init/"), S, H, L, False, tau=meta["tau"], alpha=meta["alpha"])
    rows = make_rows(K * B, S, A, seed=int(meta["seed_data"]))
    for k in range(K):
        s, r, sp, d, a = split_rows(rows[k * B:(k + 1) * B], S, A)
        np.testing.assert_allclose(o.sorl_vf_update(s, a, r, sp, d), z["v_loss"][k], rtol=LOSS_RTOL)
    _assert_params(o.P, sub(z, "final/"))


@pytest.mark.parametrize("name", ["cql_s60_a10_b64", "cql_s8_a4_b256"])
def test_cql_learn(name):
    z, meta = load_golden(name)
    S, A, B, K, N = (int(meta[k]) for k in ("S", "A", "B", "K", "N"))
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=int(meta["seed_data"]))
    o = CqlOracle(sub(z, "init/"), A, gamma=meta["gamma"], alpha=meta["alpha"], lr=meta["lr"])
    for k in range(K):
        idx = z["indices"][k]
        loss = o.learn(st[idx], ac[idx], rw[idx], ns[idx], dn[idx])
        np.testing.assert_allclose(loss, z["loss"][k], rtol=LOSS_RTOL)
        if (k + 1) % int(meta["sync_every"]) == 0:
            o.sync_target()
    _assert_params(o.Q, sub(z, "final/"))
    _assert_params(o.T, sub(z, "final_target/"))
    idx = z["indices"][0]
    np.testing.assert_allclose(o.penalty(st[idx], ac[idx]), float(z["penalty_final_on_batch0"]),
                               atol=2e-6)


# ---- costmap encoder (oracle/fasternet_oracle.py) ------------------------------------------------------
def test_costmap_oracle_matches_reference_golden():
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import fasternet_oracle as FO
    z, _ = load_golden("costmap_b24")
    st = z["state_in"].copy()
    out = FO.state2costmap(st)
    nz = np.argwhere(out != 0).astype(np.int32)
    assert np.array_equal(nz, z["nonzero"])
    assert np.array_equal(st, z["state_after"])


def test_fasternet_oracle_matches_reference_golden():
    """Weights come from the drop-in's constructor (same seed => the reference's initialisation, pinned by the
    fixture's per-tensor checksums); the oracle then reproduces the reference's eval and train forwards,
    its intermediate activations and its BatchNorm running statistics."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import fasternet_oracle as FO
    from porl_amd.agent.fasternet import FasterNet
    z, _ = load_golden("fasternet_b5")
    torch.manual_seed(int(z["seed_model"]))
    m = FasterNet(3, 256)
    sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    for i, (k, v) in enumerate(sd.items()):
        if v.ndim:
            assert np.allclose(checksum(v, i), z["wsum." + k], rtol=1e-6, atol=1e-9), k
    stats = {k: v.copy() for k, v in sd.items() if "running" in k}
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    taps = {}
    f = FO.forward(sd, stats, z["states"].copy(), False, taps=taps)
    assert rel(f, z["feat_eval"]) < 5e-6
    for k, v in taps.items():
        v = v.reshape(v.shape[0], v.shape[1], -1) if v.ndim == 4 else v
        mine = np.concatenate([[v.sum(), np.abs(v).sum()], v[:, :6].reshape(v.shape[0], -1)[:, :24].ravel()])
        assert rel(mine, z["tap_eval." + k]) < 5e-6, k
    f1 = FO.forward(sd, stats, z["states"].copy(), True, z["drop_scale1"])
    f2 = FO.forward(sd, stats, z["states2"].copy(), True, z["drop_scale2"])
    assert rel(f1, z["feat_train1"]) < 5e-6 and rel(f2, z["feat_train2"]) < 5e-6
    for k in stats:
        if k.endswith("running_var"):
            assert rel(stats[k], z["stat_after." + k]) < 2e-6, k
        elif k.endswith("running_mean"):
            assert np.abs(stats[k] - z["stat_after." + k]).max() < 1e-6, k
    f3 = FO.forward(sd, stats, z["states"].copy(), False)
    assert rel(f3, z["feat_eval_after"]) < 2e-5     # the reference's own fp32 noise: features ~1e-3 here


def test_torch_cpu_baseline_restatements_agree_with_the_oracle():
    """oracle/torch_cpu.py (eager PyTorch-CPU, what bench.py's cpu_baseline leg times) against the numpy oracle that
    the goldens above pin: same losses over three POR updates and three CQL learn steps."""
    torch = pytest.importorskip("torch")
    from oracle.por_oracle import CqlOracle, PorOracle
    from oracle.torch_cpu import CqlTorchCpu, PorTorchCpu
    from porl_amd.util.init import build_por_state_dict
    from porl_amd.util.synth import make_discrete_transitions
    S, H, L, B = 12, 48, 2, 40
    sd = build_por_state_dict(S, H, L, seed=3)
    o, t = PorOracle({k: v.copy() for k, v in sd.items()}, S, H, L), PorTorchCpu(sd, S, H, L)
    rows = make_rows(3 * B, S, 2, seed=8)
    for k in range(3):
        s, r, sp, d, _ = split_rows(rows[k * B:(k + 1) * B], S, 2)
        want = o.por_residual_update(np.ascontiguousarray(s), np.ascontiguousarray(sp), r, d)
        got = t.update(*(torch.from_numpy(np.ascontiguousarray(x)) for x in (s, sp, r, d)))
        np.testing.assert_allclose(got, want, rtol=2e-5)
    torch.manual_seed(0)
    from porl_amd.net.q_network import QNetwork
    qsd = {k: v.numpy().copy() for k, v in QNetwork(10, 5).state_dict().items()}
    co, ct = CqlOracle({k: v.copy() for k, v in qsd.items()}, 5), CqlTorchCpu(qsd, 5)
    st, ac, rw, ns, dn = make_discrete_transitions(3 * 64, 10, 5, seed=2)
    for k in range(3):
        sl = slice(k * 64, (k + 1) * 64)
        want = co.learn(st[sl], ac[sl], rw[sl], ns[sl], dn[sl])
        got = ct.learn(*(torch.from_numpy(x[sl]) for x in (st, ac, rw, ns, dn)))
        np.testing.assert_allclose(got, want if np.isscalar(want) else want[0], rtol=2e-5)


# ---- Implicit Quantile Network (oracle/iqn_oracle.py) ---------------------------------------------------
def test_iqn_oracle_matches_reference_golden():
    """Forward of the reference's live IQNNetwork and four steps of upstream's own IQNTrainer.learn (fixture:
    oracle/gen_golden.py:gen_iqn) against the numpy restatement."""
    from oracle import iqn_oracle as IO
    z = np.load(os.path.join(REPO, "tests", "golden", "iqn_s9_a5.npz"), allow_pickle=False)
    S, A, E, H, B, K, NP, NPP = (int(v) for v in z["meta"][:8])
    init = {k: v.astype(np.float64) for k, v in sub(z, "init/").items()}
    np.testing.assert_allclose(IO.cos_embed(z["probe_taus"].astype(np.float64), E), z["probe_embed"], atol=2e-5)
    np.testing.assert_allclose(IO.forward(init, z["probe_x"].astype(np.float64), z["probe_taus"].astype(np.float64)),
                               z["probe_z"], atol=1e-5)
    o = IO.IqnOracle(sub(z, "init/"), sub(z, "init_target/"), gamma=float(z["gamma"]), kappa=float(z["kappa"]),
                     lr=float(z["lr"]))
    for k in range(K):
        i = slice(k * B, (k + 1) * B)
        loss = o.learn(z["states"][i], z["actions"][i], z["rewards"][i], z["next_states"][i], z["dones"][i],
                       z["taus_prime"][k], z["taus_double_prime"][k])
        np.testing.assert_allclose(loss, z["loss"][k], rtol=LOSS_RTOL)
    _assert_params(o.P, sub(z, "final/"))
