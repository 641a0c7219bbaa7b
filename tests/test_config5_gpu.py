"""BASELINE config 5 as it is worded — "SORL (sorl_train.py) with fasternet.py occupancy-grid encoder (84x84 costmap
input), batch=512, bf16" — exercised at its full batch:

  (i)   SORL.update + FasterNet(angle_bins=84, dist_bins=84) in fp32, B=512, three updates, against the numpy oracle
        (oracle/fasternet_oracle.py for the two encoder forwards, oracle/por_oracle.py:sorl_oracle for the heads' step),
        both run in fp64 as the yardstick;
  (ii)  the same in compute_dtype="bf16" against that oracle at the bf16 bound stated below, plus the size-independent
        properties (per-sample independence in eval mode, determinism in train mode);
  (iii) the reference's own 360x256 image in bf16 at B=512 through the properties.

What pins what: the reference can only rasterise 360x256 (util/costmap.py:12,24) and has no bf16 path, so neither (i)
nor (ii) has a reference golden — "parity unpinned" for those two modes beyond this chain: the oracle is pinned to the
reference at 360x256 / fp32 (tests/test_oracle_golden.py), the 84x84 oracle is that oracle with the two geometry
constants replaced, and the HIP fp32 path is compared with it here at full batch.

bf16 bound (this build's, stated): features within 3e-2 of the largest fp32 feature magnitude; losses within 2e-2
relative; parameters after three Adam steps within 6.2e-4 (Adam's first steps move a weight by ~lr = 1e-4 whatever the
gradient's size, so a gradient sign decided by bf16 noise costs up to 2*lr per step on that weight: 3 x 2e-4; measured
4.8e-4) with a MEAN absolute deviation below 3e-5 (the assertion that carries the information).
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
B, F, H, L, A = 512, 256, 512, 2, 2          # sorl_train.py:93 hidden_dim=512, feature_dim=256


def _states(rng, n, n_ang):
    st = np.empty((n, n_ang + 2), dtype=np.float32)
    st[:, :n_ang] = rng.uniform(0.2, 3.9, size=(n, n_ang))
    st[:, n_ang:] = rng.uniform(-3, 3, size=(n, 2))
    return st


def _agent(n_ang, n_dist, dtype, seed=3):
    from porl_amd.agent.fasternet import FasterNet
    from porl_amd.agent.sorl import SORL
    torch.manual_seed(seed)
    backbone = FasterNet(3, F, max_batch=B, angle_bins=n_ang, dist_bins=n_dist, compute_dtype=dtype)
    args = SimpleNamespace(state_size=n_ang + 2, feature_dim=F, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=A,
                           max_batch=B)
    return SORL(args, max_steps=100, tau=0.9, alpha=3.0, device=DEV, backbone=backbone)


def _batches(n_ang, K, seed=17):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(K):
        s, s2 = _states(rng, B, n_ang), _states(rng, B, n_ang)
        s[5, 3] = 9.5                                                  # > 8: read as 0 and zeroed in place (costmap.py:17)
        a = rng.uniform(-1, 1, size=(B, A)).astype(np.float32)
        r = rng.normal(size=B).astype(np.float32)
        d = (rng.uniform(size=B) < 0.1).astype(np.float32)
        out.append((s, a, r, s2, d))
    return out


_ORACLE_CACHE = {}


def _oracle_run(n_ang, n_dist, K, init_sd, drop_scales, batches):
    """Three SORL updates with the encoder in the numpy oracle, fp64 throughout (encoder activations, heads, Adam)."""
    key = (n_ang, n_dist, K)
    if key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import fasternet_oracle as FO
    import oracle.por_oracle as O
    O.set_precision(np.float64)
    try:
        enc = {k[len("backbone."):]: v for k, v in init_sd.items() if k.startswith("backbone.")}
        stats = {k: v.copy() for k, v in enc.items() if "running" in k}
        heads = {k: np.asarray(v, np.float64) for k, v in init_sd.items() if not k.startswith("backbone.")}
        o = O.sorl_oracle(heads, F, H, L, tau=0.9, alpha=3.0, max_steps=100)
        losses, feats = [], []
        for k, (s, a, r, s2, d) in enumerate(batches):
            f1 = FO.forward(enc, stats, s.copy(), True, drop_scales[2 * k], angle_bins=n_ang, dist_bins=n_dist)
            f2 = FO.forward(enc, stats, s2.copy(), True, drop_scales[2 * k + 1], angle_bins=n_ang, dist_bins=n_dist)
            feats.append((f1, f2))
            losses.append(o.sorl_update(f1, a, r, f2, d))
        res = (np.array(losses), {k: v.copy() for k, v in o.P.items()}, stats, feats)
    finally:
        O.set_precision(np.float32)
    _ORACLE_CACHE[key] = res
    return res


def _run_device(n_ang, n_dist, dtype, K, batches):
    agent = _agent(n_ang, n_dist, dtype)
    init = {k: v.detach().cpu().numpy().copy() for k, v in agent.state_dict().items()}
    torch.manual_seed(99)
    scales = []
    for _ in range(K):                                                  # the draws agent.update will make, in its order
        scales += [agent.backbone.draw_drop_scale(B).numpy().copy(), agent.backbone.draw_drop_scale(B).numpy().copy()]
    torch.manual_seed(99)
    losses = []
    for s, a, r, s2, d in batches:
        t = lambda x: torch.from_numpy(x.copy()).to(DEV)
        ts = t(s)
        losses.append(agent.update(ts, t(a), t(r), t(s2), t(d)))
        assert float(ts[5, 3]) == 0.0
    final = {k: v.detach().cpu().numpy() for k, v in agent.state_dict().items()}
    return agent, init, scales, np.array(losses), final


def test_sorl_update_with_84x84_encoder_fp32_b512_vs_oracle():
    K = 3
    batches = _batches(84, K)
    agent, init, scales, losses, final = _run_device(84, 84, "fp32", K, batches)
    ref_losses, ref_P, ref_stats, _ = _oracle_run(84, 84, K, init, scales, batches)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
    worst, n_far, n_all = 0.0, 0, 0
    for k, ref in ref_P.items():
        err = np.abs(final[k].astype(np.float64) - ref)
        worst = max(worst, float(err.max()))
        n_far += int((err > 2e-6).sum())
        n_all += err.size
    # the form of tests/test_por_gpu.py:_cmp_params_robust: fp32 ReLU-mask flips near zero move single weight rows by
    # up to ~lr/10 through Adam; nothing beyond the 1e-5 bar, all but 1e-3 of the elements within 2e-6
    assert worst <= 1e-5, worst
    assert n_far <= 1e-3 * n_all, (n_far, n_all)
    for k, ref in ref_stats.items():
        got = final["backbone." + k]
        if "num_batches" in k:
            assert int(got) == int(ref) == 2 * K
        elif "running_var" in k:
            np.testing.assert_allclose(got, ref, rtol=1e-5, err_msg=k)
        else:
            assert np.abs(got - ref).max() < 1e-5 * max(1.0, np.sqrt(ref_stats[k.replace("mean", "var")].max() / 0.19)), k


def test_sorl_update_with_84x84_encoder_bf16_b512_vs_fp32_oracle_and_properties():
    K = 3
    batches = _batches(84, K)
    agent, init, scales, losses, final = _run_device(84, 84, "bf16", K, batches)
    # the fp32 agent of the previous test has the same seed, hence the same initial state: the oracle run is shared
    ref_losses, ref_P, ref_stats, feats = _oracle_run(84, 84, K, init, scales, batches)
    rel = np.abs(losses / ref_losses - 1).max()
    assert 0 < rel < 2e-2, rel                                          # > 0: the bf16 kernels really ran
    worst, mean_dev, n_all = 0.0, 0.0, 0
    for k, ref in ref_P.items():
        err = np.abs(final[k].astype(np.float64) - ref)
        worst = max(worst, float(err.max()))
        mean_dev += float(err.sum())
        n_all += err.size
    assert worst <= 6.2e-4, worst
    assert mean_dev / n_all <= 3e-5, mean_dev / n_all
    for k, ref in ref_stats.items():
        if "running_var" in k:
            np.testing.assert_allclose(final["backbone." + k], ref, rtol=1e-2, err_msg=k)
    _encoder_properties(agent.backbone, 84)


def _encoder_properties(enc, n_ang):
    rng = np.random.default_rng(5)
    st = _states(rng, B, n_ang)
    enc.eval()
    big = enc(torch.from_numpy(st.copy()).to(DEV))
    assert torch.isfinite(big).all() and float(big.abs().max()) > 0
    pick = [0, 1, 255, 256, 484, 485, 486, 511]
    small = enc(torch.from_numpy(st[pick].copy()).to(DEV))
    assert torch.equal(big[pick], small)                                # eval: a sample does not see its batch neighbours
    enc.train()
    scale = torch.ones(3, B)
    scale[1, ::3] = 0
    scale[2, 1::3] = 0
    before = {k: v.clone() for k, v in enc.state_dict().items()}
    a = enc(torch.from_numpy(st.copy()).to(DEV), drop_scale=scale)
    after_a = {k: v.clone() for k, v in enc.state_dict().items()}
    enc.load_state_dict(before)
    b = enc(torch.from_numpy(st.copy()).to(DEV), drop_scale=scale)
    assert torch.equal(a, b)                                            # train: deterministic, statistics included
    for k, v in enc.state_dict().items():
        assert torch.equal(v, after_a[k]), k
    assert torch.isfinite(a).all()
    return big


def test_encoder_360x256_bf16_b512_properties_and_distance_to_fp32():
    """The reference's own image size in the bf16 mode at config 5's batch: properties, then the train-mode features of
    fresh bf16 / fp32 encoders (same seed, inputs, DropPath masks) at the stated 3e-2 feature bound."""
    from porl_amd.agent.fasternet import FasterNet
    feats = {}
    for dt in ("bf16", "fp32"):
        torch.manual_seed(4)
        enc = FasterNet(3, F, max_batch=B, angle_bins=360, dist_bins=256, compute_dtype=dt).to(DEV)
        enc.train()
        st = _states(np.random.default_rng(6), B, 360)
        feats[dt] = enc(torch.from_numpy(st.copy()).to(DEV), drop_scale=torch.ones(3, B)).cpu().numpy().astype(np.float64)
        if dt == "bf16":
            _encoder_properties(enc, 360)
        del enc
    err = np.abs(feats["bf16"] - feats["fp32"]).max() / np.abs(feats["fp32"]).max()
    assert 0 < err < 3e-2, err
