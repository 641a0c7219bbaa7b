"""Costmap encoder (FasterNet.forward_cls, reference agent/fasternet.py:428-438) on the device against the
reference's golden vectors (tests/golden/fasternet_b5.npz, generated from /root/reference by
oracle/gen_golden.py), against the numpy oracle on fresh inputs, and — at the batch of BASELINE config 5 —
through size-independent properties.  Tolerance: 2e-5 of the largest feature magnitude (fp32, different
summation orders through 11 convolutions and 5 BatchNorms)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, checksum, load_golden

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
REL = 2e-5


def rel_err(got, ref):
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / np.abs(ref).max())


def build(seed, max_batch=8):
    from porl_amd.agent.fasternet import FasterNet
    torch.manual_seed(seed)
    return FasterNet(3, 256, max_batch=max_batch).to(DEV)


def test_weights_are_the_reference_initialisation():
    z, _ = load_golden("fasternet_b5")
    m = build(int(z["seed_model"]))
    for i, (k, v) in enumerate(m.state_dict().items()):
        if v.dim():
            cs, ref = checksum(v.cpu().numpy(), i), z["wsum." + k]
            # trunc_normal_'s erfinv_ is vectorised differently across host CPUs: ulp-level differences
            assert np.allclose(cs, ref, rtol=1e-6, atol=1e-9), k


def test_eval_and_train_forward_match_reference_golden():
    z, _ = load_golden("fasternet_b5")
    m = build(int(z["seed_model"]))
    m.eval()
    f = m(torch.from_numpy(z["states"].copy()).to(DEV))
    assert rel_err(f.cpu().numpy(), z["feat_eval"]) < REL
    m.train()
    f1 = m(torch.from_numpy(z["states"].copy()).to(DEV), drop_scale=torch.from_numpy(z["drop_scale1"]))
    f2 = m(torch.from_numpy(z["states2"].copy()).to(DEV), drop_scale=torch.from_numpy(z["drop_scale2"]))
    assert rel_err(f1.cpu().numpy(), z["feat_train1"]) < REL
    assert rel_err(f2.cpu().numpy(), z["feat_train2"]) < REL
    sd = m.state_dict()
    for k in sd:
        if k.endswith("running_var"):
            ref = z["stat_after." + k]
            assert rel_err(sd[k].cpu().numpy(), ref) < 1e-5, k
            mean_ref = z["stat_after." + k.replace("running_var", "running_mean")]
            got = sd[k.replace("running_var", "running_mean")].cpu().numpy()
            # channel means sit near zero: compare on the scale of the channel's standard deviation
            assert np.abs(got - mean_ref).max() < 1e-5 * np.sqrt(ref.max() / 0.19), k
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(z["stat_after." + k]) == 2
    m.eval()
    f3 = m(torch.from_numpy(z["states"].copy()).to(DEV))
    assert rel_err(f3.cpu().numpy(), z["feat_eval_after"]) < REL


def test_droppath_masks_come_from_the_cpu_generator_in_reference_order():
    z, _ = load_golden("fasternet_b5")
    m = build(int(z["seed_model"]))
    m.train()
    torch.manual_seed(int(z["seed_fwd"]))
    f1 = m(torch.from_numpy(z["states"].copy()).to(DEV))
    f2 = m(torch.from_numpy(z["states2"].copy()).to(DEV))
    assert rel_err(f1.cpu().numpy(), z["feat_train1"]) < REL
    assert rel_err(f2.cpu().numpy(), z["feat_train2"]) < REL


def test_against_oracle_on_fresh_inputs_and_state_is_clamped_in_place():
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import fasternet_oracle as FO
    m = build(5)
    sd = {k: v.cpu().numpy().copy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(3)
    st = np.empty((3, 362), dtype=np.float32)
    st[:, :360] = rng.uniform(0.2, 3.9, size=(3, 360))
    st[:, 360:] = rng.uniform(-3, 3, size=(3, 2))
    st[1, 7] = 9.0
    scale = np.array([[1, 1, 1], [0, 1 / 0.95, 1 / 0.95], [1 / 0.9, 0, 1 / 0.9]], dtype=np.float32)
    stats = {k: v.copy() for k, v in sd.items() if "running" in k}
    x_ref = st.copy()
    ref = FO.forward(sd, stats, x_ref, True, scale)
    m.train()
    x = torch.from_numpy(st.copy()).to(DEV)
    got = m(x, drop_scale=torch.from_numpy(scale))
    assert rel_err(got.cpu().numpy(), ref) < REL
    assert np.array_equal(x.cpu().numpy(), x_ref) and x_ref[1, 7] == 0.0       # costmap.py:17 side effect


def test_full_batch_properties():
    """B=512 (BASELINE config 5; 2.9 M stage-1 positions, GEMM operands split at the 2 GiB descriptor reach):
    eval-mode features of a sample do not depend on its batch neighbours, bit for bit; train mode is
    deterministic; a dropped sample of a block equals the block-skipped network."""
    B = 512
    m = build(1, max_batch=B)
    rng = np.random.default_rng(9)
    st = np.empty((B, 362), dtype=np.float32)
    st[:, :360] = rng.uniform(0.2, 3.9, size=(B, 360))
    st[:, 360:] = rng.uniform(-3, 3, size=(B, 2))
    m.eval()
    big = m(torch.from_numpy(st.copy()).to(DEV))
    assert torch.isfinite(big).all()
    pick = [0, 1, 255, 256, 484, 485, 486, 511]
    small = m(torch.from_numpy(st[pick].copy()).to(DEV))
    assert torch.equal(big[pick], small)
    m.train()
    scale = torch.ones(3, B)
    scale[1, ::3] = 0
    scale[2, 1::3] = 0
    before = {k: v.clone() for k, v in m.state_dict().items()}
    a = m(torch.from_numpy(st.copy()).to(DEV), drop_scale=scale)
    m.load_state_dict(before)
    b = m(torch.from_numpy(st.copy()).to(DEV), drop_scale=scale)
    assert torch.equal(a, b)
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0


def test_errors():
    from porl_amd import _native as N
    from porl_amd.agent.fasternet import FasterNet
    m = FasterNet(3, 256, max_batch=2)
    with pytest.raises(N.NativeError):
        m(torch.zeros(1, 362))                                   # no CPU path
    m = m.to(DEV)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 100, device=DEV))
    with pytest.raises(RuntimeError):
        m(torch.zeros(3, 362, device=DEV))                       # > max_batch
    with pytest.raises(NotImplementedError):
        FasterNet(3, 256, fork_feat=True)


def test_sorl_update_with_encoder_backbone_matches_reference_golden():
    """SORL.update with backbone=FasterNet(3, 256) (sorl.py:78-128, sorl_train.py:29-33): three joint updates,
    DropPath masks drawn by the drop-in itself from the seeded CPU generator in the reference's order.
    Features carry ~1e-6 relative noise (different conv summation orders), which the 64-wide heads pass on."""
    from types import SimpleNamespace
    from porl_amd.agent.fasternet import FasterNet
    from porl_amd.agent.sorl import SORL
    z, _ = load_golden("sorl_enc_b6")
    B, K, H, L, A, F = (int(v) for v in z["meta"])
    torch.manual_seed(int(z["seed_model"]))
    backbone = FasterNet(3, F, max_batch=B)
    args = SimpleNamespace(state_size=362, feature_dim=F, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=A,
                           max_batch=B)
    agent = SORL(args, max_steps=50, tau=float(z["tau"]), alpha=float(z["alpha"]), device=DEV, backbone=backbone)
    sd = agent.state_dict()
    for k in z.files:
        if k.startswith("init."):
            assert np.array_equal(sd[k[5:]].cpu().numpy(), z[k]), k         # heads: same generator stream
    torch.manual_seed(int(z["seed_fwd"]))
    for k in range(K):
        t = lambda n: torch.from_numpy(z[f"{n}{k}"].copy()).to(DEV)
        vl, gl = agent.update(t("s"), t("a"), t("r"), t("s2"), t("d"))
        np.testing.assert_allclose([vl, gl], z["losses"][k], rtol=5e-5)
    sd = agent.state_dict()
    worst = 0.0
    for k in z.files:
        if not k.startswith("final."):
            continue
        got, ref = sd[k[6:]].cpu().numpy().astype(np.float64), z[k]
        if "num_batches" in k:
            assert int(got) == int(ref) == 2 * K
        elif "running_mean" in k:
            assert np.abs(got - ref).max() < 1e-5, k
        elif "running_var" in k:
            assert rel_err(got, ref) < 1e-5, k
        else:
            worst = max(worst, float(np.abs(got - ref).max()))
    assert worst < 2e-5, worst                                               # lr 1e-4, 3 Adam steps
    act = agent.select_action(torch.from_numpy(z["s0"].copy()).to(DEV))
    assert act.shape == (B, A) and np.isfinite(act).all()


@pytest.mark.parametrize("embed,depths,n_div,mlp_ratio,B", [(64, (2, 1), 4, 2.0, 2), (32, (1, 1), 2, 3.0, 1)])
def test_other_encoder_shapes_against_oracle(embed, depths, n_div, mlp_ratio, B):
    """Shapes the reference script does not build (other widths / depths / n_div, a one-sample batch): the
    patch-matrix fallback of the partial conv and the general block loop, train mode with a dropped sample."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import fasternet_oracle as FO
    from porl_amd.agent.fasternet import FasterNet
    torch.manual_seed(11)
    m = FasterNet(3, 40, embed_dim=embed, depths=depths, n_div=n_div, mlp_ratio=mlp_ratio, feature_dim=96, max_batch=4).to(DEV)
    sd = {k: v.cpu().numpy().copy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(embed)
    st = np.empty((B, 362), dtype=np.float32)
    st[:, :360] = rng.uniform(0.2, 3.9, size=(B, 360))
    st[:, 360:] = rng.uniform(-3, 3, size=(B, 2))
    nb = sum(depths)
    scale = np.ones((nb, B), dtype=np.float32)
    scale[-1, 0] = 0.0
    stats = {k: v.copy() for k, v in sd.items() if "running" in k}
    ref = FO.forward(sd, stats, st.copy(), True, scale, depths=depths, n_div=n_div)
    m.train()
    got = m(torch.from_numpy(st.copy()).to(DEV), drop_scale=torch.from_numpy(scale))
    if np.abs(ref).max() > 1e-8:
        assert rel_err(got.cpu().numpy(), ref) < REL
    else:   # one sample: the last BatchNorm centres every channel over the positions, so the pooled features vanish
        assert np.abs(got.cpu().numpy()).max() < 1e-8
    m.eval()
    ref_eval = FO.forward(sd, stats, st.copy(), False, depths=depths, n_div=n_div)
    got_eval = m(torch.from_numpy(st.copy()).to(DEV))
    assert rel_err(got_eval.cpu().numpy(), ref_eval) < REL


def test_sparse_patch_embedding_equals_the_dense_one():
    """PatchEmbed + BatchNorm computed from the lidar state through per-patch pixel masks (no costmap image) against
    rasterise + dense 4x4 convolution + column statistics: same sums in the same order, so features agree to the
    last bits of the BatchNorm statistics (1e-6 relative), incl. goal-cross edge cases and > 8 clamping."""
    from porl_amd import engine as E
    z, _ = load_golden("costmap_b24")
    st = z["state_in"].copy()
    outs = []
    for dense in (0, 1):
        try:
            E.tune_set("enc_dense_patch", dense)
            m = build(2, max_batch=24)
            m.train()
            x = torch.from_numpy(st.copy()).to(DEV)
            f = m(x, drop_scale=torch.ones(3, 24))
            outs.append((f.cpu().numpy(), x.cpu().numpy(), {k: v.cpu().numpy() for k, v in m.state_dict().items() if "running" in k}))
        finally:
            E.tune_set("enc_dense_patch", 0)
    assert rel_err(outs[0][0], outs[1][0].astype(np.float64)) < 1e-6
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][1], z["state_after"])
    for k in outs[0][2]:
        np.testing.assert_allclose(outs[0][2][k], outs[1][2][k], rtol=1e-6, atol=1e-12)


def test_merge_conv_gathers_its_patches_in_the_gemm():
    """PatchMerging through the GEMM's grouped-row operand addressing (no space_to_depth copy) against the
    materialised patch matrix: same k order, so the features are bit-identical; B=40 spans image-row groups,
    samples and a ragged last tile."""
    from porl_amd import engine as E
    rng = np.random.default_rng(5)
    st = np.empty((40, 362), dtype=np.float32)
    st[:, :360] = rng.uniform(0.2, 3.9, size=(40, 360))
    st[:, 360:] = rng.uniform(-3, 3, size=(40, 2))
    outs = []
    for s2d in (0, 1):
        try:
            E.tune_set("enc_s2d", s2d)
            m = build(3, max_batch=40)
            m.train()
            outs.append(m(torch.from_numpy(st.copy()).to(DEV), drop_scale=torch.ones(3, 40)))
        finally:
            E.tune_set("enc_s2d", 0)
    assert torch.equal(outs[0], outs[1])


def test_batchnorm_relu_folded_into_w2_staging_equals_the_separate_sweep():
    """W1 -> BN -> ReLU -> W2: the normalised hidden activation is made inside W2's operand staging (fmaf + max, the
    arithmetic of bn_apply_kernel) instead of by a read+write sweep: features are bit-identical."""
    from porl_amd import engine as E
    rng = np.random.default_rng(8)
    st = np.empty((12, 362), dtype=np.float32)
    st[:, :360] = rng.uniform(0.2, 3.9, size=(12, 360))
    st[:, 360:] = rng.uniform(-3, 3, size=(12, 2))
    outs = []
    for sweep in (0, 1):
        try:
            E.tune_set("enc_bn_sweep", sweep)
            m = build(4, max_batch=12)
            m.train()
            a = m(torch.from_numpy(st.copy()).to(DEV), drop_scale=torch.ones(3, 12))
            m.eval()
            b = m(torch.from_numpy(st.copy()).to(DEV))
            outs.append((a, b))
        finally:
            E.tune_set("enc_bn_sweep", 0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("n_ang,n_dist,B", [(84, 84, 5), (40, 64, 3)])
def test_parametrised_costmap_geometry_against_oracle(n_ang, n_dist, B):
    """BASELINE config 5 names an 84 x 84 costmap; the reference can only build 360 x 256 (util/costmap.py:12,24), so
    there is no reference golden at this size — parity here is against the numpy oracle (pinned to the reference at
    360 x 256 by tests/test_oracle_golden.py) with the two constants replaced.  84 x 84 -> 21 x 21 patches -> the 2x2s2
    merge floors to 10 x 10 (the Conv2d rule), exercising the patch-matrix merge path and the generic partial conv."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import fasternet_oracle as FO
    from porl_amd.agent.fasternet import FasterNet
    from porl_amd.util.costmap import state2costmap
    torch.manual_seed(21)
    m = FasterNet(3, 256, max_batch=8, angle_bins=n_ang, dist_bins=n_dist).to(DEV)
    sd = {k: v.cpu().numpy().copy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(n_ang)
    st = np.empty((B, n_ang + 2), dtype=np.float32)
    st[:, :n_ang] = rng.uniform(0.2, 3.9, size=(B, n_ang))
    st[:, n_ang:] = rng.uniform(-3, 3, size=(B, 2))
    st[0, 3] = 9.5                                            # > 8: reads as 0 and is zeroed in place
    # the rasteriser itself at this geometry
    img = state2costmap(torch.from_numpy(st.copy()).to(DEV), n_ang, n_dist).cpu().numpy()
    np.testing.assert_array_equal(img, FO.state2costmap(st.copy(), n_ang, n_dist))
    assert img.shape == (B, 3, n_ang, n_dist) and img[:, 0].sum() > B * n_ang * 0.8
    stats = {k: v.copy() for k, v in sd.items() if "running" in k}
    scale = np.ones((3, B), dtype=np.float32)
    scale[1, 1] = 0.0
    ref = FO.forward(sd, stats, st.copy(), True, scale, angle_bins=n_ang, dist_bins=n_dist)
    m.train()
    x = torch.from_numpy(st.copy()).to(DEV)
    got = m(x, drop_scale=torch.from_numpy(scale))
    assert got.shape == (B, 256) and rel_err(got.cpu().numpy(), ref) < REL
    assert float(x[0, 3]) == 0.0
    m.eval()
    ref_eval = FO.forward(sd, stats, st.copy(), False, angle_bins=n_ang, dist_bins=n_dist)
    assert rel_err(m(torch.from_numpy(st.copy()).to(DEV)).cpu().numpy(), ref_eval) < REL
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 362, device=DEV))                    # the state width follows the geometry


def test_por_with_encoder_backbone_matches_reference_golden():
    """POR(backbone=FasterNet) (agent/por.py:46-57,75-79): heads on the 256 features, goal policy regressing the raw
    362-wide next state; two updates from the reference's seeds (train-mode BatchNorm, DropPath from the CPU stream)."""
    from types import SimpleNamespace
    from porl_amd.agent.fasternet import FasterNet
    from porl_amd.agent.por import POR
    z, _ = load_golden("por_enc_b4")
    B, K, H, L, F = (int(v) for v in z["meta"])
    torch.manual_seed(int(z["seed_model"]))
    backbone = FasterNet(3, F, max_batch=B)
    args = SimpleNamespace(state_size=362, feature_dim=F, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=2,
                           max_batch=B)
    agent = POR(args, max_steps=50, tau=float(z["tau"]), alpha=float(z["alpha"]), device=DEV, backbone=backbone)
    sd = agent.state_dict()
    for k in z.files:
        if k.startswith("init."):
            assert np.array_equal(sd[k[5:]].cpu().numpy(), z[k]), k
    torch.manual_seed(int(z["seed_fwd"]))
    for k in range(K):
        t = lambda n: torch.from_numpy(z[f"{n}{k}"].copy()).to(DEV)
        vl, gl = agent.por_residual_update(t("s"), t("s2"), t("r"), t("d"))
        np.testing.assert_allclose([vl, gl], z["losses"][k], rtol=5e-5)
    sd = agent.state_dict()
    worst = max(float(np.abs(sd[k[6:]].cpu().numpy().astype(np.float64) - z[k]).max()) for k in z.files if k.startswith("final."))
    assert worst < 2e-5, worst
    with pytest.raises(NotImplementedError):
        agent.update_from_replay(None, B)


@pytest.mark.parametrize("mode", ["bf16", "bf16_operands"])
@pytest.mark.parametrize("n_ang,n_dist", [(360, 256), (84, 84)])
def test_bf16_modes_stay_close_to_the_fp32_path(n_ang, n_dist, mode):
    """compute_dtype="bf16" (BASELINE config 5's wording; the reference has no bf16 path): bf16 activations in HBM behind
    the patch embedding, partial conv / fused MLP blocks / merge on the bf16 matrix pipe with fp32 accumulation
    (csrc/encoder_bf16.hpp); "bf16_operands" (round 2's mode): fp32 tensors in memory, operands of the 1x1 / merge
    convolutions rounded to bf16 on their way into LDS.  Tolerance defined by the build and stated here: features within
    3e-2 of the largest fp32 feature magnitude, running statistics within 1e-2 relative, against the fp32 path on
    identical weights, inputs and DropPath masks (a dropped sample included) — the fp32 path itself is pinned to the
    reference at 2e-5."""
    from porl_amd.agent.fasternet import FasterNet
    B = 6
    rng = np.random.default_rng(3)
    st = np.empty((B, n_ang + 2), dtype=np.float32)
    st[:, :n_ang] = rng.uniform(0.2, 3.9, size=(B, n_ang))
    st[:, n_ang:] = rng.uniform(-3, 3, size=(B, 2))
    outs = {}
    scale = torch.ones(3, B)
    scale[1, 2] = 0.0
    scale[2, 4] = 1 / 0.9
    for dt in ("fp32", mode):
        torch.manual_seed(5)
        m = FasterNet(3, 256, max_batch=8, angle_bins=n_ang, dist_bins=n_dist, compute_dtype=dt).to(DEV)
        m.train()
        tr = m(torch.from_numpy(st.copy()).to(DEV), drop_scale=scale).cpu().numpy()
        m.eval()
        ev = m(torch.from_numpy(st.copy()).to(DEV)).cpu().numpy()
        outs[dt] = (tr, ev, {k: v.cpu().numpy() for k, v in m.state_dict().items() if "running" in k})
    for i in (0, 1):
        err = rel_err(outs[mode][i], outs["fp32"][i].astype(np.float64))
        assert 0 < err < 3e-2, err                       # > 0: the bf16 kernels really ran
    for k, v in outs["fp32"][2].items():
        if "running_var" in k:
            assert rel_err(outs[mode][2][k], v.astype(np.float64)) < 1e-2, k
        elif "running_mean" in k:                        # channel means sit near zero: on the scale of the channel's deviation
            sd = np.sqrt(outs["fp32"][2][k.replace("running_mean", "running_var")].max() / 0.1)
            assert np.abs(outs[mode][2][k] - v).max() < 2e-2 * sd, k


@pytest.mark.parametrize("n_ang,n_dist,B", [(360, 256, 7), (84, 84, 5), (40, 64, 3)])
def test_partial_conv_on_the_matrix_pipe_agrees_with_the_direct_kernel(n_ang, n_dist, B):
    """Partial_conv3 (fasternet.py:110-138) as an implicit GEMM on the fp32 matrix pipe (round 3; measured slower than the
    direct kernel and therefore opt-in: porl_tune_set("enc_pconv_mfma", 1)) against the direct vector-ALU kernel /
    patch-matrix path (the default): the same exact-fp32
    products summed in another order — features agree to 2e-6 of their largest magnitude, in train mode (statistics
    included) and in eval mode, for the reference's geometry and for odd row widths (21- and 10-wide rows: tiles that span
    several image rows and end inside one)."""
    from porl_amd import engine as E
    from porl_amd.agent.fasternet import FasterNet
    rng = np.random.default_rng(n_ang)
    st = np.empty((B, n_ang + 2), dtype=np.float32)
    st[:, :n_ang] = rng.uniform(0.2, 3.9, size=(B, n_ang))
    st[:, n_ang:] = rng.uniform(-3, 3, size=(B, 2))
    outs = []
    for mfma in (1, 0):
        try:
            E.tune_set("enc_pconv_mfma", mfma)
            torch.manual_seed(9)
            m = FasterNet(3, 256, max_batch=8, angle_bins=n_ang, dist_bins=n_dist).to(DEV)
            m.train()
            scale = torch.ones(3, B)
            scale[0, 1] = 0.0
            a = m(torch.from_numpy(st.copy()).to(DEV), drop_scale=scale).cpu().numpy()
            m.eval()
            b = m(torch.from_numpy(st.copy()).to(DEV)).cpu().numpy()
            outs.append((a, b, {k: v.cpu().numpy() for k, v in m.state_dict().items() if "running" in k}))
        finally:
            E.tune_set("enc_pconv_mfma", 0)
    for i in (0, 1):
        assert rel_err(outs[0][i], outs[1][i].astype(np.float64)) < 2e-6
    for k, v in outs[1][2].items():
        np.testing.assert_allclose(outs[0][2][k], v, rtol=1e-5, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("n_ang,n_dist,B", [(360, 256, 7), (84, 84, 5)])
def test_row_product_schedules_are_bit_identical(n_ang, n_dist, B):
    """The encoder's row products run on single-buffered 64x64 blocks (round 3, eight blocks per CU); the MFMA-paced
    double-buffered schedule and the 128x96 tile for the N = 96 product stay behind porl_tune_set.  All three walk a
    tile's reduction in the same order on the same accumulators: features, and the running statistics a train-mode
    forward leaves behind, are equal bit for bit."""
    from porl_amd import engine as E
    from porl_amd.agent.fasternet import FasterNet
    rng = np.random.default_rng(n_ang + 1)
    st = np.empty((B, n_ang + 2), dtype=np.float32)
    st[:, :n_ang] = rng.uniform(0.2, 3.9, size=(B, n_ang))
    st[:, n_ang:] = rng.uniform(-3, 3, size=(B, 2))
    outs = []
    for sb, n96 in ((1, 0), (0, 0), (0, 1), (1, 1)):
        try:
            E.tune_set("enc_gemm_sb", sb)
            E.tune_set("enc_tile_n96", n96)
            torch.manual_seed(9)
            m = FasterNet(3, 256, max_batch=8, angle_bins=n_ang, dist_bins=n_dist).to(DEV)
            m.train()
            scale = torch.ones(3, B)
            scale[0, 1] = 0.0
            a = m(torch.from_numpy(st.copy()).to(DEV), drop_scale=scale).cpu()
            m.eval()
            b = m(torch.from_numpy(st.copy()).to(DEV)).cpu()
            outs.append((a, b, {k: v.cpu() for k, v in m.state_dict().items() if "running" in k}))
        finally:
            E.tune_set("enc_gemm_sb", 1)
            E.tune_set("enc_tile_n96", 0)
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
        for k, v in outs[0][2].items():
            assert torch.equal(o[2][k], v), k
