"""Parity of the fp32 MFMA GEMM (through the C ABI) against float64 numpy.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from porl_amd.engine import gemm_f32, adam_ema, gather_rows

DEV = "cuda"
MODES = {"NT": 0, "NN": 1, "TN": 2}


def _ref(mode, A, B):
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    if mode == "NT":
        return A64 @ B64.T
    if mode == "NN":
        return A64 @ B64
    return A64.T @ B64


def _operands(mode, M, N, K, rng, lda_pad=0, ldb_pad=0):
    shapeA = (M, K) if mode in ("NT", "NN") else (K, M)
    shapeB = (N, K) if mode == "NT" else (K, N)
    A = rng.standard_normal(shapeA).astype(np.float32)
    B = rng.standard_normal(shapeB).astype(np.float32)
    Ap = np.zeros((shapeA[0], shapeA[1] + lda_pad), np.float32); Ap[:, :shapeA[1]] = A
    Bp = np.zeros((shapeB[0], shapeB[1] + ldb_pad), np.float32); Bp[:, :shapeB[1]] = B
    return A, B, Ap, Bp


def _tol(K):
    # fp32 fmaf chain vs exact: ~1e-7 * sum|a*b| ~ 1e-7 * K * E|ab|; generous factor
    return 4e-7 * max(K, 8) + 1e-6


@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 192, 1024), (100, 70, 60), (37, 300, 129), (1, 5, 3)])
def test_gemm_plain(mode, tile, shape):
    M, N, K = shape
    rng = np.random.default_rng(hash((mode, tile, shape)) % (2 ** 32))
    A, B, Ap, Bp = _operands(mode, M, N, K, rng)
    At, Bt = torch.from_numpy(Ap).to(DEV), torch.from_numpy(Bp).to(DEV)
    C = torch.full((M, N + 3), 7.0, device=DEV)          # ldc > N: padding must stay untouched
    gemm_f32(MODES[mode], At, Bt, M, N, K, Ap.shape[1], Bp.shape[1], C, N + 3, tile=tile)
    got = C.cpu().numpy()
    np.testing.assert_allclose(got[:, :N], _ref(mode, A, B), atol=_tol(K) * 4, rtol=1e-5)
    assert np.all(got[:, N:] == 7.0)


@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
def test_gemm_unaligned_leading_dims(mode):
    # odd leading dimensions force the dword-load kernel variant
    M, N, K = 130, 61, 362
    rng = np.random.default_rng(5)
    A, B, Ap, Bp = _operands(mode, M, N, K, rng, lda_pad=1, ldb_pad=3)
    At, Bt = torch.from_numpy(Ap).to(DEV), torch.from_numpy(Bp).to(DEV)
    C = torch.zeros((M, N), device=DEV)
    gemm_f32(MODES[mode], At, Bt, M, N, K, Ap.shape[1], Bp.shape[1], C, N)
    np.testing.assert_allclose(C.cpu().numpy(), _ref(mode, A, B), atol=_tol(K) * 4, rtol=1e-5)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_bias_act_mask(act):
    M, N, K = 200, 136, 96
    rng = np.random.default_rng(11 + act)
    A, B, Ap, Bp = _operands("NT", M, N, K, rng)
    bias = rng.standard_normal(N).astype(np.float32)
    mask = rng.standard_normal((M, N)).astype(np.float32)
    C = torch.zeros((M, N), device=DEV)
    gemm_f32(0, torch.from_numpy(Ap).to(DEV), torch.from_numpy(Bp).to(DEV), M, N, K, K, K, C, N,
             bias=torch.from_numpy(bias).to(DEV), act=act, mask=torch.from_numpy(mask).to(DEV), ldmask=N)
    ref = _ref("NT", A, B) + bias
    ref = np.maximum(ref, 0) if act == 1 else (np.tanh(ref) if act == 2 else ref)
    ref = ref * (mask > 0)
    np.testing.assert_allclose(C.cpu().numpy(), ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("mode,shape,sk", [("NT", (1024, 60, 1024), 16), ("TN", (60, 1024, 1024), 8),
                                           ("TN", (1024, 60, 1000), 7), ("NN", (96, 64, 300), 3)])
def test_gemm_splitk(mode, shape, sk):
    M, N, K = shape
    rng = np.random.default_rng(3)
    A, B, Ap, Bp = _operands(mode, M, N, K, rng)
    C = torch.zeros((M, N), device=DEV)
    slab = torch.empty(sk * M * N, device=DEV)
    bias = rng.standard_normal(N).astype(np.float32)
    gemm_f32(MODES[mode], torch.from_numpy(Ap).to(DEV), torch.from_numpy(Bp).to(DEV), M, N, K, Ap.shape[1],
             Bp.shape[1], C, N, bias=torch.from_numpy(bias).to(DEV), splitk=sk, slab=slab)
    np.testing.assert_allclose(C.cpu().numpy(), _ref(mode, A, B) + bias, atol=_tol(K) * 4, rtol=1e-5)


def test_gemm_exact_small_integers():
    # integer-valued operands: every product and partial sum is exact in fp32 -> bitwise equality.
    # Asymmetric B catches a transposed C write (cdna_hip_programming.md §3).
    M, N, K = 128, 128, 32
    rng = np.random.default_rng(0)
    A = rng.integers(-4, 5, (M, K)).astype(np.float32)
    B = rng.integers(-4, 5, (N, K)).astype(np.float32)
    C = torch.zeros((M, N), device=DEV)
    gemm_f32(0, torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV), M, N, K, K, K, C, N)
    assert np.array_equal(C.cpu().numpy(), A @ B.T)


def test_adam_ema_matches_torch_adam():
    n = 4096 + 8
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g); tgt0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=3e-4)
    p, m, v, t = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), tgt0.to(DEV)
    tgt_ref = tgt0.clone()
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** (-step))
        ref_p.grad = grad.clone()
        opt.step()
        tgt_ref.mul_(1 - 0.005).add_(ref_p.data, alpha=0.005)
        adam_ema(p, grad.to(DEV), m, v, t, 3e-4, step, ema_beta=0.005)
    np.testing.assert_allclose(p.cpu().numpy(), ref_p.data.numpy(), atol=1e-7, rtol=1e-6)
    np.testing.assert_allclose(t.cpu().numpy(), tgt_ref.numpy(), atol=1e-7, rtol=1e-6)
    st = opt.state[ref_p]
    np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].numpy(), atol=1e-9, rtol=1e-5)
    np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].numpy(), atol=1e-12, rtol=1e-5)


@pytest.mark.parametrize("width", [124, 60, 1, 2, 33])
def test_gather_rows_bit_exact(width):
    rng = np.random.default_rng(width)
    rows = rng.standard_normal((5000, width)).astype(np.float32)
    idx = rng.integers(0, 5000, size=777)
    out = gather_rows(torch.from_numpy(rows).to(DEV), torch.from_numpy(idx).to(DEV))
    assert np.array_equal(out.cpu().numpy(), rows[idx])
