// Grouped GEMM with bf16 OPERANDS and fp32 accumulation on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16) — the
// opt-in mode of the costmap encoder (BASELINE config 5 asks for bf16; the reference itself is fp32 everywhere, so fp32
// stays the default and the parity path).
//
// Tensors stay fp32 in HBM: both operands are read as fp32 (16-byte buffer loads), rounded to bf16 (round to nearest
// even, v_cvt_pk_bf16_f32) while they are parked in LDS, multiplied on the bf16 matrix pipe (16x the fp32-input rate)
// and accumulated in fp32; the epilogue is the fp32 kernel's (bias / activation, DropPath-scaled residual, BatchNorm
// column statistics, 16-byte C rows through LDS).  With the arithmetic out of the way the products of the encoder —
// millions of rows times K = 96 ... 384 — are bound by HBM: rows x (K + N) x 4 bytes.
//
// Only what the encoder needs: forward layout (A (M, K) and B (N, K), both k-contiguous), 16-byte aligned operands,
// K % 4 == 0, no split-K.  The optional A-operand prologue is the fp32 kernel's APRO_AFFINE_RELU (BatchNorm + ReLU of
// the producer), applied in fp32 BEFORE the rounding to bf16.
#pragma once
#include "gemm_f32.hpp"

namespace porl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// BM x BN block tile, 2 x 2 waves, each wave (BM/2) x (BN/2) = WTM x WTN MFMA tiles of 32 x 32.  K-tile = 32 columns.
// LDS image per operand and buffer: [rows][32 + 8] bf16 (80-byte rows: 16-byte aligned fragments, no bank aliasing
// between the rows a half-wave reads).
template <int BM, int BN, bool APRO>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmGroup g) {
  constexpr int BK = 32, WM = 2, WN = 2, THREADS = 256;      // (a 64-column K-tile measured 4 % slower: 52.8 vs 54.9 updates/s)
  constexpr int WTM = BM / WM / 32, WTN = BN / WN / 32;
  constexpr int SKB = BK + 8;                           // LDS row stride in bf16 elements
  constexpr int A_TILE = BM * SKB, B_TILE = BN * SKB;   // bf16 elements
  constexpr int NLA = BM * BK / 4 / THREADS, NLB = BN * BK / 4 / THREADS;
  constexpr int APRO_MAX_K = 1024;
  // the C sub-tiles of the epilogue need WM*WN*(BM/WM)*((BN/WN)+4) floats; operands need 2*(A_TILE+B_TILE) bf16
  constexpr int WROWS = BM / WM, WCOLS = BN / WN, CS = WCOLS + 4;
  constexpr int LDS_OPER_FLOATS = (2 * (A_TILE + B_TILE) + 1) / 2;
  constexpr int LDS_C_FLOATS = WM * WN * WROWS * CS;
  constexpr int LDS_MAIN = LDS_OPER_FLOATS > LDS_C_FLOATS ? LDS_OPER_FLOATS : LDS_C_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN + (APRO ? 2 * APRO_MAX_K : 0)];
  __bf16* const ldsb = reinterpret_cast<__bf16*>(lds);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, kh = lane >> 5;

  int lin;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_GROUP; ++i)
    if (i < g.nprob && lin >= g.p[i].block_start) pi = i;
  const GemmProb& P = g.p[pi];
  const int local = lin - P.block_start;
  const int tm = local / P.tiles_n, tn = local - tm * P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = P.M, N = P.N, K = P.K;
  const int nkt = (K + BK - 1) / BK;

  const BufRsrc a_rsrc = make_rsrc(P.A, (size_t)M * P.lda * sizeof(float));
  const BufRsrc b_rsrc = make_rsrc(P.B, (size_t)N * P.ldb * sizeof(float));
  // slot f = t + 256 i: row f / (BK/4), columns (f % (BK/4)) * 4 .. +3 of the K-tile
  unsigned a_off[NLA], b_off[NLB];
  int a_lds[NLA], b_lds[NLB], a_k[NLA], b_k[NLB];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int f = t + THREADS * i, kq = f % (BK / 4), row = f / (BK / 4);
    a_k[i] = kq * 4;
    a_lds[i] = row * SKB + kq * 4;
    a_off[i] = (m0 + row < M) ? (unsigned)(((size_t)(m0 + row) * P.lda + kq * 4) * sizeof(float)) : BUF_OOB;
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int f = t + THREADS * i, kq = f % (BK / 4), row = f / (BK / 4);
    b_k[i] = kq * 4;
    b_lds[i] = row * SKB + kq * 4;
    b_off[i] = (n0 + row < N) ? (unsigned)(((size_t)(n0 + row) * P.ldb + kq * 4) * sizeof(float)) : BUF_OOB;
  }
  float* const apro_cs = lds + LDS_MAIN;
  if constexpr (APRO) {
    for (int k = t; k < K; k += THREADS) { apro_cs[k] = P.a_colscale[k]; apro_cs[APRO_MAX_K + k] = P.a_colshift[k]; }
  }

  float4 ra[NLA], rb[NLB];
  auto load_tile = [&](int kt) {
    const unsigned soff = (unsigned)(kt * BK * sizeof(float));
    // columns at or past K read 0.0 through the range check of an out-of-range offset (K % 4 == 0)
#pragma unroll
    for (int i = 0; i < NLA; ++i) ra[i] = buf_ld128(a_rsrc, (kt * BK + a_k[i] < K) ? a_off[i] : BUF_OOB, soff);
#pragma unroll
    for (int i = 0; i < NLB; ++i) rb[i] = buf_ld128(b_rsrc, (kt * BK + b_k[i] < K) ? b_off[i] : BUF_OOB, soff);
  };
  auto to_bf16 = [](const float4& v) {
    bf16x4 o;
    o.x = (__bf16)v.x; o.y = (__bf16)v.y; o.z = (__bf16)v.z; o.w = (__bf16)v.w;
    return o;
  };
  auto store_tile = [&](int kt, int buf) {
    __bf16* As = ldsb + buf * (A_TILE + B_TILE);
    __bf16* Bs = As + A_TILE;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      float4 v = ra[i];
      if constexpr (APRO) {
        const int k = kt * BK + a_k[i];
        if (k < K) {
          const float4 cs = *reinterpret_cast<const float4*>(apro_cs + k);
          const float4 cb = *reinterpret_cast<const float4*>(apro_cs + APRO_MAX_K + k);
          v.x = fmaxf(fmaf(v.x, cs.x, cb.x), 0.f); v.y = fmaxf(fmaf(v.y, cs.y, cb.y), 0.f);
          v.z = fmaxf(fmaf(v.z, cs.z, cb.z), 0.f); v.w = fmaxf(fmaf(v.w, cs.w, cb.w), 0.f);
        }
      }
      *reinterpret_cast<bf16x4*>(As + a_lds[i]) = to_bf16(v);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) *reinterpret_cast<bf16x4*>(Bs + b_lds[i]) = to_bf16(rb[i]);
  };

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nkt > 0) load_tile(0);
  if constexpr (APRO) __syncthreads();
  if (nkt > 0) store_tile(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);                 // in flight while this tile is multiplied
    const __bf16* As = ldsb + buf * (A_TILE + B_TILE);
    const __bf16* Bs = As + A_TILE;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fa[WTM], fb[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(As + (wm * (BM / WM) + i * 32 + li) * SKB + kk * 16 + kh * 8);
#pragma unroll
      for (int j = 0; j < WTN; ++j)
        fb[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn * (BN / WN) + j * 32 + li) * SKB + kk * 16 + kh * 8);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(kt + 1, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: the fp32 kernel's (gemm_f32.hpp), same accumulator layout ---------------------------------------
  float* __restrict__ Cg = P.C;
  const int act = P.act;
  const bool store_c = P.store_c != 0;
  const bool fast_c = store_c && P.c_vec && (m0 + BM <= M) && (n0 + BN <= N);
  float* ctile = lds + wave * (WROWS * CS);
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    const int rbase = m0 + wm * (BM / WM) + i * 32 + 4 * kh;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 32 + li;
      const bool col_ok = col < N;
      const float bv = (col_ok && P.bias) ? P.bias[col] : 0.f;
      float vals[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv;
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ACT_TANH) v = tanhf(v);
        vals[r] = v;
      }
      if (P.resid && !fast_c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < M && col_ok) {
            const float rs = P.rscale ? P.rscale[(row + P.rs_row0) / P.rs_rows] : 1.f;
            vals[r] = __fadd_rn(P.resid[(size_t)row * P.ldc + col], __fmul_rn(vals[r], rs));
          }
        }
      }
      if (P.cstat) {
        float cs = 0.f, cq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          const float v = row < M ? vals[r] : 0.f;
          cs += v;
          cq = fmaf(v, v, cq);
        }
        cs += __shfl_xor(cs, 32);
        cq += __shfl_xor(cq, 32);
        if (kh == 0 && col_ok) {
          float* o = P.cstat + (size_t)((m0 + wm * (BM / WM)) / 32 + i) * 2 * N + col;
          o[0] = cs;
          o[N] = cq;
        }
      }
      if (store_c) {
        if (fast_c) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ctile[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * CS + j * 32 + li] = vals[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (row < M && col_ok) Cg[(size_t)row * P.ldc + col] = vals[r];
          }
        }
      }
    }
  }
  if (fast_c) {
    constexpr int C4 = WCOLS / 4;
    const int wr0 = m0 + wm * WROWS, wc0 = n0 + wn * WCOLS;
#pragma unroll 4
    for (int f = lane; f < WROWS * C4; f += 64) {
      const int r = f / C4, c = (f % C4) * 4;
      float4 v = *reinterpret_cast<const float4*>(ctile + r * CS + c);
      if (P.resid) {
        const float rs = P.rscale ? P.rscale[(wr0 + r + P.rs_row0) / P.rs_rows] : 1.f;
        const float4 x4 = *reinterpret_cast<const float4*>(P.resid + (size_t)(wr0 + r) * P.ldc + wc0 + c);
        v.x = __fadd_rn(x4.x, __fmul_rn(v.x, rs)); v.y = __fadd_rn(x4.y, __fmul_rn(v.y, rs));
        v.z = __fadd_rn(x4.z, __fmul_rn(v.z, rs)); v.w = __fadd_rn(x4.w, __fmul_rn(v.w, rs));
      }
      *reinterpret_cast<float4*>(Cg + (size_t)(wr0 + r) * P.ldc + wc0 + c) = v;
    }
  }
}

// tile: TILE_128x64 or TILE_64x64.  Every problem: forward layout, 16-byte operands, no split-K, dense A.
inline hipError_t launch_gemm_bf16_group(int tile, GemmGroup& g, hipStream_t s) {
  if (g.nprob < 1 || g.nprob > MAX_GROUP) return hipErrorInvalidValue;
  bool apro = g.p[0].apro != APRO_NONE;
  for (int i = 0; i < g.nprob; ++i) {
    const GemmProb& p = g.p[i];
    if (!p.a_kc || !p.b_kc || !p.a_vec || !p.b_vec || p.splitk > 1 || p.a_grp > 0 || p.mask || p.headw || p.colsum)
      return hipErrorInvalidValue;
    if ((p.apro != APRO_NONE) != apro) return hipErrorInvalidValue;
    if (apro && (p.K > 1024 || p.K % 4 || !p.a_colscale || !p.a_colshift)) return hipErrorInvalidValue;
  }
  if (plan_group(g, tile) == 0) return hipSuccess;
  dim3 grid(g.total_blocks), block(256);
  if (tile == TILE_128x64) {
    if (apro) hipLaunchKernelGGL((gemm_bf16_kernel<128, 64, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_bf16_kernel<128, 64, false>), grid, block, 0, s, g);
  } else if (tile == TILE_64x64) {
    if (apro) hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, false>), grid, block, 0, s, g);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace porl
