// Products with one dimension <= 64 on kernels of their own (gfx950, v_mfma_f32_32x32x2_f32, exact fp32).
//
// The POR / SORL update has four of them per step, all a few hundred MFLOP moving a few MB, and the grouped GEMM
// (gemm_f32.hpp: a K loop built to hide its own set-up) spent 7-15 us on each (profiles/r02_step_table.txt):
//   * input-layer weight gradients   dW0 (H, S)  = dZ0^T X         S = obs_dim <= 64    (value twins, policy)
//   * policy mean                    mu  (B, D)  = A W_L^T         D = pol_out_dim <= 64
//   * policy output-layer backward   dZ  (B, H)  = (dmu W_L) . 1[A > 0]   and   dW_L (D, H) = dmu^T A
// Here a block owns one 64 x 64 tile of the BIG matrix (dZ0 / A), fetches it and the matching 64 x 64 piece of the
// skinny operand(s) in ONE round of 16-byte loads, multiplies from LDS and stores.  Contractions over the batch (the
// weight gradients) and over H (the mean) leave per-tile partial results as SLABS in the layout the split-K GEMM
// used, so the consumers are unchanged: the Adam launch sums weight-gradient slabs in its sweep (kernels.hpp:
// adam_ema_kernel), the NLL kernel sums the mean's.  Summation order inside a 64-deep chunk is the natural one here
// (the GEMM walks k in its 8-group order), so results agree with the GEMM path to rounding, not bit for bit:
// porl_tune_set("skinny", 0) keeps the GEMM path for the comparison test.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_f32.hpp"

namespace porl {

constexpr int SKN_T = 64;            // tile edge: rows, columns, and the padded skinny dimension
constexpr int SKN_S = SKN_T + 4;     // LDS row stride (floats): 16-byte rows, conflict-free dword columns
constexpr int SKN_MAX_NETS = 4;

// One 64 x 64 tile of a row-major matrix as 4 float4 per thread (256 threads): rows r0.., columns c0..; elements
// outside [0, R) x [0, C) read as zero.  C and ld are multiples of 4 (host check), so a float4 is in or out as a whole.
// Out-of-range lanes load from the matrix base and select zero: no branch around a load.
__device__ __forceinline__ void skn_load(float4 (&v)[4], const float* __restrict__ M, long ld, int r0, int c0, int R,
                                         int C, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = t + 256 * i, row = f >> 4, c4 = (f & 15) * 4;
    const bool ok = (r0 + row < R) && (c0 + c4 < C);
    const float4 x = *reinterpret_cast<const float4*>(M + (ok ? (long)(r0 + row) * ld + c0 + c4 : 0));
    v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void skn_park(const float4 (&v)[4], float* __restrict__ L, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = t + 256 * i, row = f >> 4, c4 = (f & 15) * 4;
    *reinterpret_cast<float4*>(L + row * SKN_S + c4) = v[i];
  }
}
// acc += P^T Q over the 64 rows of two parked tiles: out[m][n] = sum_k P[k][32 pm + m] * Q[k][32 qn + n]
// (both fragments are dword reads along a row: lanes take consecutive columns)
__device__ __forceinline__ void skn_mma_tn(f32x16& acc, const float* __restrict__ P, const float* __restrict__ Q, int pm,
                                           int qn, int li, int kh) {
  const float* p = P + kh * SKN_S + 32 * pm + li;
  const float* q = Q + kh * SKN_S + 32 * qn + li;
#pragma unroll
  for (int s = 0; s < SKN_T / 2; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p[2 * s * SKN_S], q[2 * s * SKN_S], acc, 0, 0, 0);
}
// row of accumulator register r for lane half kh (v_mfma_f32_32x32x2_f32 output layout); the column is lane & 31
__device__ __forceinline__ int skn_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// ---------------------------------------------------------------------------------------------------------------------
// dW (H, S) = Z^T X and db (H) = column sums of Z, as slabs over batch chunks.  Z (B, H) is the pre-activation gradient
// of an input layer, X (B, S) the minibatch.  grid = nets x slabs x column tiles; a block walks `rtiles` 64-row tiles
// of its batch chunk and keeps its 64 (n) x 64 (s) result in the accumulators.
// ---------------------------------------------------------------------------------------------------------------------
struct WgradSkinnyNet { const float* Z; const float* X; float* slabW; float* slabC; };
struct WgradSkinnyArgs {
  int nnets, B, H, S;                 // S <= 64 valid columns of X
  int ldz, ldx, ldo;                  // ldo: row length of dW (= S for an (H, S) weight)
  int tiles_n, nslab, rtiles;
  long slabW_stride, slabC_stride;    // floats between slabs
  WgradSkinnyNet net[SKN_MAX_NETS];
};

__global__ __launch_bounds__(256) void wgrad_skinny_kernel(const WgradSkinnyArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * SKN_T * SKN_S];
  float* Zl = lds;
  float* Xl = lds + SKN_T * SKN_S;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;                     // n-tile, s-tile of this wave
  const int per_net = a.nslab * a.tiles_n;
  const int ni = blockIdx.x / per_net, rem = blockIdx.x - ni * per_net;
  const int ts = rem / a.tiles_n, tn = rem - ts * a.tiles_n;
  const WgradSkinnyNet& P = a.net[ni];
  const int n0 = tn * SKN_T;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float cs = 0.f;
  for (int rt = 0; rt < a.rtiles; ++rt) {
    const int b0 = (ts * a.rtiles + rt) * SKN_T;
    if (b0 >= a.B) break;                                      // block-uniform
    float4 vz[4], vx[4];
    skn_load(vz, P.Z, a.ldz, b0, n0, a.B, a.H, t);
    skn_load(vx, P.X, a.ldx, b0, 0, a.B, a.ldx, t);            // padding columns of X are zeros (pack / sampler)
    if (rt) __syncthreads();                                   // the previous tiles have been consumed
    skn_park(vz, Zl, t);
    skn_park(vx, Xl, t);
    __syncthreads();
    skn_mma_tn(acc, Zl, Xl, wm, wn, li, kh);
    if (t < SKN_T) {
#pragma unroll 8
      for (int b = 0; b < SKN_T; ++b) cs += Zl[b * SKN_S + t];
    }
  }
  float* __restrict__ oW = P.slabW + (long)ts * a.slabW_stride;
  const int s = 32 * wn + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + 32 * wm + skn_row(r, kh);
    if (n < a.H && s < a.S) oW[(long)n * a.ldo + s] = acc[r];
  }
  if (t < SKN_T && n0 + t < a.H) P.slabC[(long)ts * a.slabC_stride + n0 + t] = cs;
}

// ---------------------------------------------------------------------------------------------------------------------
// mu (B, D) = A W^T as slabs over chunks of the H (contraction) dimension: A (B, H) last hidden activation, W (D, H)
// output-layer weight, both k-contiguous like a forward product.  grid = row tiles x slabs; a block walks `ktiles`
// 64-wide K-tiles.  Padding columns D..ldo-1 of the slab rows are written as zeros.
// ---------------------------------------------------------------------------------------------------------------------
struct MeanSkinnyArgs {
  const float* A; const float* W; float* slab;
  int B, H, D, lda, ldw, ldo;
  int tiles_m, nslab, ktiles;
  long slab_stride;
};

__global__ __launch_bounds__(256) void mean_skinny_kernel(const MeanSkinnyArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * SKN_T * SKN_S];
  float* Al = lds;
  float* Wl = lds + SKN_T * SKN_S;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;                     // b-tile, d-tile
  const int tm = blockIdx.x % a.tiles_m, ts = blockIdx.x / a.tiles_m;
  const int b0 = tm * SKN_T;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kt = 0; kt < a.ktiles; ++kt) {
    const int k0 = (ts * a.ktiles + kt) * SKN_T;
    if (k0 >= a.H) break;
    float4 va[4], vw[4];
    skn_load(va, a.A, a.lda, b0, k0, a.B, a.H, t);
    skn_load(vw, a.W, a.ldw, 0, k0, a.D, a.H, t);
    if (kt) __syncthreads();
    skn_park(va, Al, t);
    skn_park(vw, Wl, t);
    __syncthreads();
    // both images are k-contiguous: a lane reads 4 consecutive k with one 16-byte load; lane half kh takes
    // k = 8 g + 4 kh + j in step (g, j) for both operands
    const float* arow = Al + (32 * wm + li) * SKN_S + kh * 4;
    const float* wrow = Wl + (32 * wn + li) * SKN_S + kh * 4;
#pragma unroll
    for (int g = 0; g < SKN_T / 8; ++g) {
      const float4 fa = *reinterpret_cast<const float4*>(arow + g * 8);
      const float4 fw = *reinterpret_cast<const float4*>(wrow + g * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fw.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fw.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fw.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fw.w, acc, 0, 0, 0);
    }
  }
  float* __restrict__ o = a.slab + (long)ts * a.slab_stride;
  const int d = 32 * wn + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int b = b0 + 32 * wm + skn_row(r, kh);
    if (b < a.B && d < a.ldo) o[(long)b * a.ldo + d] = acc[r];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Output-layer backward of the policy in one pass over the last hidden activation A (B, H):
//   dZ (B, H)  = (dmu W) . 1[A > 0]          dmu (B, D) gradient w.r.t. the pre-activation mean, W (D, H)
//   dW (D, H)  = dmu^T A,   db (D) = column sums of dmu        as slabs over batch chunks
// grid = slabs x column tiles; a block keeps its W tile and its 64 (d) x 64 (n) dW result while it walks `rtiles`
// 64-row tiles, writing the dZ tile of each.
// ---------------------------------------------------------------------------------------------------------------------
struct OutBwdArgs {
  const float* dmu; const float* W; const float* A;
  float* dZ; float* slabW; float* slabC;
  int B, H, D;
  int lddmu, ldw, lda, lddz;
  int tiles_n, nslab, rtiles;
  long slabW_stride, slabC_stride;
};

__global__ __launch_bounds__(256) void out_bwd_kernel(const OutBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[3 * SKN_T * SKN_S];
  float* Wl = lds;                           // [d][n]
  float* Dl = lds + SKN_T * SKN_S;           // [b][d]
  float* Al = lds + 2 * SKN_T * SKN_S;       // [b][n]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ts = blockIdx.x / a.tiles_n, tn = blockIdx.x - ts * a.tiles_n;
  const int n0 = tn * SKN_T;
  {
    float4 vw[4];
    skn_load(vw, a.W, a.ldw, 0, n0, a.D, a.H, t);
    skn_park(vw, Wl, t);
  }
  f32x16 accw;
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[r] = 0.f;
  float cs = 0.f;
  const bool do_cs = tn == 0;
  for (int rt = 0; rt < a.rtiles; ++rt) {
    const int b0 = (ts * a.rtiles + rt) * SKN_T;
    if (b0 >= a.B) break;
    float4 vd[4], va[4];
    skn_load(vd, a.dmu, a.lddmu, b0, 0, a.B, a.lddmu, t);      // padding columns of dmu are zeros (policy_nll_kernel)
    skn_load(va, a.A, a.lda, b0, n0, a.B, a.H, t);
    if (rt) __syncthreads();
    skn_park(vd, Dl, t);
    skn_park(va, Al, t);
    __syncthreads();                                           // (also makes the W tile visible in the first trip)
    // dZ tile: rows 32 wm.., columns 32 wn..; A operand (m = b, k = d) is k-contiguous: 16-byte reads, lane half kh
    // takes d = 8 g + 4 kh + j; the B operand W[d][n] follows the same d
    f32x16 accz;
#pragma unroll
    for (int r = 0; r < 16; ++r) accz[r] = 0.f;
    const float* drow = Dl + (32 * wm + li) * SKN_S + kh * 4;
    const float* wcol = Wl + (kh * 4) * SKN_S + 32 * wn + li;
#pragma unroll
    for (int g = 0; g < SKN_T / 8; ++g) {
      const float4 fd = *reinterpret_cast<const float4*>(drow + g * 8);
      accz = __builtin_amdgcn_mfma_f32_32x32x2f32(fd.x, wcol[(g * 8 + 0) * SKN_S], accz, 0, 0, 0);
      accz = __builtin_amdgcn_mfma_f32_32x32x2f32(fd.y, wcol[(g * 8 + 1) * SKN_S], accz, 0, 0, 0);
      accz = __builtin_amdgcn_mfma_f32_32x32x2f32(fd.z, wcol[(g * 8 + 2) * SKN_S], accz, 0, 0, 0);
      accz = __builtin_amdgcn_mfma_f32_32x32x2f32(fd.w, wcol[(g * 8 + 3) * SKN_S], accz, 0, 0, 0);
    }
    // dW tile (d-tile wm, n-tile wn): contraction over the 64 rows
    skn_mma_tn(accw, Dl, Al, wm, wn, li, kh);
    if (do_cs && t < SKN_T) {
#pragma unroll 8
      for (int b = 0; b < SKN_T; ++b) cs += Dl[b * SKN_S + t];
    }
    // ReLU mask from the parked activation tile, then out: a half-wave writes 32 consecutive floats of one row
    const int n = 32 * wn + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bl = 32 * wm + skn_row(r, kh);
      const float v = Al[bl * SKN_S + n] > 0.f ? accz[r] : 0.f;
      if (b0 + bl < a.B && n0 + n < a.H) a.dZ[(long)(b0 + bl) * a.lddz + n0 + n] = v;
    }
  }
  float* __restrict__ oW = a.slabW + (long)ts * a.slabW_stride;
  const int n = n0 + 32 * wn + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = 32 * wm + skn_row(r, kh);
    if (d < a.D && n < a.H) oW[(long)d * a.H + n] = accw[r];
  }
  if (do_cs && t < a.D) a.slabC[(long)ts * a.slabC_stride + t] = cs;
}

// ---- host side --------------------------------------------------------------------------------------------------------
// slabs / tiles-per-block for a contraction of `tiles` 64-wide tiles with at most `max_slabs` partial results
inline void skn_split(int tiles, int max_slabs, int& nslab, int& per_block) {
  per_block = (tiles + max_slabs - 1) / max_slabs;
  if (per_block < 1) per_block = 1;
  nslab = (tiles + per_block - 1) / per_block;
}

inline bool skn_ok4(const void* p, long ld) { return p && aligned16(p) && ld % 4 == 0; }

}  // namespace porl
