// Input layer of the update's MLPs on its own kernel:  out = relu(X W^T + b) with K = obs_dim <= 64.
//
// The grouped GEMM (gemm_f32.hpp) is built around a K loop that hides its set-up; at K = 60 a block has two K-tiles and
// the launch is all prologue and epilogue: 14.9 us for the four value nets at B = H = 1024 (0.5 GFLOP, 16 MB written).
// Here a block takes a 64 x 128 output tile of one net, fetches its whole operand set (64 x K of X, 128 x K of W) in
// ONE round of 16-byte loads, multiplies from LDS and leaves through the same transposing 16-byte epilogue.
//
// Arithmetic is the GEMM's, bit for bit: v_mfma_f32_32x32x2_f32 over two K-tiles of 32 in the GEMM's k order (group g,
// step j: lanes 0-31 take k = 32 kt + 8 g + j, lanes 32-63 k + 4), columns >= K are zeros in both operands, then
// bias and ReLU — so results do not depend on which path ran (porl_tune_set("l0_kernel", 0) keeps the GEMM: tested).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_f32.hpp"

namespace porl {

constexpr int L0_MAX_NETS = 8;
constexpr int L0_BM = 64, L0_BN = 128, L0_KP = 64, L0_SK = L0_KP + 4;

struct L0Net { const float* X; const float* W; const float* b; float* out; int ldx; };
struct L0Args {
  int nnets, B, H, K, ldw, ldo, tiles_m, tiles_n;
  L0Net net[L0_MAX_NETS];
};

__global__ __launch_bounds__(256) void l0_fwd_kernel(const L0Args a) {
  __shared__ __attribute__((aligned(16))) float lds[(L0_BM + L0_BN) * L0_SK];
  float* As = lds;
  float* Bs = lds + L0_BM * L0_SK;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;            // 2 x 2 waves: 32 rows x 64 columns each
  const int per_net = a.tiles_m * a.tiles_n;
  const int ni = blockIdx.x / per_net, rem = blockIdx.x - ni * per_net;
  const int tn = rem / a.tiles_m, tm = rem - tn * a.tiles_m;
  const L0Net& P = a.net[ni];
  const int m0 = tm * L0_BM, n0 = tn * L0_BN;
  const int K4 = a.K >> 2;                             // valid float4 per row (K is a multiple of 4: host check)

  // ---- the whole operand set in one round of loads -------------------------------------------------------------
  float4 xa[4], wb[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = t + 256 * i, row = f >> 4, k4 = f & 15;
    const bool ok = (m0 + row < a.B) && (k4 < K4);
    const float4 v = *reinterpret_cast<const float4*>(P.X + (size_t)(ok ? m0 + row : 0) * P.ldx + (ok ? k4 * 4 : 0));
    xa[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = t + 256 * i, row = f >> 4, k4 = f & 15;
    const bool ok = (n0 + row < a.H) && (k4 < K4);
    const float4 v = *reinterpret_cast<const float4*>(P.W + (size_t)(ok ? n0 + row : 0) * a.ldw + (ok ? k4 * 4 : 0));
    wb[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = t + 256 * i, row = f >> 4, k4 = f & 15;
    *reinterpret_cast<float4*>(As + row * L0_SK + k4 * 4) = xa[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = t + 256 * i, row = f >> 4, k4 = f & 15;
    *reinterpret_cast<float4*>(Bs + row * L0_SK + k4 * 4) = wb[i];
  }
  __syncthreads();

  // ---- 64 k-steps on the matrix pipe ------------------------------------------------------------------------------
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float* arow = As + (wm * 32 + li) * L0_SK + kh * 4;
  const float* brow0 = Bs + (wn * 64 + li) * L0_SK + kh * 4;
  const float* brow1 = brow0 + 32 * L0_SK;
#pragma unroll
  for (int g = 0; g < L0_KP / 8; ++g) {
    const float4 fa = *reinterpret_cast<const float4*>(arow + g * 8);
    const float4 f0 = *reinterpret_cast<const float4*>(brow0 + g * 8);
    const float4 f1 = *reinterpret_cast<const float4*>(brow1 + g * 8);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, f0.x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, f1.x, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, f0.y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, f1.y, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, f0.z, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, f1.z, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, f0.w, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, f1.w, acc[1], 0, 0, 0);
  }
  __syncthreads();                                     // every wave is done with the operand images

  // ---- bias + ReLU, transposed through LDS, out as 16-byte rows -----------------------------------------------------
  constexpr int CS = 64 + 4;
  float* ctile = lds + wave * (32 * CS);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + li;
    const float bv = col < a.H ? P.b[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      ctile[((r & 3) + 8 * (r >> 2) + 4 * kh) * CS + j * 32 + li] = fmaxf(acc[j][r] + bv, 0.f);
  }
  // (a wave reads back only what it wrote: no block barrier needed, the compiler orders the LDS accesses of a wave)
  const int wr0 = m0 + wm * 32, wc0 = n0 + wn * 64;
#pragma unroll
  for (int f = lane; f < 32 * 16; f += 64) {
    const int r = f >> 4, c = (f & 15) * 4;
    if (wr0 + r < a.B && wc0 + c < a.H) {              // H is a multiple of 4 (host check): a float4 is in or out
      const float4 v = *reinterpret_cast<const float4*>(ctile + r * CS + c);
      *reinterpret_cast<float4*>(P.out + (size_t)(wr0 + r) * a.ldo + wc0 + c) = v;
    }
  }
}

// true when the kernel can take the layer (the caller falls back to the grouped GEMM otherwise)
inline bool l0_fwd_supported(const L0Args& a) {
  if (a.nnets < 1 || a.nnets > L0_MAX_NETS || a.K < 4 || a.K > L0_KP || a.K % 4 || a.H % 4 || a.ldw % 4 || a.ldo % 4)
    return false;
  for (int i = 0; i < a.nnets; ++i) {
    const L0Net& n = a.net[i];
    if (!n.X || !n.W || !n.b || !n.out || n.ldx % 4 || !aligned16(n.X) || !aligned16(n.W) || !aligned16(n.out)) return false;
  }
  return true;
}

inline hipError_t launch_l0_fwd(L0Args& a, hipStream_t s) {
  a.tiles_m = (a.B + L0_BM - 1) / L0_BM;
  a.tiles_n = (a.H + L0_BN - 1) / L0_BN;
  hipLaunchKernelGGL(l0_fwd_kernel, dim3(a.tiles_m * a.tiles_n * a.nnets), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace porl
