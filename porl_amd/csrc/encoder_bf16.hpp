// The costmap encoder's bf16 mode (compute_dtype="bf16", BASELINE config 5's wording) with bf16 ACTIVATIONS in HBM.
//
// Round 2's bf16 mode kept every tensor fp32 in memory and only rounded operands on their way into LDS: its products
// ran at 3.4 TB/s, i.e. bound by fp32 traffic, and "bf16" bought 1.4x.  Here the activations between the convolutions
// live in HBM as bf16 and the encoder's MLPBlock (agent/fasternet.py:141-190)
//         x <- x + drop_scale * W2 relu(BN(W1 [PConv3x3(x[:, :C/4]) | x[:, C/4:]]))
// is ONE kernel per pass over x instead of four launches with a (rows, 2C) hidden tensor in between:
//   pass 1 (train mode only)  h = A W1^T per 128-row tile on the bf16 matrix pipe, only its per-32-row column sums and
//                             sums of squares leave the chip (the BatchNorm statistics need the whole batch first);
//   pass 2                    h again (the matrix work is ~free at bf16 rates), BatchNorm scale/shift + ReLU on the fp32
//                             accumulators, h as bf16 through LDS into the second product, DropPath-scaled residual,
//                             x written back in place.
// Traffic per block and forward: ~3.25 x rows x C x 2 B instead of 7 x rows x C x 4 B.  The partial 3x3 convolution is
// an implicit GEMM on the bf16 matrix pipe (9 taps x C/4 input channels), the 2x2s2 PatchMerging a gathered-operand
// product with its weight walked tap by tap through LDS.  BatchNorm statistics, scale/shift, the patch embedding's
// arithmetic, the pooled head and everything after it stay fp32; accumulation is fp32 everywhere.
// fp32 (360x256) remains the parity path: this mode has no reference counterpart and its tolerance is this build's
// (tests/test_fasternet_gpu.py, tests/test_config5_gpu.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16.hpp"

namespace porl {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
// 16 bytes = 8 bf16 from global memory, or zeros (no branch around the load)
__device__ __forceinline__ u32x4 ld16_or_zero(const __bf16* base, long off, bool ok) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(base + (ok ? off : 0));
  return ok ? v : u32x4{0u, 0u, 0u, 0u};
}

// fp32 -> bf16 (round to nearest even), linear
__global__ void pack_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = (__bf16)src[i];
}
// Partial_conv3 weight (oc, ci, 3, 3) fp32 -> [tap = ky*3+kx][oc (CPN, zero padded)][ci (XS: CPP + 8, zero padded)] bf16
__global__ void pack_pconv_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int CP, int CPN, int XS) {
  const int n = 9 * CPN * XS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int tap = i / (CPN * XS), rem = i - tap * CPN * XS, oc = rem / XS, ci = rem - oc * XS;
    dst[i] = (oc < CP && ci < CP) ? (__bf16)src[((long)oc * CP + ci) * 9 + tap] : (__bf16)0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Partial_conv3 (fasternet.py:110-138) as an implicit GEMM: out (POS positions) x (CP output channels), K = 9 taps x CP
// input channels, v_mfma_f32_32x32x16_bf16.  x (B, H, W, C) bf16 NHWC, the convolution reads channels [0, CP); yc is the
// dense (B*H*W, CP) result (the untouched channels are NOT copied: the MLP kernel reads them from x itself).
// A block takes POS consecutive positions of one sample; the input rows they touch (+ halo, zero outside the image) and
// the whole weight sit in LDS.  Pixel stride XS = CPP + 8 bf16 (CPP = CP rounded up to 16): 16-byte fragments, rows of a
// half-wave on different banks.
// ---------------------------------------------------------------------------------------------------------------------
struct PconvBf16Args {
  const __bf16* x; __bf16* yc; const __bf16* w;   // w: pack_pconv_bf16_kernel's image
  int ldx, Hh, Ww, tiles_per_sample, nr_max;
  int tiles_total, tiles_per_block;              // a block walks tiles_per_block consecutive tiles with the weight parked once
};

template <int CP, int CPP, int CPN, int POS>
__global__ __launch_bounds__(256) void pconv_bf16_kernel(const PconvBf16Args a) {
  constexpr int XS = CPP + 8, MT = POS / 32, NT = CPN / 32, U = CPP / 8;
  static_assert(MT * NT == 4, "four waves, one 32 x 32 tile each");
  extern __shared__ __attribute__((aligned(16))) unsigned char pcb_lds[];
  __bf16* Wl = reinterpret_cast<__bf16*>(pcb_lds);                 // [9][CPN][XS]
  __bf16* X = Wl + 9 * CPN * XS;                                  // [nr][Ww + 2][XS]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kh = lane >> 5;
  const int P = a.Hh * a.Ww, WW2 = a.Ww + 2;
  const int mt = wave % MT, nt = wave / MT;
  // weights: linear 16-byte copy, once per block
  for (int i = t; i < 9 * CPN * XS / 8; i += 256)
    reinterpret_cast<u32x4*>(Wl)[i] = reinterpret_cast<const u32x4*>(a.w)[i];
  for (int k = 0; k < a.tiles_per_block; ++k) {
    const int gt = blockIdx.x * a.tiles_per_block + k;
    if (gt >= a.tiles_total) break;                                // block-uniform
    const int b = gt / a.tiles_per_sample, tile = gt - b * a.tiles_per_sample;
    const int p0 = tile * POS, p1 = min(P, p0 + POS) - 1;
    const int y_lo = p0 / a.Ww - 1, nr = p1 / a.Ww + 1 - y_lo + 1;
    if (k) __syncthreads();                                        // the previous tile's fragments have been read
    // input rows with halo; units of 8 channels
    for (int i = t; i < nr * WW2 * U; i += 256) {
      const int pix = i / U, u = i - pix * U, ry = pix / WW2, rx = pix - ry * WW2;
      const int yy = y_lo + ry, xx = rx - 1;
      const bool ok = yy >= 0 && yy < a.Hh && xx >= 0 && xx < a.Ww && u * 8 < CP;
      const u32x4 v = ld16_or_zero(a.x, (((long)b * a.Hh + yy) * a.Ww + xx) * a.ldx + u * 8, ok);
      *reinterpret_cast<u32x4*>(X + pix * XS + u * 8) = v;
    }
    __syncthreads();
    const int p = p0 + 32 * mt + li;
    const int pc = p < P ? p : p0;                                 // lanes past the sample compute a valid pixel, unused
    const int yy = pc / a.Ww, xx = pc - yy * a.Ww;
    const __bf16* xa = X + ((yy - y_lo) * WW2 + xx + 1) * XS + kh * 8;
    const __bf16* wb = Wl + (32 * nt + li) * XS + kh * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int off = ((tap / 3 - 1) * WW2 + (tap % 3 - 1)) * XS;
#pragma unroll
      for (int ks = 0; ks < CPP / 16; ++ks) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(xa + off + ks * 16);
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(wb + tap * CPN * XS + ks * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
      }
    }
    const int oc = 32 * nt + li;
    if (oc < CP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pos = p0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (pos < P) a.yc[((long)b * P + pos) * CP + oc] = (__bf16)acc[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// MLPBlock body on 128-row tiles (see the header).  A = [yc | x[:, CP:]] (rows, DIM) bf16; W1 (HID, DIM), W2 (DIM, HID)
// bf16 (converted once per weight change); hidden columns are walked in chunks of 64: W1's chunk (64 rows) and W2's
// chunk (64 columns of every row) are requested into registers one chunk ahead and parked in LDS between two
// barriers.  Wave w owns rows 32 w .. 32 w + 31 of the tile for BOTH products, so the hidden chunk goes through LDS
// wave-locally (written and read by the same wave: no block barrier).
//   PASS 1: cstat[tile, 0, j] = sum over the tile's 128 rows of h[., j], [tile, 1, j] = sum of squares   (GemmProb::cstat's
//           format with one entry per 128 rows: colstats_from_blocks_kernel only adds the entries up)
//   PASS 2: out = x + rscale[row / rs_rows] * (relu(h * alpha + beta) W2^T), bf16, in place over x
// ---------------------------------------------------------------------------------------------------------------------
struct EncMlpArgs {
  const __bf16* x; const __bf16* yc; const __bf16* w1; const __bf16* w2;
  const float* alpha; const float* beta; const float* rscale; int rs_rows;
  float* cstat; __bf16* out; long rows;
};

template <int DIM, int HID, int CP, int PASS>
__global__ __launch_bounds__(256) void enc_mlp_bf16_kernel(const EncMlpArgs a) {
  constexpr int ROWS = 128, HC = 64, SA = DIM + 8, SH = HC + 8, NC = HID / HC, NT2 = DIM / 32;
  constexpr int UPR = DIM / 8, CPU = CP / 8;                     // 16-byte units per row of A; of them from yc
  constexpr int NA = ROWS * UPR / 256, NW1 = HC * UPR / 256, NW2 = DIM * (HC / 8) / 256;
  constexpr int NR = (ROWS * CPU + 255) / 256;
  static_assert(DIM % 32 == 0 && HID % HC == 0 && CP % 8 == 0 && ROWS * UPR % 256 == 0 && HC * UPR % 256 == 0 &&
                DIM * (HC / 8) % 256 == 0, "tile shapes");
  extern __shared__ __attribute__((aligned(16))) unsigned char mlp_lds[];
  __bf16* A = reinterpret_cast<__bf16*>(mlp_lds);                // [ROWS][SA]
  __bf16* W1c = A + ROWS * SA;                                   // [HC][SA]
  __bf16* Hc = W1c + HC * SA;                                    // [ROWS][SH]     (PASS 2)
  __bf16* W2c = Hc + ROWS * SH;                                  // [DIM][SH]      (PASS 2)
  __bf16* R = W2c + DIM * SH;                                    // [ROWS][CP]     (PASS 2: residual of the conv'd channels)
  float* ab = reinterpret_cast<float*>(R + ROWS * CP);           // [2][HID]       (PASS 2)
  float* rsl = ab + 2 * HID;                                     // [ROWS]         (PASS 2: DropPath factor of each row)
  float* st = reinterpret_cast<float*>(W1c + HC * SA);           // [4][2][HC]     (PASS 1: the four waves' column sums)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kh = lane >> 5;
  const long ntiles = (a.rows + ROWS - 1) / ROWS;
  // PFA = persistent blocks (grid = resident blocks, each walking tiles blockIdx.x, + gridDim.x, ...) that request the
  // next tile's operands under the current tile.  Built and measured in round 3, and OFF: the apply pass went from 224 to
  // 336 VGPRs (96-wide: one block per CU instead of two) and to 512 + spills (192-wide); the statistics pass kept its
  // registers but got SLOWER, 1.0 -> 1.64 ms per update at 360x256 — with one tile per block three 42 KB blocks share a
  // CU and hide each other's round trips better than two persistent ones with a prefetch.  One tile per block it stays.
  constexpr bool PFA = false;

  // PERSISTENT blocks: a block walks tiles blockIdx.x, + gridDim.x, ...; the next tile's operands (A, the residual's
  // first CP channels) are requested into registers right after the current tile has been parked, and weight chunk 0 of
  // the next tile under the current tile's last chunk, so a tile starts with its operands already on the chip.
  u32x4 va[NA], vr[PASS == 2 ? NR : 1], vw1[NW1], vw2[PASS == 2 ? NW2 : 1];
  auto fetch_a = [&](long r0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = t + 256 * i, row = f / UPR, u = f - row * UPR;
      const bool ok = r0 + row < a.rows;
      va[i] = u < CPU ? ld16_or_zero(a.yc, (r0 + row) * CP + u * 8, ok) : ld16_or_zero(a.x, (r0 + row) * DIM + u * 8, ok);
    }
    if constexpr (PASS == 2) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int f = t + 256 * i, row = f / CPU, u = f - row * CPU;
        const bool ok = f < ROWS * CPU && r0 + row < a.rows;
        vr[i] = ld16_or_zero(a.x, (r0 + row) * DIM + u * 8, ok);
      }
    }
  };
  auto park_a = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = t + 256 * i, row = f / UPR, u = f - row * UPR;
      *reinterpret_cast<u32x4*>(A + row * SA + u * 8) = va[i];
    }
    if constexpr (PASS == 2) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int f = t + 256 * i, row = f / CPU, u = f - row * CPU;
        if (f < ROWS * CPU) *reinterpret_cast<u32x4*>(R + row * CP + u * 8) = vr[i];
      }
    }
  };
  auto fetch_w = [&](int c) {
#pragma unroll
    for (int i = 0; i < NW1; ++i) {
      const int f = t + 256 * i, n = f / UPR, u = f - n * UPR;
      vw1[i] = *reinterpret_cast<const u32x4*>(a.w1 + (long)(c * HC + n) * DIM + u * 8);
    }
    if constexpr (PASS == 2) {
#pragma unroll
      for (int i = 0; i < NW2; ++i) {
        const int f = t + 256 * i, d = f / (HC / 8), u = f - d * (HC / 8);
        vw2[i] = *reinterpret_cast<const u32x4*>(a.w2 + (long)d * HID + c * HC + u * 8);
      }
    }
  };
  auto park_w = [&]() {
#pragma unroll
    for (int i = 0; i < NW1; ++i) {
      const int f = t + 256 * i, n = f / UPR, u = f - n * UPR;
      *reinterpret_cast<u32x4*>(W1c + n * SA + u * 8) = vw1[i];
    }
    if constexpr (PASS == 2) {
#pragma unroll
      for (int i = 0; i < NW2; ++i) {
        const int f = t + 256 * i, d = f / (HC / 8), u = f - d * (HC / 8);
        *reinterpret_cast<u32x4*>(W2c + d * SH + u * 8) = vw2[i];
      }
    }
  };
  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  fetch_a(tile * ROWS);
  fetch_w(0);
  if constexpr (PASS == 2) {
    for (int i = t; i < HID; i += 256) { ab[i] = a.alpha[i]; ab[HID + i] = a.beta[i]; }
  }
  const __bf16* arow = A + (32 * wave + li) * SA + kh * 8;
  for (; tile < ntiles; tile += PFA ? (long)gridDim.x : ntiles) {
    const long r0 = tile * ROWS;
    const bool has_next = PFA && tile + gridDim.x < ntiles;      // block-uniform; PASS 2: one tile per block
    park_a();
    park_w();
    if constexpr (PASS == 2) {
      // one (64-bit) division per row here instead of one per accumulator element in the epilogue
      if (t < ROWS) rsl[t] = (a.rscale && r0 + t < a.rows) ? a.rscale[(r0 + t) / a.rs_rows] : 1.f;
    }
    __syncthreads();
    if constexpr (PFA) { if (has_next) fetch_a((tile + gridDim.x) * ROWS); }      // in flight for the whole tile

    f32x16 acc2[PASS == 2 ? NT2 : 1];
    if constexpr (PASS == 2) {
#pragma unroll
      for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[n][r] = 0.f;
    }
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) fetch_w(c + 1);                            // in flight under this chunk's matrix work
      else if (has_next) fetch_w(0);                             // the next tile starts with its first chunk on the chip
      // ---- h chunk = A W1c^T: 32 rows x 64 hidden columns per wave ---------------------------------------------------
      f32x16 acc1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < DIM / 16; ++ks) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(arow + ks * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 fb = *reinterpret_cast<const bf16x8*>(W1c + (32 * j + li) * SA + ks * 16 + kh * 8);
          acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1[j], 0, 0, 0);
        }
      }
      if constexpr (PASS == 1) {
        // rows past the end of the tensor were loaded as zeros and h has no bias: they add nothing to either sum
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float cs = 0.f, cq = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { cs += acc1[j][r]; cq = fmaf(acc1[j][r], acc1[j][r], cq); }
          cs += __shfl_xor(cs, 32);
          cq += __shfl_xor(cq, 32);
          if (kh == 0) { st[(wave * 2 + 0) * HC + 32 * j + li] = cs; st[(wave * 2 + 1) * HC + 32 * j + li] = cq; }
        }
        __syncthreads();
        if (t < 2 * HC) {                                        // one entry per 128-row tile: waves added in order
          const int q = t / HC, col = t - q * HC;
          const float v = ((st[(0 * 2 + q) * HC + col] + st[(1 * 2 + q) * HC + col]) + st[(2 * 2 + q) * HC + col]) + st[(3 * 2 + q) * HC + col];
          a.cstat[(tile * 2 + q) * HID + c * HC + col] = v;
        }
        if (PFA && c + 1 == NC) __syncthreads();                 // (st is rewritten only after the next chunk's barriers; a
                                                                 //  persistent block's next tile starts with none)
      } else {
        // ---- BatchNorm (folded scale / shift) + ReLU on the fp32 accumulators, bf16 into this wave's rows of Hc -------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = c * HC + 32 * j + li;
          const float al = ab[col], be = ab[HID + col];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            Hc[(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh) * SH + 32 * j + li] = (__bf16)fmaxf(fmaf(acc1[j][r], al, be), 0.f);
        }
        // (the wave reads back only rows it wrote itself; LDS operations of one wave complete in order)
        const __bf16* hrow = Hc + (32 * wave + li) * SH + kh * 8;
#pragma unroll
        for (int ks = 0; ks < HC / 16; ++ks) {
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(hrow + ks * 16);
#pragma unroll
          for (int n = 0; n < NT2; ++n) {
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(W2c + (32 * n + li) * SH + ks * 16 + kh * 8);
            acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2[n], 0, 0, 0);
          }
        }
      }
      if (c + 1 < NC) {
        __syncthreads();                                         // every wave is done with this chunk's weights
        park_w();
        __syncthreads();
      }
    }
    if constexpr (PASS == 2) {
      // ---- x + rscale * y2, bf16, through this wave's rows of the A tile, out as 16-byte pieces ----------------------------
#pragma unroll
      for (int n = 0; n < NT2; ++n) {
        const int d = 32 * n + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const float xv = bf2f(d < CP ? R[row * CP + d] : A[row * SA + d]);
          A[row * SA + d] = (__bf16)(xv + rsl[row] * acc2[n][r]);
        }
      }
      for (int f = lane; f < 32 * UPR; f += 64) {
        const int row = 32 * wave + f / UPR, u = f % UPR;
        if (r0 + row < a.rows)
          *reinterpret_cast<u32x4*>(a.out + (r0 + row) * DIM + u * 8) = *reinterpret_cast<const u32x4*>(A + row * SA + u * 8);
      }
    }
    if (has_next) __syncthreads();                               // every wave is done with this tile's images
  }
}

template <int DIM, int HID, int CP>
constexpr int enc_mlp_lds_bytes(int pass) {
  return pass == 1 ? 2 * (128 * (DIM + 8) + 64 * (DIM + 8)) + 4 * 2 * 64 * 4
                   : 2 * (128 * (DIM + 8) + 64 * (DIM + 8) + 128 * 72 + DIM * 72 + 128 * CP) + 8 * HID + 4 * 128;
}

// ---------------------------------------------------------------------------------------------------------------------
// PatchMerging 2x2s2 (fasternet.py:253) on bf16: out (B*H2*W2, 2E) = patches(x1) Wm^T, K = 4 taps x E, plus the per-32-row
// column statistics of the (pre-BatchNorm) result.  A block owns 64 output positions: their 4 x E inputs are gathered
// into LDS once, the weight (2E, 4E; k = (ky, kx, ci)) walks through LDS one tap (2E x E) at a time, requested a tap ahead.
// ---------------------------------------------------------------------------------------------------------------------
struct EncMergeArgs {
  const __bf16* x1; const __bf16* w; __bf16* out; float* cstat;
  int Hp, Wp, H2, W2; long rows2;
};

template <int E>
__global__ __launch_bounds__(256) void enc_merge_bf16_kernel(const EncMergeArgs a) {
  constexpr int E2 = 2 * E, K = 4 * E, SA = K + 8, SW = E + 8, UE = E / 8, NT = E2 / 32, NTW = NT / 2;
  constexpr int NAU = 64 * 4 * UE / 256, NWU = E2 * UE / 256;
  static_assert(E % 32 == 0 && 64 * 4 * UE % 256 == 0 && E2 * UE % 256 == 0, "shapes");
  extern __shared__ __attribute__((aligned(16))) unsigned char mg_lds[];
  __bf16* A = reinterpret_cast<__bf16*>(mg_lds);                 // [64][SA]
  __bf16* Wc = A + 64 * SA;                                      // [E2][SW]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kh = lane >> 5;
  const long r0 = (long)blockIdx.x * 64;
  const int P2 = a.H2 * a.W2;
  u32x4 va[NAU], vw[NWU];
#pragma unroll
  for (int i = 0; i < NAU; ++i) {
    const int f = t + 256 * i, row = f / (4 * UE), rem = f - row * 4 * UE, tap = rem / UE, u = rem - tap * UE;
    const long r2 = r0 + row;
    const bool ok = r2 < a.rows2;
    const unsigned rr = ok ? (unsigned)r2 : 0u;                  // rows2 < 2^31 (host check): 32-bit divisions
    const unsigned b = rr / (unsigned)P2;
    const int q = (int)(rr - b * (unsigned)P2), oy = q / a.W2, ox = q - oy * a.W2;
    const long pix = ((long)b * a.Hp + 2 * oy + (tap >> 1)) * a.Wp + 2 * ox + (tap & 1);
    va[i] = ld16_or_zero(a.x1, pix * E + u * 8, ok);
  }
  auto fetch_w = [&](int tap) {
#pragma unroll
    for (int i = 0; i < NWU; ++i) {
      const int f = t + 256 * i, n = f / UE, u = f - n * UE;
      vw[i] = *reinterpret_cast<const u32x4*>(a.w + (long)n * K + tap * E + u * 8);
    }
  };
  auto park_w = [&]() {
#pragma unroll
    for (int i = 0; i < NWU; ++i) {
      const int f = t + 256 * i, n = f / UE, u = f - n * UE;
      *reinterpret_cast<u32x4*>(Wc + n * SW + u * 8) = vw[i];
    }
  };
  fetch_w(0);
#pragma unroll
  for (int i = 0; i < NAU; ++i) {
    const int f = t + 256 * i, row = f / (4 * UE), rem = f - row * 4 * UE;
    *reinterpret_cast<u32x4*>(A + row * SA + rem * 8) = va[i];
  }
  park_w();
  __syncthreads();
  const int mt = wave & 1, nb = wave >> 1;                       // n-tiles nb, nb + 2, ...
  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const __bf16* arow = A + (32 * mt + li) * SA + kh * 8;
  for (int tap = 0; tap < 4; ++tap) {
    if (tap < 3) fetch_w(tap + 1);
#pragma unroll
    for (int ks = 0; ks < E / 16; ++ks) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(arow + tap * E + ks * 16);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(Wc + (32 * (nb + 2 * j) + li) * SW + ks * 16 + kh * 8);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
      }
    }
    if (tap < 3) {
      __syncthreads();
      park_w();
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int col = 32 * (nb + 2 * j) + li;
    float cs = 0.f, cq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = r0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
      // the statistics are taken of the value that is STORED (bf16), which is what the BatchNorm sweep will read
      const __bf16 hv = (__bf16)acc[j][r];
      const float v = row < a.rows2 ? bf2f(hv) : 0.f;
      cs += v;
      cq = fmaf(v, v, cq);
      if (row < a.rows2) a.out[row * E2 + col] = hv;
    }
    cs += __shfl_xor(cs, 32);
    cq += __shfl_xor(cq, 32);
    if (kh == 0 && r0 + 32 * mt < a.rows2 && a.cstat) {
      float* o = a.cstat + ((r0 >> 5) + mt) * 2 * E2 + col;
      o[0] = cs;
      o[E2] = cq;
    }
  }
}

// y = x * alpha[c] + beta[c] on a (rows, C) bf16 tensor, in place (the PatchMerging BatchNorm); n8 = rows * C / 8
__global__ __launch_bounds__(256) void bn_apply_bf16_kernel(__bf16* __restrict__ x, long n8, int C8,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    bf16x8 v = reinterpret_cast<bf16x8*>(x)[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (__bf16)fmaf((float)v[k], alpha[c + k], beta[c + k]);
    reinterpret_cast<bf16x8*>(x)[i] = v;
  }
}

// AdaptiveAvgPool2d(1) over a (B, P, C) bf16 tensor -> (B, C) fp32: thread = (8 channels, one of 256 / (C/8) position
// lanes), 16-byte loads, fp64 partial sums combined in lane order like gap_kernel.  C % 8 == 0, C / 8 <= 256.
__global__ __launch_bounds__(256) void gap_bf16_kernel(const __bf16* __restrict__ x, float* __restrict__ out, int P, int C) {
  __shared__ double part[256][8];
  const int C8 = C >> 3, lanes = 256 / C8, t = threadIdx.x;
  const int c8 = t % C8, pl = t / C8;
  const long b = blockIdx.x;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pl < lanes)
    for (int p = pl; p < P; p += lanes) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (b * P + p) * C + c8 * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += (double)(float)v[k];
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) part[t][k] = s[k];
  __syncthreads();
  if (t < C) {
    const int cc = t >> 3, k = t & 7;
    double tt = 0;
    for (int l = 0; l < lanes; ++l) tt += part[l * C8 + cc][k];
    out[b * C + t] = (float)(tt / (double)P);
  }
}

}  // namespace porl
