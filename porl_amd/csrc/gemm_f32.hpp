// Grouped fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), gfx950 only.
//
// One launch runs up to MAX_GROUP independent GEMMs ("problems") that share an operand layout
// and a tile shape: the twin value nets / target nets / policy net of one update step are
// 1024x1024x1024-class problems, so a single problem has only 64 tiles of 128x128 and cannot fill
// 256 CUs by itself — grouping is what fills the chip.
//
// Arithmetic is exact fp32: the f32-input MFMA is bit-for-bit a k-ordered fmaf chain
// (cdna_hip_programming.md §3 "FP32-input MFMA"); no xf32/bf16 shortcuts.
//
// Operand layouts (row-major everywhere, like torch):
//   A_KC=true : A is (M, K), k contiguous        A_KC=false: A is (K, M), m contiguous
//   B_KC=true : B is (N, K), k contiguous        B_KC=false: B is (K, N), n contiguous
//   forward  Y = X W^T        : A_KC, B_KC      ("NT")   X (B,K_in), W (N_out,K_in)
//   dgrad    dX = dY W        : A_KC, !B_KC     ("NN")   dY (B,N_out), W (N_out,K_in)
//   wgrad    dW = dY^T X      : !A_KC, !B_KC    ("TN")   dY (B,N_out), X (B,K_in), K = batch
// The layout is a per-problem runtime flag (wave-uniform branches in the loaders), so forward,
// dgrad and wgrad problems can share one launch — e.g. the value backward runs dW1 (TN) and dZ0 (NN)
// of both twins as one 256-tile grid.
//
// LDS image is k-major for both operands (As[k][m], Bs[k][n]); k-contiguous sources are transposed
// while being written (ds_write_b32), m/n-contiguous sources are copied (ds_write_b128).  A wave
// reads one dword per operand per MFMA (lanes 0-31: k, lanes 32-63: k+1) — conflict-free, and at
// 64 cycles per MFMA the LDS is <15% busy, so no swizzle is needed for fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace porl {

enum : int { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

// A-operand prologue (applied to the value loaded from memory, before it is staged in LDS)
enum : int {
  APRO_NONE = 0,
  // a(b, j) = rowscale[b] * colscale[j] * 1[A(b, j) > 0]   — dZ of the scalar V head made on the fly:
  // rowscale = dL/dv (B,), colscale = w_out (H,), A = post-ReLU activations of the last hidden layer.
  APRO_RANK1_MASK = 1,
};

struct GemmProb {
  const float* A;
  const float* B;
  float* C;               // output, or slab base when splitk > 1 (slab s at C + s*M*ldc)
  const float* bias;      // (N,) added before the activation, or null
  const float* mask;      // (M,N) ld=ldmask: C = acc * 1[mask > 0] (ReLU backward), or null
  const float* headw;     // (N,) fused scalar head: headout[part][m] = sum_n C(m,n)*headw[n]
  float* headout;         // (parts, M) with parts = tiles_n * 2
  float* colsum;          // !A_KC only: (M,) column sums of A over K (bias gradient); slab s at +s*M
  const float* a_rowscale;  // APRO_RANK1_MASK
  const float* a_colscale;  // APRO_RANK1_MASK
  int M, N, K;
  int lda, ldb, ldc, ldmask;
  int act;
  int apro;
  int splitk;
  int store_c;
  int a_vec, b_vec;       // 16-byte loads allowed (pointer and ld aligned)
  int a_kc, b_kc;         // operand is k-contiguous ((M,K)/(N,K)) vs m/n-contiguous ((K,M)/(K,N))
  // filled by the host planner
  int tiles_m, tiles_n, block_start, kchunk;
};

constexpr int MAX_GROUP = 8;
struct GemmGroup {
  int nprob;
  int total_blocks;
  GemmProb p[MAX_GROUP];
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_THREADS = 256;

template <int BK>
__host__ __device__ constexpr int lds_stride(int R) { return R + 4; }

// Per-thread description of one staged float4 (4 consecutive elements of the operand's contiguous
// dimension).  Built once before the K loop from the problem's layout flag, so the loop itself is
// straight-line code: loads are issued back to back and waited for only after the MFMAs.
struct StageSlot {
  const float* ptr;   // address of element 0 in K-tile 0
  int kpos;           // k index of element 0 in K-tile 0 (absolute)
  int cpos;           // index along the non-k dimension of element 0
  int lds;            // LDS offset (floats) of element 0
};

// BM x BN block tile, 4 waves as 2(M) x 2(N), wave tile (BM/2) x (BN/2) made of 32x32 MFMA tiles.
//   VEC : every operand of every problem may be read with 16-byte loads (pointer, leading dimension
//         and contiguous extent are multiples of 4 floats) — the fast path.  VEC=false reads dwords.
//   APRO: A-operand prologue enabled (APRO_RANK1_MASK) for every problem of the group.
template <int BM, int BN, int BK, bool VEC, bool APRO>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_kernel(const GemmGroup g) {
  constexpr int WTM = BM / 64;            // MFMA tiles per wave along M
  constexpr int WTN = BN / 64;
  constexpr int SA = lds_stride<BK>(BM);
  constexpr int SB = lds_stride<BK>(BN);
  constexpr int A_TILE = BK * SA;
  constexpr int B_TILE = BK * SB;
  constexpr int NLA = BM * BK / 4 / GEMM_THREADS;   // float4 slots per thread per tile
  constexpr int NLB = BN * BK / 4 / GEMM_THREADS;
  static_assert(NLA >= 1 && NLB >= 1, "tile too small for 256 threads");
  static_assert(BK % 4 == 0 && BM % 64 == 0 && BN % 64 == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) float lds[2 * (A_TILE + B_TILE)];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  // ---- XCD-aware block -> work mapping (blocks b, b+8, ... share an XCD / L2) ------------------
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
  const int tiles = P.tiles_m * P.tiles_n;
  const int split = local / tiles;
  const int tl = local - split * tiles;
  const int tm = tl / P.tiles_n, tn = tl - tm * P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = P.M, N = P.N;
  const int ks = split * P.kchunk;
  const int ke = min(P.K, ks + P.kchunk);
  const int nkt = (ke - ks + BK - 1) / BK;
  const float* __restrict__ Ag = P.A;
  const float* __restrict__ Bg = P.B;
  const bool a_kc = P.a_kc != 0, b_kc = P.b_kc != 0;   // wave-uniform (per problem)
  const bool do_colsum = (!a_kc) && (P.colsum != nullptr) && (tn == 0);

  // ---- staging slots -------------------------------------------------------------------------------
  StageSlot sa[NLA], sb[NLB];
  // element j of a slot sits at k = kpos + j*a_dk, c = cpos + j*a_dc (one of dk/dc is 1, the other 0)
  const int a_dk = a_kc ? 1 : 0, a_dc = 1 - a_dk;
  const int b_dk = b_kc ? 1 : 0, b_dc = 1 - b_dk;
  const size_t a_step = a_kc ? (size_t)BK : (size_t)BK * P.lda;   // pointer advance per K-tile
  const size_t b_step = b_kc ? (size_t)BK : (size_t)BK * P.ldb;
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int f = t + GEMM_THREADS * i;
    if (a_kc) {
      const int kq = f % (BK / 4), row = f / (BK / 4);
      sa[i].kpos = ks + kq * 4; sa[i].cpos = m0 + row; sa[i].lds = (kq * 4) * SA + row;
      sa[i].ptr = Ag + (size_t)(m0 + row) * P.lda + (ks + kq * 4);
    } else {
      const int c4 = f % (BM / 4), kr = f / (BM / 4);
      sa[i].kpos = ks + kr; sa[i].cpos = m0 + c4 * 4; sa[i].lds = kr * SA + c4 * 4;
      sa[i].ptr = Ag + (size_t)(ks + kr) * P.lda + (m0 + c4 * 4);
    }
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int f = t + GEMM_THREADS * i;
    if (b_kc) {
      const int kq = f % (BK / 4), row = f / (BK / 4);
      sb[i].kpos = ks + kq * 4; sb[i].cpos = n0 + row; sb[i].lds = (kq * 4) * SB + row;
      sb[i].ptr = Bg + (size_t)(n0 + row) * P.ldb + (ks + kq * 4);
    } else {
      const int c4 = f % (BN / 4), kr = f / (BN / 4);
      sb[i].kpos = ks + kr; sb[i].cpos = n0 + c4 * 4; sb[i].lds = kr * SB + c4 * 4;
      sb[i].ptr = Bg + (size_t)(ks + kr) * P.ldb + (n0 + c4 * 4);
    }
  }

  // Guarded slot access without divergence.  `slot_ok` gives per-element validity for K-tile kt;
  // `load_raw` issues the loads (out-of-range lanes read `safe`) and does NOT touch the result, so no
  // wait is needed until `store_tile` zeroes the invalid elements and parks the tile in LDS.
  auto slot_ok = [&](const StageSlot& s, int kadv, int dk, int dc, int cmax, bool (&ok)[4]) {
    const int k = s.kpos + kadv;
    if constexpr (VEC) {
      ok[0] = ok[1] = ok[2] = ok[3] = (k < ke) && (s.cpos < cmax);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) ok[j] = (k + j * dk < ke) && (s.cpos + j * dc < cmax);
    }
  };
  auto load_raw = [&](const StageSlot& s, const float* __restrict__ safe, size_t off, const bool (&ok)[4]) -> float4 {
    float4 v;
    const float* p = s.ptr + off;
    if constexpr (VEC) {
      v = *reinterpret_cast<const float4*>(ok[0] ? p : safe);
    } else {
      v.x = *(ok[0] ? p : safe); v.y = *(ok[1] ? p + 1 : safe);
      v.z = *(ok[2] ? p + 2 : safe); v.w = *(ok[3] ? p + 3 : safe);
    }
    return v;
  };

  float4 ra[NLA], rb[NLB];
  float pr_rs[APRO ? NLA : 1];       // prologue row scale (one per slot)
  float4 pr_cs[APRO ? NLA : 1];      // prologue column scales
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

  // rank-1 prologue: element (k, c) of A becomes rowscale[b] * colscale[j] * 1[A > 0] where
  // (b, j) = (c, k) for k-contiguous A (dgrad) and (k, c) for m-contiguous A (wgrad); the 4 elements
  // of a slot always run along j (the contiguous, hidden dimension).
  const float* __restrict__ rsc = APRO ? P.a_rowscale : nullptr;
  const float* __restrict__ csc = APRO ? P.a_colscale : nullptr;

  auto load_tile = [&](int kt) {
    const int kadv = kt * BK;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      bool ok[4];
      slot_ok(sa[i], kadv, a_dk, a_dc, M, ok);
      ra[i] = load_raw(sa[i], Ag, (size_t)kt * a_step, ok);
      if constexpr (APRO) {
        const int k = sa[i].kpos + kadv, c = sa[i].cpos;
        const int b0 = a_kc ? c : k, j0 = a_kc ? k : c;
        pr_rs[i] = *(ok[0] ? rsc + b0 : rsc);
        pr_cs[i].x = *(ok[0] ? csc + j0 : csc);
        pr_cs[i].y = *(ok[1] ? csc + j0 + 1 : csc);
        pr_cs[i].z = *(ok[2] ? csc + j0 + 2 : csc);
        pr_cs[i].w = *(ok[3] ? csc + j0 + 3 : csc);
      }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      bool ok[4];
      slot_ok(sb[i], kadv, b_dk, b_dc, N, ok);
      rb[i] = load_raw(sb[i], Bg, (size_t)kt * b_step, ok);
    }
  };

  // kt = the K-tile the registers hold
  auto store_tile = [&](int kt) {
    const int kadv = kt * BK;
    float* As = lds + (kt & 1) * (A_TILE + B_TILE);
    float* Bs = As + A_TILE;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      bool ok[4];
      slot_ok(sa[i], kadv, a_dk, a_dc, M, ok);
      float4 v = ra[i];
      if constexpr (APRO) {
        const float rs = pr_rs[i];
        v.x = v.x > 0.f ? rs * pr_cs[i].x : 0.f;
        v.y = v.y > 0.f ? rs * pr_cs[i].y : 0.f;
        v.z = v.z > 0.f ? rs * pr_cs[i].z : 0.f;
        v.w = v.w > 0.f ? rs * pr_cs[i].w : 0.f;
      }
      v.x = ok[0] ? v.x : 0.f; v.y = ok[1] ? v.y : 0.f; v.z = ok[2] ? v.z : 0.f; v.w = ok[3] ? v.w : 0.f;
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      bool ok[4];
      slot_ok(sb[i], kadv, b_dk, b_dc, N, ok);
      float4 v = rb[i];
      v.x = ok[0] ? v.x : 0.f; v.y = ok[1] ? v.y : 0.f; v.z = ok[2] ? v.z : 0.f; v.w = ok[3] ? v.w : 0.f;
      rb[i] = v;
    }
    if (a_kc) {
#pragma unroll
      for (int i = 0; i < NLA; ++i) {
        As[sa[i].lds] = ra[i].x;
        As[sa[i].lds + SA] = ra[i].y;
        As[sa[i].lds + 2 * SA] = ra[i].z;
        As[sa[i].lds + 3 * SA] = ra[i].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NLA; ++i) {
        *reinterpret_cast<float4*>(As + sa[i].lds) = ra[i];
        csum.x += ra[i].x; csum.y += ra[i].y; csum.z += ra[i].z; csum.w += ra[i].w;
      }
    }
    if (b_kc) {
#pragma unroll
      for (int i = 0; i < NLB; ++i) {
        Bs[sb[i].lds] = rb[i].x;
        Bs[sb[i].lds + SB] = rb[i].y;
        Bs[sb[i].lds + 2 * SB] = rb[i].z;
        Bs[sb[i].lds + 3 * SB] = rb[i].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NLB; ++i) *reinterpret_cast<float4*>(Bs + sb[i].lds) = rb[i];
    }
  };

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nkt > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  const int a_off = wm * (BM / 2) + li;
  const int b_off = wn * (BN / 2) + li;

  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt+1 is requested before the MFMAs of tile kt and parked in LDS after them; past the last
    // tile every lane is out of range, so the loads degenerate to reads of the operand base.
    load_tile(kt + 1);
    __builtin_amdgcn_sched_barrier(0);   // keep the global loads ahead of the MFMA block
    const float* As = lds + (kt & 1) * (A_TILE + B_TILE);
    const float* Bs = As + A_TILE;
    // fragments are double-buffered in registers: k-step kk+2 is read while kk is multiplied
    float a[2][WTM], b[2][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i) a[0][i] = As[kh * SA + a_off + i * 32];
#pragma unroll
    for (int j = 0; j < WTN; ++j) b[0][j] = Bs[kh * SB + b_off + j * 32];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
      if (kk + 2 < BK) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) a[nxt][i] = As[(kk + 2 + kh) * SA + a_off + i * 32];
#pragma unroll
        for (int j = 0; j < WTN; ++j) b[nxt][j] = Bs[(kk + 2 + kh) * SB + b_off + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);   // next fragments are requested before these MFMAs issue
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_tile(kt + 1);
    __syncthreads();
  }

  // ---- bias gradient: column sums of A (only tn == 0 blocks), reduced through LDS --------------
  // (csum also swallowed the all-zero phantom tile nkt, which adds nothing)
  if (do_colsum) {
    float4* red = reinterpret_cast<float4*>(lds);
    red[t] = csum;
    __syncthreads();
    constexpr int C4 = BM / 4;                       // threads t, t+C4, ... share a column group
    if (t < C4) {
      float4 s = red[t];
      for (int u = t + C4; u < GEMM_THREADS; u += C4) {
        const float4 o = red[u];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      float* out = P.colsum + (size_t)split * M;
      const int c = m0 + t * 4;
      if (c < M) out[c] = s.x;
      if (c + 1 < M) out[c + 1] = s.y;
      if (c + 2 < M) out[c + 2] = s.z;
      if (c + 3 < M) out[c + 3] = s.w;
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  const bool raw = P.splitk > 1;
  float* __restrict__ Cg = P.C + (raw ? (size_t)split * M * P.ldc : 0);
  const bool has_head = (!raw) && P.headw != nullptr;
  const float* __restrict__ maskp = raw ? nullptr : P.mask;
  const int act = raw ? ACT_NONE : P.act;
  const bool store_c = P.store_c != 0;
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    float hsum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hsum[r] = 0.f;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + li;
      const bool col_ok = col < N;
      float bv = 0.f, hw = 0.f;
      if (!raw && col_ok) {
        if (P.bias) bv = P.bias[col];
        if (has_head) hw = P.headw[col];
      }
      const int rbase = m0 + wm * (BM / 2) + i * 32 + 4 * kh;
      float vals[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv;
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ACT_TANH) v = tanhf(v);
        vals[r] = v;
      }
      if (maskp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          const bool ok = row < M && col_ok;
          const float mv = *(ok ? maskp + (size_t)row * P.ldmask + col : maskp);
          vals[r] = (ok && mv > 0.f) ? vals[r] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) hsum[r] += vals[r] * hw;
      if (store_c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < M && col_ok) Cg[(size_t)row * P.ldc + col] = vals[r];
        }
      }
    }
    if (has_head) {
      // reduce over the 32 columns held by each half-wave, then one lane per half writes 16 rows
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = hsum[r];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 1);
        hsum[r] = s;
      }
      if (li == 0) {
        const int part = tn * 2 + wn;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
          if (row < M) P.headout[(size_t)part * M + row] = hsum[r];
        }
      }
    }
  }
}

// out[i] = act( sum_s slab[s*stride + i] + bias[i % ncols] )   (split-K combine, fixed order)
__global__ void slab_reduce_kernel(float* __restrict__ out, const float* __restrict__ slab, int nslab,
                                   long n, long stride, const float* __restrict__ bias, int ncols, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = slab[i];
    for (int k = 1; k < nslab; ++k) s += slab[(long)k * stride + i];
    if (bias) s += bias[i % ncols];
    if (act == ACT_RELU) s = fmaxf(s, 0.f);
    else if (act == ACT_TANH) s = tanhf(s);
    out[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
enum GemmMode : int { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
enum GemmTile : int { TILE_128x128 = 0, TILE_128x64 = 1, TILE_64x128 = 2, TILE_64x64 = 3 };

constexpr int GEMM_BK = 16;

inline void tile_dims(int tile, int& bm, int& bn) {
  switch (tile) {
    case TILE_128x128: bm = 128; bn = 128; break;
    case TILE_128x64: bm = 128; bn = 64; break;
    case TILE_64x128: bm = 64; bn = 128; break;
    default: bm = 64; bn = 64; break;
  }
}

inline int head_parts(int N, int tile) {
  int bm, bn;
  tile_dims(tile, bm, bn);
  return ((N + bn - 1) / bn) * 2;
}

// Fill the planner fields; returns total blocks.
inline int plan_group(GemmGroup& g, int tile) {
  int bm, bn;
  tile_dims(tile, bm, bn);
  int start = 0;
  for (int i = 0; i < g.nprob; ++i) {
    GemmProb& p = g.p[i];
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = (p.N + bn - 1) / bn;
    if (p.splitk < 1) p.splitk = 1;
    int kc = (p.K + p.splitk - 1) / p.splitk;
    kc = ((kc + GEMM_BK - 1) / GEMM_BK) * GEMM_BK;
    p.kchunk = kc;
    p.block_start = start;
    start += p.tiles_m * p.tiles_n * p.splitk;
  }
  g.total_blocks = start;
  return start;
}

template <int BM, int BN>
inline hipError_t launch_tile(const GemmGroup& g, hipStream_t s) {
  dim3 grid(g.total_blocks), block(GEMM_THREADS);
  bool vec = true, apro = g.p[0].apro != APRO_NONE;
  for (int i = 0; i < g.nprob; ++i) {
    vec = vec && g.p[i].a_vec && g.p[i].b_vec;
    if ((g.p[i].apro != APRO_NONE) != apro) return hipErrorInvalidValue;   // a group shares the prologue
  }
  if (vec && !apro) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, GEMM_BK, true, false>), grid, block, 0, s, g);
  else if (vec && apro) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, GEMM_BK, true, true>), grid, block, 0, s, g);
  else if (!apro) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, GEMM_BK, false, false>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, GEMM_BK, false, true>), grid, block, 0, s, g);
  return hipGetLastError();
}

// Problems of different modes (NT/NN/TN) may share one launch; only the tile shape is common.
inline hipError_t launch_gemm_group(int tile, GemmGroup& g, hipStream_t s) {
  if (g.nprob < 1 || g.nprob > MAX_GROUP) return hipErrorInvalidValue;
  if (plan_group(g, tile) == 0) return hipSuccess;
  switch (tile) {
    case TILE_128x128: return launch_tile<128, 128>(g, s);
    case TILE_128x64: return launch_tile<128, 64>(g, s);
    case TILE_64x128: return launch_tile<64, 128>(g, s);
    case TILE_64x64: return launch_tile<64, 64>(g, s);
  }
  return hipErrorInvalidValue;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Convenience: a problem with defaults; the caller overrides epilogue fields.
inline GemmProb make_prob(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                          int N, int K) {
  GemmProb p{};
  p.a_kc = (mode == GEMM_NT || mode == GEMM_NN);
  p.b_kc = (mode == GEMM_NT);
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.act = ACT_NONE; p.apro = APRO_NONE; p.splitk = 1; p.store_c = 1;
  // contiguous extent: K for k-contiguous operands, M / N otherwise
  p.a_vec = aligned16(A) && (lda % 4 == 0) && ((p.a_kc ? K : M) % 4 == 0);
  p.b_vec = aligned16(B) && (ldb % 4 == 0) && ((p.b_kc ? K : N) % 4 == 0);
  return p;
}

}  // namespace porl
