// One-launch CQL(H) gradient for small Q-networks (reference src/porl/train/cql_trainer.py:88-124, config 3:
// 60 -> 64 -> 128 -> 64 -> 10 at batch 4096).  Every layer is at most 128 wide, so a block keeps 32 minibatch
// rows, one layer's weights and all activations of those rows in LDS and walks the whole step itself:
//   target net on s' -> online net on s -> TD target, logsumexp penalty, dL/dQ -> backward through the layers.
// The multi-launch path (porl_api.hip) spends ~7 us per dependent launch on ~20 launches of a few us each; this
// kernel is latency-bound inside one block instead.  Matrix work runs on v_mfma_f32_32x32x2_f32 (exact fp32
// fmaf chains): forward and dgrad tiles are 32 rows x 32 columns per wave, wgrad tiles 32 x 32 of dW per wave
// with the 32 rows as the reduction.  Per-block partial gradients go to a slab in the flat parameter layout
// and are summed in block order by qnet_reduce_kernel (deterministic).
//
// LDS images are row-major with stride = width + 4 floats (16-byte rows, conflict-free 8-byte fragment reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

namespace porl {

constexpr int QF_ROWS = 32;          // minibatch rows per block = one MFMA tile
constexpr int QF_MAX_LIN = 5;        // Linear layers (hidden + output)
constexpr int QF_MAX_W = 128;        // widest layer
constexpr int QF_MAX_LDS_BYTES = 160 * 1024 - 1024;  // dynamic LDS budget (the kernel also has ~0.5 KB of static LDS)

struct QnetFusedArgs {
  const float* params;               // online, flat (W (out,in) row-major then bias, per layer)
  const float* params_tgt;
  // minibatch source: row b of the batch is row idx[b] (or b when idx is null) of these arrays
  const float* states; long s_rs;
  const float* next_states; long n_rs;
  const int64_t* actions; const float* rew; const float* done;
  const int64_t* idx;
  float* slab; long slab_stride;     // (blocks, slab_stride): partial gradients, flat parameter layout
  float* part_td; float* part_pen;   // (blocks,)
  int B, n_lin;
  int dims[QF_MAX_LIN + 1];          // dims[0] = state_dim, dims[n_lin] = n_actions
  long w_off[QF_MAX_LIN], b_off[QF_MAX_LIN];
  // LDS offsets (floats), strides = round32(width) + 4
  int lds_act[QF_MAX_LIN + 1];       // lds_act[0] = input rows, lds_act[l + 1] = output of layer l (online net)
  int lds_tmp[2];                    // ping-pong: target-net activations, then dZ
  int lds_w;                         // one layer's weights, (round32(out), round4(in) + 4)
  float gamma, alpha, inv_batch, log_A;
  // DQN variants on the same kernel (src/porl/train/dqn_per_trainer.py:75-123):
  int double_dqn;                    // target = Q_tgt(s')[argmax_a Q_online(s', a)] instead of max_a Q_tgt(s', a)
  const float* is_w;                 // (B,) per-sample loss weights (importance sampling), or null
  const float* w_uniform;            // device scalar multiplying every sample's loss (null = 1)
  float* td_abs;                     // (B,) |Q(s)[a] - target| for the priority write-back, or null
  // BCQ (src/porl/policy/bcq.py:50-86): the bootstrap action is argmax_a [Q_tgt(s', a) + (mask[b, a] - 1) * 1e10],
  // valued by Q_tgt — mask (B, n_actions) fp32 0/1 by minibatch position, from the behaviour policy; null = off
  const float* next_mask;
  int td_off;                        // 1: no TD term at all (loss = alpha * penalty: the cross-entropy pre-training of
                                     // the behaviour policy, bcq.py:23-47, is logsumexp(z) - z[a] = penalty + ln A)
  unsigned long long* stamps;        // diagnostics: shader-clock stamps of block 0 at the phase boundaries, or null
  // In-kernel sampling (two-group kernel only): with samp_n > 0 batch row b is replay row perm_{seed,step}(b) of
  // [0, samp_n) — the keyed Feistel permutation of porl_sample_indices (kernels.hpp: feistel_index) — computed by the
  // block's first lanes instead of being read from `idx`: no sampler launch, no index round trip at kernel entry.
  long samp_n; unsigned long long samp_seed, samp_step; int samp_hb;
  int wgrad_share;                   // 16-row kernel: sixteenths of a big layer's dW tiles left to the dW group (0 or 16: all)
};

typedef float qf_f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier that retires LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait at the end
// of every layer for the NEXT layer's weights (requested into registers just before the layer's MFMA work) — a full
// round trip to L2 per layer.  Nothing in this kernel communicates through global memory between barriers.
__device__ __forceinline__ void qf_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int qf_r32(int x) { return (x + 31) & ~31; }
__device__ __forceinline__ int qf_r4(int x) { return (x + 3) & ~3; }
// K extent of a weight image row: the MFMA loops run four k-steps (16 columns) per trip, zero padded
__device__ __forceinline__ int qf_rk(int x) { return (x + 15) & ~15; }
// i / d for 0 <= i < 2^16, 1 <= d <= 256 in three instructions (the half-unit margin dwarfs fp32 rounding)
__device__ __forceinline__ int qf_div(int i, float inv_d) { return __float2int_rz(((float)i + 0.5f) * inv_d); }

// Minibatch rows of s' and s -> LDS [32][ld] each, zero padded (columns and rows past B).  The source row numbers
// are read first (one round trip), then every element of both inputs is requested before the first LDS store.
constexpr int QF_XREGS = (QF_ROWS * (QF_MAX_W + 4) + 255) / 256;      // 17
__device__ __forceinline__ void qf_load_inputs(float* dst_n, float* dst_s, int ld, const float* next_states, long n_rs,
                                               const float* states, long s_rs, const int64_t* idx, int row0, int B,
                                               int cols, int t) {
  const int total = QF_ROWS * ld;
  const float inv_ld = 1.0f / (float)ld;
  float vn[QF_XREGS], vs[QF_XREGS];
#pragma unroll
  for (int u = 0; u < QF_XREGS; ++u) {
    if (u * 256 < total) {                            // uniform
      const int i = u * 256 + t;
      const int r = qf_div(i, inv_ld), c = i - r * ld;
      const int b = row0 + r;
      const bool ok = i < total && b < B && c < cols;
      const long row = idx ? idx[b < B ? b : 0] : (long)b;
      const float xn = next_states[ok ? row * n_rs + c : 0L];
      const float xs = states[ok ? row * s_rs + c : 0L];
      vn[u] = ok ? xn : 0.f;
      vs[u] = ok ? xs : 0.f;
    }
  }
#pragma unroll
  for (int u = 0; u < QF_XREGS; ++u) {
    const int i = u * 256 + t;
    if (u * 256 < total && i < total) { dst_n[i] = vn[u]; dst_s[i] = vs[u]; }
  }
}

// One layer's weights + bias -> LDS.  The flat parameter group stores every layer as the very image the kernel wants in
// LDS (porl_qnet_create: round32(N) rows of round16(K) + 4 floats, zero padded, then round32(N) bias floats), so
// staging is a linear copy: qf_fetch_w requests float4 number u * 256 + t of the image into registers (QF_WREGS per
// thread cover 128 x 132 + 128), qf_park_w writes them to LDS at the same index.  No per-element address arithmetic,
// no bounds selects (measured before: ~2.7 k cycles to ISSUE a layer's 17 requests, ~2.5 k to park them).  The kernel
// fetches stage j+1 before it computes stage j, so the round trip to L2 hides behind the MFMA work.
constexpr int QF_WREGS = ((QF_MAX_W * (QF_MAX_W + 4) + QF_MAX_W) / 4 + 255) / 256;      // 17
__device__ __forceinline__ int qf_image4(int N, int K) { return (qf_r32(N) * (qf_rk(K) + 4) + qf_r32(N)) >> 2; }
// (Fold expressions over a compile-time index pack: every slot is a straight-line statement with a constant index.)
template <int... U>
__device__ __forceinline__ void qf_fetch_impl(float4 (&v)[QF_WREGS], const float4* src, int total4, int t,
                                              std::integer_sequence<int, U...>) {
  // every slot issues its load unconditionally; slots past the image re-read float4 0 (one shared cache line) and
  // are replaced by zeros afterwards.  (Without that select the compiler kept `v` in scratch memory: global_load ->
  // s_waitcnt vmcnt(0) -> scratch_store per slot in the ISA, 85 scratch instructions, +10 % kernel time.)
  ((v[U] = (U * 256 + t < total4) ? src[U * 256 + t < total4 ? U * 256 + t : 0] : make_float4(0.f, 0.f, 0.f, 0.f)), ...);
}
template <int... U>
__device__ __forceinline__ void qf_park_impl(float4* dst, const float4 (&v)[QF_WREGS], int total4, int t,
                                             std::integer_sequence<int, U...>) {
  ((U * 256 + t < total4 ? (void)(dst[U * 256 + t] = v[U]) : (void)0), ...);
}
__device__ __forceinline__ void qf_fetch_w(float4 (&v)[QF_WREGS], const float* img, int N, int K, int t) {
  qf_fetch_impl(v, reinterpret_cast<const float4*>(img), qf_image4(N, K), t, std::make_integer_sequence<int, QF_WREGS>{});
}
__device__ __forceinline__ void qf_park_w(float* wl, const float4 (&v)[QF_WREGS], int N, int K, int t) {
  qf_park_impl(reinterpret_cast<float4*>(wl), v, qf_image4(N, K), t, std::make_integer_sequence<int, QF_WREGS>{});
}

// NB batches of 16 reduction columns of one 32 x 32 forward tile: two accumulator chains (a chain of dependent
// 32x32x2 MFMAs issues at half rate), float2 fragments (k, k+1 -> chain 0, chain 1)
template <int NB>
__device__ __forceinline__ void qf_fwd_batches(const float* ap, const float* bp, qf_f32x16& acc, qf_f32x16& acc1) {
  float2 av[NB][4], bv[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      av[b][j] = *reinterpret_cast<const float2*>(ap + 16 * b + 4 * j);
      bv[b][j] = *reinterpret_cast<const float2*>(bp + 16 * b + 4 * j);
    }
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][j].x, bv[b][j].x, acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][j].y, bv[b][j].y, acc1, 0, 0, 0);
    }
}

// NB batches of 16 reduction rows n of one 32 x 32 dgrad tile (see qf_dgrad for the index mapping)
template <int NB>
__device__ __forceinline__ void qf_dgrad_batches(const float* ap, const float* bp, int ldw, qf_f32x16& acc, qf_f32x16& acc1) {
  float4 av[NB][2];
  float bv[NB][8];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      av[b][i] = *reinterpret_cast<const float4*>(ap + 16 * b + 8 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[b][4 * i + j] = bp[(16 * b + 8 * i + j) * ldw];
    }
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][i].x, bv[b][4 * i + 0], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][i].y, bv[b][4 * i + 1], acc1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][i].z, bv[b][4 * i + 2], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][i].w, bv[b][4 * i + 3], acc1, 0, 0, 0);
    }
}

// out[32][ldo] = act(in[32][ldi] . Wl^T + bias): wave w computes the 32-column slabs w, w+4, ...
__device__ __forceinline__ void qf_forward(const float* in, int ldi, const float* wl, int K, int N, const float* bias /* LDS */,
                                           bool relu, float* out, int ldo, int wave, int li, int kh) {
  const int ldw = qf_rk(K) + 4, Kp = qf_rk(K);        // activations are zero up to round32(K) >= round16(K)
  for (int tn = wave; tn < qf_r32(N) / 32; tn += 4) {
    // two accumulators: a chain of dependent 32x32x2 MFMAs issues at half rate (the next one waits for the
    // previous result), two interleaved chains fill the matrix pipe
    qf_f32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
    const float* ap = in + li * ldi + 2 * kh;
    const float* bp = wl + (tn * 32 + li) * ldw + 2 * kh;
    // K runs in batches of 16 columns (8 LDS reads, 8 MFMAs); the loop over the batches is unrolled for the batch
    // count (1..8), so the compiler sees straight-line code and can request a tile's fragments well ahead of the MFMAs
    // that use them.  (A run-time loop paid one LDS round trip per batch: 6.5 k cycles for K = 128 against 4.1 k of
    // matrix work; a hand-rolled prefetch inside a run-time loop made the compiler copy both accumulators every trip.)
    // (at most four batches = 64 fragment registers in flight: all eight at once spilled)
    const int nb = Kp >> 4;
    if (nb > 4) {
      qf_fwd_batches<4>(ap, bp, acc, acc1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float* ap2 = nb > 4 ? ap + 64 : ap;
    const float* bp2 = nb > 4 ? bp + 64 : bp;
    switch (nb > 4 ? nb - 4 : nb) {
      case 1: qf_fwd_batches<1>(ap2, bp2, acc, acc1); break;
      case 2: qf_fwd_batches<2>(ap2, bp2, acc, acc1); break;
      case 3: qf_fwd_batches<3>(ap2, bp2, acc, acc1); break;
      default: qf_fwd_batches<4>(ap2, bp2, acc, acc1); break;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
    const int col = tn * 32 + li;
    const float bv = bias[col];                    // zero past N
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + bv;
      if (relu) v = fmaxf(v, 0.f);
      if (col >= N) v = 0.f;
      out[((r & 3) + 8 * (r >> 2) + 4 * kh) * ldo + col] = v;
    }
  }
}

// One row of the loss stage with the row's Q values in registers: NC float4 chunks cover the A <= 4 NC actions (the
// padding columns of Q are zero, rows are 16-byte aligned and 36 floats long when A <= 32).  Same arithmetic in the
// same order as the general branch of qf_loss_rows; dynamic positions (taken action, bootstrap action) are select
// chains.  (Reading q[j] inside run-time loops cost one LDS round trip per element: ~7 k cycles for 10 actions.)
template <int NC>
__device__ __forceinline__ void qf_loss_row_regs(const QnetFusedArgs& a, const float* q, const float* qn, float* dq, int A, int b,
                                                 int act, float row_rew, float row_done, int am, float& td, float& pen) {
  constexpr int W = 4 * NC;
  float qv[W], nv[W];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float4 x = *reinterpret_cast<const float4*>(q + 4 * c);
    const float4 y4 = *reinterpret_cast<const float4*>(qn + 4 * c);
    qv[4 * c] = x.x; qv[4 * c + 1] = x.y; qv[4 * c + 2] = x.z; qv[4 * c + 3] = x.w;
    nv[4 * c] = y4.x; nv[4 * c + 1] = y4.y; nv[4 * c + 2] = y4.z; nv[4 * c + 3] = y4.w;
  }
  float mx = -INFINITY, mxn = -INFINITY;
#pragma unroll
  for (int j = 0; j < W; ++j) if (j < A) { mx = fmaxf(mx, qv[j]); mxn = fmaxf(mxn, nv[j]); }
  float se = 0.f;
#pragma unroll
  for (int j = 0; j < W; ++j) if (j < A) se += expf(qv[j] - mx);
  const float lse = mx + logf(se);
  float qa = 0.f;
#pragma unroll
  for (int j = 0; j < W; ++j) qa = j == act ? qv[j] : qa;
  float qnext = mxn;
  if (a.double_dqn) {
#pragma unroll
    for (int j = 0; j < W; ++j) qnext = j == am ? nv[j] : qnext;
  }
  if (a.next_mask) {
    const float* mk = a.next_mask + (long)b * A;
    float bestv = nv[0] + (mk[0] - 1.f) * 1e10f;
    qnext = nv[0];
#pragma unroll
    for (int j = 1; j < W; ++j) {
      if (j < A) {
        const float v = nv[j] + (mk[j] - 1.f) * 1e10f;
        if (v > bestv) { bestv = v; qnext = nv[j]; }         // first maximum, like torch.argmax
      }
    }
  }
  const float y = row_rew + a.gamma * qnext * (1.f - row_done);
  const float diff = a.td_off ? 0.f : qa - y;
  float wgt = a.is_w ? a.is_w[b] : 1.f;
  if (a.w_uniform) wgt *= a.w_uniform[0];
  td = wgt * (diff * diff);
  pen = lse - a.log_A - qa;
  if (a.td_abs) a.td_abs[b] = fabsf(diff);
  const float ab = a.alpha * a.inv_batch;
  float g[W];
#pragma unroll
  for (int j = 0; j < W; ++j) {
    float gj = ab * expf(qv[j] - lse);
    if (j == act) gj += 2.f * a.inv_batch * wgt * diff - ab;
    g[j] = j < A ? gj : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 9; ++c)        // the whole 36-float row: dL/dQ, then zeros
    *reinterpret_cast<float4*>(dq + 4 * c) = c < NC ? make_float4(g[4 * (c < NC ? c : 0)], g[4 * (c < NC ? c : 0) + 1],
                                                                  g[4 * (c < NC ? c : 0) + 2], g[4 * (c < NC ? c : 0) + 3])
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- loss and dL/dQ: one lane per row  (cql_trainer.py:94-118; same arithmetic as cql_loss_kernel).  dq may be
// the Q buffer itself: a lane reads q[j] before it writes dq[j].  Leaves the block's partial sums in red[0..1].
__device__ __forceinline__ void qf_loss_rows(const QnetFusedArgs& a, const float* Q, const float* Qn, float* dz, const int* amax,
                                             int row_act, float row_rew, float row_done, int row0, int lane, int wave,
                                             float* red, int rows = QF_ROWS) {
  const int A = a.dims[a.n_lin], ldq = qf_r32(A) + 4;
  if (wave == 0) {
    float td = 0.f, pen = 0.f;
    if (lane < rows) {
      const int b = row0 + lane;
      float* dq = dz + lane * ldq;
      if (b < a.B && A <= 32) {
        const float* q = Q + lane * ldq;
        const float* qn = Qn + lane * ldq;
        const int am = a.double_dqn ? amax[lane] : 0;
        switch ((A + 3) >> 2) {
          case 1: qf_loss_row_regs<1>(a, q, qn, dq, A, b, row_act, row_rew, row_done, am, td, pen); break;
          case 2: qf_loss_row_regs<2>(a, q, qn, dq, A, b, row_act, row_rew, row_done, am, td, pen); break;
          case 3: qf_loss_row_regs<3>(a, q, qn, dq, A, b, row_act, row_rew, row_done, am, td, pen); break;
          case 4: qf_loss_row_regs<4>(a, q, qn, dq, A, b, row_act, row_rew, row_done, am, td, pen); break;
          case 5: case 6: qf_loss_row_regs<6>(a, q, qn, dq, A, b, row_act, row_rew, row_done, am, td, pen); break;
          default: qf_loss_row_regs<8>(a, q, qn, dq, A, b, row_act, row_rew, row_done, am, td, pen); break;
        }
      } else if (b < a.B) {
        const float* q = Q + lane * ldq;
        const float* qn = Qn + lane * ldq;
        float mx = -INFINITY, mxn = -INFINITY;
        for (int j = 0; j < A; ++j) { mx = fmaxf(mx, q[j]); mxn = fmaxf(mxn, qn[j]); }
        float se = 0.f;
        for (int j = 0; j < A; ++j) se += expf(q[j] - mx);
        const float lse = mx + logf(se);
        const int act = row_act;
        const float qa = q[act];
        float qnext = a.double_dqn ? qn[amax[lane]] : mxn;
        if (a.next_mask) {
          const float* mk = a.next_mask + (long)b * A;
          int best = 0;
          float bestv = qn[0] + (mk[0] - 1.f) * 1e10f;
          for (int j = 1; j < A; ++j) {
            const float v = qn[j] + (mk[j] - 1.f) * 1e10f;
            if (v > bestv) { bestv = v; best = j; }               // first maximum, like torch.argmax
          }
          qnext = qn[best];
        }
        const float y = row_rew + a.gamma * qnext * (1.f - row_done);
        const float diff = a.td_off ? 0.f : qa - y;
        float wgt = a.is_w ? a.is_w[b] : 1.f;
        if (a.w_uniform) wgt *= a.w_uniform[0];
        td = wgt * (diff * diff);
        pen = lse - a.log_A - qa;
        if (a.td_abs) a.td_abs[b] = fabsf(diff);
        const float ab = a.alpha * a.inv_batch;
        for (int j = 0; j < A; ++j) {
          float g = ab * expf(q[j] - lse);
          if (j == act) g += 2.f * a.inv_batch * wgt * diff - ab;
          dq[j] = g;
        }
        for (int j = A; j < ldq; ++j) dq[j] = 0.f;
      } else {
        for (int j = 0; j < ldq; ++j) dq[j] = 0.f;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { td += __shfl_xor(td, o); pen += __shfl_xor(pen, o); }
    if (lane == 0) { red[0] = td; red[1] = pen; }
  }
}

// ---- the same stage on 256 lanes: EIGHT lanes per row (round 3) --------------------------------------------------
// One lane per row spends ~5 k cycles in two expf per action (12 columns x 2 x ~35 instructions, one wave, seven
// other waves waiting at the barrier).  Here thread t of the first four waves takes row t / 8 and the actions
// sub, sub + 8, sub + 16, sub + 24 (sub = t % 8; A <= 32): one or two expf per lane, row-wide maximum / sum / picks as
// three-step butterflies over the 8 lanes on the DPP path (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror: no LDS
// traffic).  A butterfly leaves the SAME bits in all 8 lanes (each step adds the same two values in either order), so
// every lane can form the row's target / weight terms itself.  Against the one-lane form the row sum is associated as
// a tree instead of left to right: results agree to rounding (tests/test_cql_gpu.py compares the paths at 2e-6).
// Rows' scalars (action, reward, done) arrive in the 8-lane mapping (act8 / rew8 / done8, requested at kernel entry).
// The masked-argmax variant (BCQ) and A > 32 keep the one-lane form.
template <int CTRL> __device__ __forceinline__ float qf_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float qf_sum8(float x) {
  x += qf_dpp<0xB1>(x);
  x += qf_dpp<0x4E>(x);
  x += qf_dpp<0x141>(x);
  return x;
}
__device__ __forceinline__ float qf_max8(float x) {
  x = fmaxf(x, qf_dpp<0xB1>(x));
  x = fmaxf(x, qf_dpp<0x4E>(x));
  x = fmaxf(x, qf_dpp<0x141>(x));
  return x;
}
__device__ __forceinline__ bool qf_loss_wide_ok(const QnetFusedArgs& a) { return a.dims[a.n_lin] <= 32 && !a.next_mask; }

// t: thread number among the 256 that run the stage (four whole waves); red: >= 8 floats (per-wave partial sums)
__device__ __forceinline__ void qf_loss_rows8(const QnetFusedArgs& a, const float* Q, const float* Qn, float* dz, const int* amax,
                                              int act8, float rew8, float done8, int row0, int t, float* red, int rows) {
  const int A = a.dims[a.n_lin], ldq = qf_r32(A) + 4;          // 36
  const int r = t >> 3, sub = t & 7;
  const int b = row0 + r;
  const bool live = r < rows && b < a.B;
  const int rr = r < rows ? r : 0;
  const float* q = Q + rr * ldq;
  const float* qn = Qn + rr * ldq;
  float qv[4], nv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = sub + 8 * k;
    qv[k] = j < A ? q[j] : -INFINITY;
    nv[k] = j < A ? qn[j] : -INFINITY;
  }
  const float mx = qf_max8(fmaxf(fmaxf(qv[0], qv[1]), fmaxf(qv[2], qv[3])));
  const float mxn = qf_max8(fmaxf(fmaxf(nv[0], nv[1]), fmaxf(nv[2], nv[3])));
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) if (sub + 8 * k < A) se += expf(qv[k] - mx);
  se = qf_sum8(se);
  const float lse = mx + logf(se);
  float qa = 0.f, qpick = 0.f;
  const int am = a.double_dqn ? amax[rr] : -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = sub + 8 * k;
    qa += (j == act8 && j < A) ? qv[k] : 0.f;                   // exactly one lane of the row adds a non-zero term
    qpick += (j == am && j < A) ? nv[k] : 0.f;
  }
  qa = qf_sum8(qa);
  qpick = qf_sum8(qpick);
  const float qnext = a.double_dqn ? qpick : mxn;
  const float y = rew8 + a.gamma * qnext * (1.f - done8);
  const float diff = a.td_off ? 0.f : qa - y;
  float wgt = (a.is_w && live) ? a.is_w[b] : 1.f;
  if (a.w_uniform) wgt *= a.w_uniform[0];
  const float td = wgt * (diff * diff);
  const float pen = lse - a.log_A - qa;
  if (a.td_abs && live && sub == 0) a.td_abs[b] = fabsf(diff);
  const float ab = a.alpha * a.inv_batch;
  if (r < rows) {
    float* dq = dz + r * ldq;
#pragma unroll
    for (int k = 0; k < 5; ++k) {                               // the whole 36-float row: dL/dQ, then zeros
      const int j = sub + 8 * k;
      if (j < ldq) {
        float g = 0.f;
        if (k < 4 && j < A && live) {
          g = ab * expf(qv[k < 4 ? k : 0] - lse);
          if (j == act8) g += 2.f * a.inv_batch * wgt * diff - ab;
        }
        dq[j] = g;
      }
    }
  }
  float std_ = (live && sub == 0) ? td : 0.f, spen = (live && sub == 0) ? pen : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { std_ += __shfl_xor(std_, o); spen += __shfl_xor(spen, o); }
  if ((t & 63) == 0) { red[2 * (t >> 6)] = std_; red[2 * (t >> 6) + 1] = spen; }
}

// dW_l = dZ^T . in over the block's 32 rows (32 x 32 tiles of (n, k), dealt over `nw` waves, wave id `w`) and db_l =
// column sums of dZ (over `nt` threads, thread id `t`), written to the block's slab in the flat parameter layout.
// A tile's 32 + 32 fragment values are all requested before its 16 MFMAs (one LDS round trip per tile, not four).
__device__ __forceinline__ void qf_wgrad_n(const QnetFusedArgs& a, int l, const float* dz, const float* in, float* slab, int w,
                                           int nw, int li, int kh, int t, int nt) {
  const int N = a.dims[l + 1], K = a.dims[l];
  const int lddz = qf_r32(N) + 4, ldin = qf_r32(K) + 4;
  const int tiles_n = qf_r32(N) / 32, tiles_k = qf_r32(K) / 32;
  // (hoisted: left inside the predicated stores, a.w_off[l] was re-read from the kernel arguments — a scalar load and
  //  s_waitcnt lgkmcnt(0) — before each of a tile's 16 stores: ~8 k cycles per layer)
  float* const wslab = slab + a.w_off[l];
  float* const bslab = slab + a.b_off[l];
  const int ldp = qf_rk(K) + 4;
  for (int tile = w; tile < tiles_n * tiles_k; tile += nw) {
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    qf_f32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
    float av[QF_ROWS / 2], bv4[QF_ROWS / 2];
#pragma unroll
    for (int j = 0; j < QF_ROWS / 2; ++j) {
      const int row = 2 * j + kh;
      av[j] = dz[row * lddz + tn * 32 + li];
      bv4[j] = in[row * ldin + tk * 32 + li];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < QF_ROWS / 2; j += 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv4[j], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j + 1], bv4[j + 1], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
    const int k = tk * 32 + li;
    float* const wp = wslab + (tn * 32 + 4 * kh) * ldp + k;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dn = (r & 3) + 8 * (r >> 2);
      if (tn * 32 + 4 * kh + dn < N && k < K) wp[dn * ldp] = acc[r];
    }
  }
  for (int n = t; n < N; n += nt) {
    float v[QF_ROWS];
#pragma unroll
    for (int r = 0; r < QF_ROWS; ++r) v[r] = dz[r * lddz + n];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < QF_ROWS; ++r) s += v[r];
    bslab[n] = s;
  }
}

// dZ_{l-1} = (dZ . W_l) * 1[in > 0]: 32-column slabs of K over the waves.  dzp must not overlap `in`.
// The reduction runs over n in batches of 16: the half-wave kh takes n0 + 8 i + 4 kh + j (i < 2, j < 4), so its dZ
// values are two 16-byte reads of its row (the k index of v_mfma_f32_32x32x2_f32 only has to pair A with B), its
// weights eight 4-byte reads of consecutive columns.
__device__ __forceinline__ void qf_dgrad(const float* dz, int lddz, const float* wl, int N, int K, const float* in, int ldin,
                                         float* dzp, int wave, int li, int kh) {
  const int ldw = qf_rk(K) + 4, tiles_k = qf_r32(K) / 32, Np = qf_r32(N);
  for (int tk = wave; tk < tiles_k; tk += 4) {
    qf_f32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
    const float* ap = dz + li * lddz + 4 * kh;
    const float* bp = wl + 4 * kh * ldw + tk * 32 + li;
    // (the weight image has round32(N) rows and dZ rows are zero up to round32(N) columns: Np is a multiple of 32;
    //  unrolled per batch count like the forward tiles)
    for (int n0 = 0; n0 < Np; n0 += 64) {
      if (Np - n0 >= 64) qf_dgrad_batches<4>(ap + n0, bp + n0 * ldw, ldw, acc, acc1);
      else qf_dgrad_batches<2>(ap + n0, bp + n0 * ldw, ldw, acc, acc1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
    const int col = tk * 32 + li;
    // the 16 activations first, then the 16 stores: interleaved, every read waited for the store before it (the
    // compiler must assume dzp aliases in) — 16 LDS round trips, ~3 k cycles per layer
    float act[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) act[r] = in[((r & 3) + 8 * (r >> 2) + 4 * kh) * ldin + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      dzp[row * ldin + col] = (col < K && act[r] > 0.f) ? acc[r] : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void qnet_fused_kernel(const QnetFusedArgs a) {
  extern __shared__ float qf_lds[];
  __shared__ float red[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kh = lane >> 5;
  const int row0 = blockIdx.x * QF_ROWS;
  const int L = a.n_lin - 1;
  float* wl = qf_lds + a.lds_w;
  float* X = qf_lds + a.lds_act[0];
  const int ldx = qf_r32(a.dims[0]) + 4;

  // Weight pipeline: stage j = target layer j (j <= L), online layer j-L-1 (j <= 2L+1), then the online layers
  // L..1 again for the backward pass.  Stage j+1 is requested into registers right after stage j is parked in LDS.
  int n_stamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (a.stamps && blockIdx.x == 0 && t == 0) a.stamps[n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  float4 wr[QF_WREGS];
  const int n0 = a.double_dqn ? L + 1 : 0;          // stages of the leading online pass over s' (Double DQN)
  const int n_stages = n0 + 3 * L + 2;
  // (always_inline: a lambda left out of line takes `wr` by address, which moves the staging registers to scratch)
  auto stage = [&](int j, const float*& W, int& N, int& K) __attribute__((always_inline)) {
    const bool lead = j < n0;
    if (!lead) j -= n0;
    const int l = lead ? j : (j <= L ? j : (j <= 2 * L + 1 ? j - L - 1 : 3 * L + 2 - j));
    const float* P = (!lead && j <= L) ? a.params_tgt : a.params;
    W = P + a.w_off[l];                              // the layer's image: weights, then bias
    N = a.dims[l + 1]; K = a.dims[l];
  };
  auto fetch = [&](int j) __attribute__((always_inline)) {
    if (j >= n_stages) return;
    const float* W; int N, K;
    stage(j, W, N, K);
    qf_fetch_w(wr, W, N, K, t);
    // compiler barrier: without it the loads are sunk below the layer's MFMA loop (nothing there depends on them),
    // i.e. issued right before their first use, and every layer pays the full round trip to L2
    asm volatile("" ::: "memory");
  };
  auto park = [&](int j) __attribute__((always_inline)) {
    const float* W; int N, K;
    stage(j, W, N, K);
    qf_park_w(wl, wr, N, K, t);
  };
  fetch(0);
  // The loss stage's per-row scalars (action, reward, done flag) are requested NOW, by the lanes that will use them
  // (thread t < 32 = lane t of wave 0 = row t): they sit behind the same index indirection as the rows, and fetching
  // them only when the loss needs them exposed a full HBM round trip there.  Branch-free; they stay in registers.
  const int my_row = row0 + (t < QF_ROWS ? t : 0);
  const long my_src = my_row < a.B ? (a.idx ? a.idx[my_row] : (long)my_row) : 0L;
  const int row_act = (int)a.actions[my_src];
  const float row_rew = a.rew[my_src], row_done = a.done[my_src];
  // the same in the 8-lanes-per-row mapping of qf_loss_rows8 (thread t -> row t / 8)
  const int my_row8 = row0 + (t >> 3);
  const long my_src8 = my_row8 < a.B ? (a.idx ? a.idx[my_row8] : (long)my_row8) : 0L;
  const int act8 = (int)a.actions[my_src8];
  const float rew8 = a.rew[my_src8], done8 = a.done[my_src8];

  // ---- target network on s'  (cql_trainer.py:99-101) ---------------------------------------------------
  // s' waits in tmp[1] (free until the target net's layer 1 writes there), s in the online net's input buffer
  float* Xn = qf_lds + a.lds_tmp[1];
  qf_load_inputs(Xn, X, ldx, a.next_states, a.n_rs, a.states, a.s_rs, a.idx, row0, a.B, a.dims[0], t);
  __shared__ int amax[QF_ROWS];

  // Forward passes share ONE copy of the park / fetch / layer code (the kernel runs each instruction once per block,
  // so its size is its instruction-fetch cost): pass 0 = online net on s' (Double DQN only, argmax kept),
  // pass 1 = target net on s' (activations ping-pong in tmp), pass 2 = online net on s (activations kept).
  int stage_no = 0;
  for (int pass = a.double_dqn ? 0 : 1; pass < 3; ++pass) {
    for (int l = 0; l <= L; ++l) {
      park(stage_no);
      qf_barrier();
      fetch(stage_no + 1);
      ++stage_no;
      const float* in = l == 0 ? (pass == 2 ? X : Xn) : (pass == 1 ? qf_lds + a.lds_tmp[(l - 1) & 1] : qf_lds + a.lds_act[l]);
      float* out = pass == 1 ? qf_lds + a.lds_tmp[l & 1] : qf_lds + a.lds_act[l + 1];
      const float* bl = wl + qf_r32(a.dims[l + 1]) * (qf_rk(a.dims[l]) + 4);       // bias row of the parked image
      qf_forward(in, qf_r32(a.dims[l]) + 4, wl, a.dims[l], a.dims[l + 1], bl, l < L, out, qf_r32(a.dims[l + 1]) + 4, wave, li, kh);
      qf_barrier();
    }
    if (pass == 0 && t < QF_ROWS) {
      const float* q = qf_lds + a.lds_act[L + 1] + t * (qf_r32(a.dims[L + 1]) + 4);
      int best = 0;
      for (int j = 1; j < a.dims[L + 1]; ++j) best = q[j] > q[best] ? j : best;     // first maximum, like torch.max
      amax[t] = best;
    }
    if (pass == 1) stamp();
  }
  const float* Qn = qf_lds + a.lds_tmp[L & 1];
  float* dz = qf_lds + a.lds_tmp[(L + 1) & 1];

  stamp();
  // ---- loss and dL/dQ: one lane per row  (cql_trainer.py:94-118; same arithmetic as cql_loss_kernel) ----
  const bool wide = qf_loss_wide_ok(a);
  if (wide) qf_loss_rows8(a, qf_lds + a.lds_act[L + 1], Qn, dz, amax, act8, rew8, done8, row0, t, red, QF_ROWS);
  else qf_loss_rows(a, qf_lds + a.lds_act[L + 1], Qn, dz, amax, row_act, row_rew, row_done, row0, lane, wave, red);
  qf_barrier();
  if (t == 0) {
    a.part_td[blockIdx.x] = wide ? ((red[0] + red[2]) + red[4]) + red[6] : red[0];
    a.part_pen[blockIdx.x] = wide ? ((red[1] + red[3]) + red[5]) + red[7] : red[1];
  }
  stamp();

  // ---- backward, top down ----------------------------------------------------------------------------------
  float* slab = a.slab + (long)blockIdx.x * a.slab_stride;
  for (int l = L; l >= 0; --l) {
    const int N = a.dims[l + 1], K = a.dims[l];
    const int lddz = qf_r32(N) + 4, ldin = qf_r32(K) + 4;
    const float* in = qf_lds + a.lds_act[l];
    if (l > 0) park(n0 + 3 * L + 2 - l);                                   // for dZ_{l-1}; nobody reads wl right now
    qf_wgrad_n(a, l, dz, in, slab, wave, 4, li, kh, t, 256);
    stamp();
    if (l == 0) break;
    qf_barrier();                                                  // wl is parked
    fetch(n0 + 3 * L + 3 - l);
    // dZ_{l-1} = (dZ . W_l) * 1[in > 0]: 32-column slabs of K over the waves
    float* dzp = qf_lds + a.lds_tmp[l & 1];                           // dZ_l lives in tmp[(l + 1) & 1]
    qf_dgrad(dz, lddz, wl, N, K, in, ldin, dzp, wave, li, kh);
    qf_barrier();
    dz = dzp;
    stamp();
  }
}

// ---- two wave groups per block -----------------------------------------------------------------------------------
// The one-group kernel above is a chain of ~3L+3 dependent stages (park a layer, barrier, 32-row MFMA chains, barrier),
// each worth 5-9 k cycles of latency whatever the arithmetic.  Two of its chains are independent of each other:
//   * the target network on s' and the online network on s (cql_trainer.py:94 and :99-101), and
//   * dW_l = dZ_l^T . in_l and dZ_{l-1} = (dZ_l . W_l) * 1[in_l > 0] of one layer.
// This kernel runs them side by side on 8 waves: group 0 (waves 0-3) walks the target net, then the dgrad chain with
// the weight staging that chain needs; group 1 (waves 4-7) walks the online net, then the wgrad tiles.  Both groups
// execute the same barrier sequence (the layer shapes are the same), so the shared s_barrier costs nothing extra.
// In the backward pass nobody reads forward weights any more, so the two weight buffers alternate: W_{l-1} is parked
// while layer l is differentiated, and a layer costs ONE barrier.  Same arithmetic, same summation orders: results are
// bit-identical to the one-group kernel (tests/test_cql_gpu.py).  Needs a second weight image in LDS (config 3:
// 151 KB); the host falls back to the one-group kernel when that does not fit.
// One input (32 rows) of a wave group, in three steps so that the caller can put independent work between the two
// dependent round trips: (1) the source row numbers, (2) the elements, (3) the LDS stores.
// Every slot is straight-line code: a slot past the image re-reads element 0 and is dropped at the store.  (With a
// branch per slot — `if (u * 256 < total)`, `idx ? idx[b] : b` — the compiler closed every slot with s_waitcnt
// vmcnt(0): nine dependent round trips to HBM at kernel entry instead of two, ~12 k cycles.)
struct QfInput { long row[QF_XREGS]; float v[QF_XREGS]; };
// NU: staging slots per thread that can be live for the block's row count (ceil(rows * (QF_MAX_W + 4) / 256): 17 for 32
// rows, 9 for 16 — the 16-row kernel spent 2.9 k cycles of its entry on the 17-slot form's row-number look-ups)
template <int NU = QF_XREGS>
__device__ __forceinline__ void qf_input_rows(QfInput& in, int ld, const int64_t* idx, int row0, int B, int t, int rows = QF_ROWS,
                                              const long* lds_rows = nullptr) {
  const int total = rows * ld;
  const float inv_ld = 1.0f / (float)ld;
  if (lds_rows) {                                     // sampled in the kernel: the block's row numbers sit in LDS
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = u * 256 + t;
      in.row[u] = lds_rows[qf_div(i < total ? i : 0, inv_ld)];
    }
  } else if (idx) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = u * 256 + t;
      const int b = row0 + qf_div(i < total ? i : 0, inv_ld);
      in.row[u] = idx[b < B ? b : 0];
    }
  } else {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = u * 256 + t;
      in.row[u] = row0 + qf_div(i < total ? i : 0, inv_ld);
    }
  }
}
template <int NU = QF_XREGS>
__device__ __forceinline__ void qf_input_load(QfInput& in, int ld, const float* src, long rs, int row0, int B, int cols, int t,
                                              int rows = QF_ROWS) {
  const int total = rows * ld;
  const float inv_ld = 1.0f / (float)ld;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = u * 256 + t;
    const int r = qf_div(i < total ? i : 0, inv_ld), c = i - r * ld;
    const bool ok = i < total && row0 + r < B && c < cols;
    const float x = src[ok ? in.row[u] * rs + c : 0L];
    in.v[u] = ok ? x : 0.f;
  }
}
template <int NU = QF_XREGS>
__device__ __forceinline__ void qf_input_store(const QfInput& in, float* dst, int ld, int t, int rows = QF_ROWS) {
  const int total = rows * ld;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = u * 256 + t;
    if (i < total) dst[i] = in.v[u];
  }
}

// ---- 16 minibatch rows per block on v_mfma_f32_16x16x4_f32 (the default since round 3 while 32-row blocks would leave
// ---- CUs idle; porl_tune_set("qnet_rows16", 0) = the 32-row kernel) -----------------------------------------------------
// Config 3 at B = 4096 is 128 blocks of 32 rows: half of the chip idles, and inside a block the stage time is the
// dependent chain of 64-cycle 32x32x2 MFMAs of one or two column slabs.  With 16 rows per block the same batch is 256
// blocks, and a layer is 16 x 16 tiles (32-cycle MFMAs over 4 reduction indices): a 64-wide layer is four tiles = one
// per wave, its chain K/4 x 32 cycles — a quarter of the 32-row kernel's.  MEASURED: no faster (43.2 vs 42.4 us; the
// dZ chain 9.6 k -> 3.9 k cycles per layer, but a forward tile of 16 MFMAs takes 2.3-3.7 k cycles where the bare
// instruction stream (scripts/mfma_rate.hip: 32.2 cycles per MFMA, 86 with LDS operands at two waves per SIMD) needs
// 1.4 k, and the 256 partial-gradient slabs cost the reduce launch +1.8 us) — in ROUND 2.  Round 3: with the loss
// stage on eight lanes per row, the dW tiles shared between the wave groups and the loss-stage weight request moved to
// the idle group, this form runs 26 400 updates/s against 22 800 (step kernel ~31 against 38.8 us by rocprofv3).  The stage
// time of these kernels is not the matrix pipe's.  Lane l = (l16 = l & 15, kq = l >> 4) supplies
// A[row l16][k] and B[k][col l16] for k = 4 kq + j in MFMA j of a batch of 16 reduction indices (any k order is legal as
// long as A and B agree), so each operand is ONE 16-byte read per batch; D register i is row 4 kq + i, column l16.
typedef float qf_f32x4 __attribute__((ext_vector_type(4)));

template <int NB>
__device__ __forceinline__ void qf16_kk_batches(const float* ap, const float* bp, qf_f32x4& acc, qf_f32x4& acc1) {
  float4 av[NB], bv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    av[b] = *reinterpret_cast<const float4*>(ap + 16 * b);
    bv[b] = *reinterpret_cast<const float4*>(bp + 16 * b);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].x, bv[b].x, acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].y, bv[b].y, acc1, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].z, bv[b].z, acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].w, bv[b].w, acc1, 0, 0, 0);
  }
}

// out[16][ldo] = act(in[16][ldi] . Wl^T + bias): 16-column tiles w, w+4, ... per wave (all round32(N) columns are
// written: the zero rows of the weight image give the zero padding the next layer's reduction reads)
__device__ __forceinline__ void qf16_forward(const float* in, int ldi, const float* wl, int K, int N, const float* bias, bool relu,
                                             float* out, int ldo, int wave, int l16, int kq) {
  const int ldw = qf_rk(K) + 4, nb = qf_rk(K) >> 4;
  for (int tn = wave; tn < qf_r32(N) / 16; tn += 4) {
    qf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* ap = in + l16 * ldi + 4 * kq;
    const float* bp = wl + (tn * 16 + l16) * ldw + 4 * kq;
    switch (nb) {
      case 1: qf16_kk_batches<1>(ap, bp, acc, acc1); break;
      case 2: qf16_kk_batches<2>(ap, bp, acc, acc1); break;
      case 3: qf16_kk_batches<3>(ap, bp, acc, acc1); break;
      case 4: qf16_kk_batches<4>(ap, bp, acc, acc1); break;
      case 5: qf16_kk_batches<5>(ap, bp, acc, acc1); break;
      case 6: qf16_kk_batches<6>(ap, bp, acc, acc1); break;
      case 7: qf16_kk_batches<7>(ap, bp, acc, acc1); break;
      default: qf16_kk_batches<8>(ap, bp, acc, acc1); break;
    }
    const int col = tn * 16 + l16;
    const float bv = bias[col];                    // zero past N
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[i] + acc1[i] + bv;
      if (relu) v = fmaxf(v, 0.f);
      if (col >= N) v = 0.f;
      out[(4 * kq + i) * ldo + col] = v;
    }
  }
}

// dZ_{l-1} = (dZ . W_l) * 1[in > 0] for 16 rows: 16-column tiles of K over the waves, reduction over n in batches of 16
// (lane quarter kq takes n0 + 4 kq + j: one 16-byte read of its dZ row, four 4-byte reads of consecutive weight columns)
template <int NB>
__device__ __forceinline__ void qf16_nn_batches(const float* ap, const float* bp, int ldw, qf_f32x4& acc, qf_f32x4& acc1) {
  float4 av[NB];
  float bv[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    av[b] = *reinterpret_cast<const float4*>(ap + 16 * b);
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[b][j] = bp[(16 * b + j) * ldw];
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].x, bv[b][0], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].y, bv[b][1], acc1, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].z, bv[b][2], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].w, bv[b][3], acc1, 0, 0, 0);
  }
}
__device__ __forceinline__ void qf16_dgrad(const float* dz, int lddz, const float* wl, int N, int K, const float* in, int ldin,
                                           float* dzp, int wave, int l16, int kq) {
  const int ldw = qf_rk(K) + 4, Np = qf_r32(N);
  for (int tk = wave; tk < qf_r32(K) / 16; tk += 4) {
    qf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* ap = dz + l16 * lddz + 4 * kq;
    const float* bp = wl + 4 * kq * ldw + tk * 16 + l16;
    for (int n0 = 0; n0 < Np; n0 += 64) {          // Np is a multiple of 32
      if (Np - n0 >= 64) qf16_nn_batches<4>(ap + n0, bp + n0 * ldw, ldw, acc, acc1);
      else qf16_nn_batches<2>(ap + n0, bp + n0 * ldw, ldw, acc, acc1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int col = tk * 16 + l16;
    float act[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) act[i] = in[(4 * kq + i) * ldin + col];
#pragma unroll
    for (int i = 0; i < 4; ++i) dzp[(4 * kq + i) * ldin + col] = (col < K && act[i] > 0.f) ? acc[i] + acc1[i] : 0.f;
  }
}

// dW_l = dZ^T . in over the block's 16 rows: 16 x 16 tiles of (n, k), four MFMAs each, dealt over `nw` waves; db_l = column
// sums of dZ over `nt` threads.  Tiles that lie entirely in the padding are skipped.
// (tile_lo, tile_hi): the share of the layer's tiles this call covers, as sixteenths of the tile count — the dZ-chain
// group takes the upper share of the 32-tile layers after its own product (qnet_fused2_kernel), nt = 0 skips db
__device__ __forceinline__ void qf16_wgrad(const QnetFusedArgs& a, int l, const float* dz, const float* in, float* slab, int w,
                                           int nw, int l16, int kq, int t, int nt, int share_lo = 0, int share_hi = 16) {
  const int N = a.dims[l + 1], K = a.dims[l];
  const int lddz = qf_r32(N) + 4, ldin = qf_r32(K) + 4;
  const int tiles_n = (N + 15) / 16, tiles_k = (K + 15) / 16;
  float* const wslab = slab + a.w_off[l];
  float* const bslab = slab + a.b_off[l];
  const int ldp = qf_rk(K) + 4;
  const int T = tiles_n * tiles_k;
  for (int tile = T * share_lo / 16 + w; tile < T * share_hi / 16; tile += nw) {
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    float av[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      av[j] = dz[(4 * j + kq) * lddz + tn * 16 + l16];
      bv[j] = in[(4 * j + kq) * ldin + tk * 16 + l16];
    }
    qf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
    const int k = tk * 16 + l16;
    float* const wp = wslab + (tn * 16 + 4 * kq) * ldp + k;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (tn * 16 + 4 * kq + i < N && k < K) wp[i * ldp] = acc[i] + acc1[i];
  }
  if (nt > 0)
  for (int n = t; n < N; n += nt) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = dz[r * lddz + n];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
    bslab[n] = s;
  }
}

template <int ROWS>
__global__ __launch_bounds__(512) void qnet_fused2_kernel(const QnetFusedArgs a, int lds_w2) {
  extern __shared__ float qf_lds[];
  __shared__ float red[8];
  __shared__ int amax[QF_ROWS];
  const int t = threadIdx.x, grp = t >> 8, tg = t & 255, lane = t & 63, li = lane & 31, kh = lane >> 5;
  // A block's waves are dealt to the SIMDs in the order 0,2,1,3,0,2,1,3: wave w of group 0 and wave w of group 1 share
  // a SIMD.  Layers with fewer than four 32-column slabs occupy waves 0.. only, so group 1 numbers its waves from 2:
  // the busy waves of the two groups then sit on different SIMDs.
  const int wave = (((t >> 6) & 3) + 2 * grp) & 3;
  const int l16 = lane & 15, kq = lane >> 4;          // lane coordinates of the 16 x 16 x 4 tiles (ROWS == 16)
  const int row0 = blockIdx.x * ROWS;
  const int L = a.n_lin - 1;
  float* const wbuf0 = qf_lds + a.lds_w;
  float* const wbuf1 = qf_lds + lds_w2;
  float* wl = grp ? wbuf1 : wbuf0;                   // the group's forward weight buffer
  float* X = qf_lds + a.lds_act[0];
  float* Xn = qf_lds + a.lds_tmp[1];
  const int ldx = qf_r32(a.dims[0]) + 4;
  const int dd = a.double_dqn;

  int n_stamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (a.stamps && blockIdx.x == 0 && t == 0) a.stamps[n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  int n_stamp2 = 0;                                  // group 1's own timeline (its wave 0), entries 32..
  auto stamp2 = [&]() __attribute__((always_inline)) {
    if (a.stamps && blockIdx.x == 0 && t == 256) a.stamps[32 + n_stamp2++] = __builtin_amdgcn_s_memtime();
  };
  // Weight staging by LDS-DMA (global_load_lds_dwordx4): the parameter group stores every layer as its LDS image, so a
  // layer is a linear copy of 1 KiB pieces (wave-uniform LDS base + lane x 16 B) that needs no registers and no store
  // pass.  (Register staging — 17 float4 per thread fetched a stage ahead, parked with ds_write_b128 — cost ~2.9 k
  // cycles per layer to park, ~1.7 k to issue and another ~2 k of barrier skew: scripts/bench_cql_prof.py.)
  // Weight stages of a group, in the order it uses them:
  //   group 0: target layers 0..L, then the online layers L..1 (dgrad)
  //   group 1: [Double DQN: online layers 0..L for the pass over s',] online layers 0..L
  const int n_lead = dd ? L + 1 : 0;
  // (as_grp: whose stage list j refers to — the loss stage has the idle group issue the other group's next image)
  auto dma = [&](int j, float* dst, int as_grp = -1) __attribute__((always_inline)) {
    int l;
    const float* P = a.params;
    const int g_ = as_grp < 0 ? grp : as_grp;
    if (g_) l = j < n_lead ? j : j - n_lead;
    else if (j <= L) { l = j; P = a.params_tgt; }
    else l = 2 * L + 1 - j;
    const float* img = P + a.w_off[l];
    const int total4 = qf_image4(a.dims[l + 1], a.dims[l]);
    // Written as an asm statement: the compiler orders every later LDS read behind a pending
    // __builtin_amdgcn_global_load_lds with s_waitcnt vmcnt(0) (it cannot tell the buffers apart), which would expose a
    // piece's whole flight time in the dgrad chain and in the loss stage.  The waits are placed by hand (barrier_vm).
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)dst;
    for (int base = (tg >> 6) * 64; base < total4; base += 256) {            // wave-uniform
      const int i = base + lane;
      if (i < total4) {
        unsigned keep;
        const float* gsrc = img + 4 * i;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds0 + 16u * (unsigned)base);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
      }
    }
  };
  // barrier that also retires this wave's LDS-DMA pieces (they count in vmcnt)
  auto barrier_vm = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // The longest dependent chain at entry is index -> row -> LDS: the row numbers are requested first, the first weight
  // image and the loss stage's per-row scalars (lanes 0..31 of wave 0; see the one-group kernel) while they are in flight.
  // Group 0 gathers s' (for the target net), group 1 gathers s.
  QfInput xin;
  __shared__ long srow[QF_ROWS];
  const bool sampled = a.samp_n > 0;
  if (sampled) {
    if (t < ROWS) {
      const int b = row0 + t;
      srow[t] = b < a.B ? (long)feistel_index(a.samp_n, b, a.samp_seed, a.samp_step, a.samp_hb) : 0L;
    }
    dma(0, wl);                                      // (in flight across the barrier: it is an asm statement)
    qf_barrier();
  }
  constexpr int NU = (ROWS * (QF_MAX_W + 4) + 255) / 256;
  qf_input_rows<NU>(xin, ldx, a.idx, row0, a.B, tg, ROWS, sampled ? srow : nullptr);
  const int my_row = row0 + (t < ROWS ? t : 0);
  const long my_src = sampled ? srow[t < ROWS ? t : 0] : (my_row < a.B ? (a.idx ? a.idx[my_row] : (long)my_row) : 0L);
  if (!sampled) dma(0, wl);
  qf_input_load<NU>(xin, ldx, grp ? a.states : a.next_states, grp ? a.s_rs : a.n_rs, row0, a.B, a.dims[0], tg, ROWS);
  const int row_act = (int)a.actions[my_src];
  const float row_rew = a.rew[my_src], row_done = a.done[my_src];
  // the same in the 8-lanes-per-row mapping of qf_loss_rows8 (thread tg of group 0 -> row tg / 8)
  const int r8 = (tg >> 3) < ROWS ? (tg >> 3) : 0;
  const long my_src8 = sampled ? srow[r8] : (row0 + r8 < a.B ? (a.idx ? a.idx[row0 + r8] : (long)(row0 + r8)) : 0L);
  const int act8 = (int)a.actions[my_src8];
  const float rew8 = a.rew[my_src8], done8 = a.done[my_src8];
  qf_input_store<NU>(xin, grp ? X : Xn, ldx, tg, ROWS);
  stamp();

  // ---- forward: role 1 = target net on s' (ping-pong in tmp), 2 = online net on s' (Double DQN: argmax kept),
  //      3 = online net on s (activations kept), 0 = idle (barriers only) -----------------------------------------------
  int stage_no = 0;
  for (int phase = dd ? 0 : 1; phase < 2; ++phase) {
    const int role = phase == 0 ? (grp ? 2 : 1) : (grp ? 3 : (dd ? 0 : 1));
    for (int l = 0; l <= L; ++l) {
      if (role && stage_no > 0) dma(stage_no, wl);       // (stage 0 was requested at entry; wl is free: barrier below)
      stamp();
      barrier_vm();
      stamp();
      if (role) {
        ++stage_no;
        const float* in = l == 0 ? (role == 3 ? X : Xn) : (role == 1 ? qf_lds + a.lds_tmp[(l - 1) & 1] : qf_lds + a.lds_act[l]);
        float* out = role == 1 ? qf_lds + a.lds_tmp[l & 1] : qf_lds + a.lds_act[l + 1];
        const float* bl = wl + qf_r32(a.dims[l + 1]) * (qf_rk(a.dims[l]) + 4);
        if constexpr (ROWS == 16)
          qf16_forward(in, qf_r32(a.dims[l]) + 4, wl, a.dims[l], a.dims[l + 1], bl, l < L, out, qf_r32(a.dims[l + 1]) + 4, wave, l16, kq);
        else
          qf_forward(in, qf_r32(a.dims[l]) + 4, wl, a.dims[l], a.dims[l + 1], bl, l < L, out, qf_r32(a.dims[l + 1]) + 4, wave, li, kh);
        stamp();
      }
      qf_barrier();
      stamp();
    }
    if (role == 2 && tg < ROWS) {
      const float* q = qf_lds + a.lds_act[L + 1] + tg * (qf_r32(a.dims[L + 1]) + 4);
      int best = 0;
      for (int j = 1; j < a.dims[L + 1]; ++j) best = q[j] > q[best] ? j : best;     // first maximum, like torch.max
      amax[tg] = best;
    }
  }
  const float* Qn = qf_lds + a.lds_tmp[L & 1];
  float* dz = qf_lds + a.lds_tmp[(L + 1) & 1];

  // ---- loss (wave 0 of group 0) ---------------------------------------------------------------------------------------
  // The dgrad chain needs the ONLINE weights W_L .. W_1.  W_L is what group 1 staged last, so it is already in wbuf1;
  // the two buffers alternate from there (bsel), and W_{L-1} is requested into wbuf0 now — every wave is past the last
  // forward barrier, so group 0's forward buffer is dead — and lands while the loss is computed.
  auto bsel = [&](int l) __attribute__((always_inline)) { return ((L - l) & 1) ? wbuf0 : wbuf1; };
  // (issued by group 1, which has nothing else to do during the loss: a group's four waves need ~2 k cycles to issue an
  //  image, and group 0 used to spend them in front of the loss arithmetic)
  if (grp == 1 && L > 1) dma(L + 2, bsel(L - 1), 0);
  const bool wide = qf_loss_wide_ok(a);
  if (grp == 0) {
    if (wide) qf_loss_rows8(a, qf_lds + a.lds_act[L + 1], Qn, dz, amax, act8, rew8, done8, row0, tg, red, ROWS);
    else qf_loss_rows(a, qf_lds + a.lds_act[L + 1], Qn, dz, amax, row_act, row_rew, row_done, row0, lane, wave, red, ROWS);
  }
  barrier_vm();                                    // group 1's image has landed, group 0's loss rows are written
  if (t == 0) {
    a.part_td[blockIdx.x] = wide ? ((red[0] + red[2]) + red[4]) + red[6] : red[0];
    a.part_pen[blockIdx.x] = wide ? ((red[1] + red[3]) + red[5]) + red[7] : red[1];
  }
  stamp();

  // ---- backward, top down: group 0 = dZ chain (+ requests the weights two layers down), group 1 = dW / db ---------
  // While layer l is differentiated from bsel(l), W_{l-1} is on its way into the other buffer: requested at the top of
  // this iteration (that buffer held W_{l+1}, which every wave is done with since the last barrier), waited for at the
  // closing barrier.
  // Only group 0 waits on vmcnt at the closing barrier: group 1's counter holds its slab stores, which nobody reads here.
  float* slab = a.slab + (long)blockIdx.x * a.slab_stride;
  for (int l = L; l >= 1; --l) {
    const int N = a.dims[l + 1], K = a.dims[l];
    const int lddz = qf_r32(N) + 4, ldin = qf_r32(K) + 4;
    const float* in = qf_lds + a.lds_act[l];
    float* dzp = qf_lds + a.lds_tmp[l & 1];                           // dZ_l lives in tmp[(l + 1) & 1]
    const int wshare = (ROWS == 16 && a.wgrad_share > 0 && ((N + 15) / 16) * ((K + 15) / 16) >= 16) ? a.wgrad_share : 16;
    if (grp == 0) {
      if (l < L && l > 1) dma(2 * L + 2 - l, bsel(l - 1));            // W_{l-1} into the buffer dgrad(l + 1) is done with
      stamp();
      if constexpr (ROWS == 16) qf16_dgrad(dz, lddz, bsel(l), N, K, in, ldin, dzp, wave, l16, kq);
      else qf_dgrad(dz, lddz, bsel(l), N, K, in, ldin, dzp, wave, li, kh);
      // 16-row blocks: the dW tiles of a 128-wide layer keep group 1 busy for ~8.7 k cycles while this group's product
      // takes ~3 k (stamps: scripts/bench_cql_prof.py) — it takes the upper 5/16 of the tiles behind its product
      if constexpr (ROWS == 16) {
        if (wshare < 16) qf16_wgrad(a, l, dz, in, slab, wave, 4, l16, kq, 0, 0, wshare, 16);
      }
      stamp();
      barrier_vm();
    } else {
      stamp2();
      if constexpr (ROWS == 16) qf16_wgrad(a, l, dz, in, slab, wave, 4, l16, kq, tg, 256, 0, wshare);
      else qf_wgrad_n(a, l, dz, in, slab, wave, 4, li, kh, tg, 256);
      stamp2();
      qf_barrier();
    }
    dz = dzp;
    stamp();
  }
  // layer 0 has no dZ to pass on: all eight waves share its dW tiles
  if constexpr (ROWS == 16) qf16_wgrad(a, 0, dz, qf_lds + a.lds_act[0], slab, (t >> 6), 8, l16, kq, t, 512);
  else qf_wgrad_n(a, 0, dz, qf_lds + a.lds_act[0], slab, (t >> 6), 8, li, kh, t, 512);
  stamp();
}

// optional Adam step fused into the gradient reduction (p == null: off)
struct QnetAdam { float* p; float* m; float* v; float omb1, beta2, omb2, eps, step_size, bc2_sqrt; };

// grads[i] = sum over blocks of slab[b][i], in a fixed order: 32 parameters x 8 slab lanes per block, lane j adds
// slabs j, j+8, ... and the 8 lane sums are added in lane order.  Block 0 also folds the loss partials into
// stats[0] = loss, [1] = td, [2] = penalty (this rank's shares, like cql_finalize_kernel).
__global__ __launch_bounds__(256) void qnet_reduce_kernel(const float* __restrict__ slab, long stride, int nblocks, long n,
                                                          float* __restrict__ grads, const float* __restrict__ part_td,
                                                          const float* __restrict__ part_pen, float inv_batch, float alpha,
                                                          float* __restrict__ stats, QnetAdam ad) {
  __shared__ float part[8][32];
  const int c = threadIdx.x & 31, j = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + c;
  float s = 0.f;
  if (i < n) {
    for (int b0 = j; b0 < nblocks; b0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = b0 + 8 * u < nblocks ? slab[(long)(b0 + 8 * u) * stride + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  part[j][c] = s;
  __syncthreads();
  if (j == 0 && i < n) {
    float r = part[0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) r += part[k][c];
    grads[i] = r;
    if (ad.p) {       // torch.optim.Adam, the arithmetic of adam_ema_kernel (kernels.hpp)
      float mm = ad.m[i], vv = ad.v[i], pp = ad.p[i];
      mm = mm + ad.omb1 * (r - mm);
      vv = vv * ad.beta2 + ad.omb2 * r * r;
      pp = pp - ad.step_size * (mm / (sqrtf(vv) / ad.bc2_sqrt + ad.eps));
      ad.m[i] = mm; ad.v[i] = vv; ad.p[i] = pp;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float td = 0.f, pen = 0.f;
    for (int k = threadIdx.x; k < nblocks; k += 64) { td += part_td[k]; pen += part_pen[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { td += __shfl_xor(td, o); pen += __shfl_xor(pen, o); }
    if (threadIdx.x == 0) {
      td *= inv_batch; pen *= inv_batch;
      stats[0] = td + alpha * pen; stats[1] = td; stats[2] = pen;
    }
  }
}

}  // namespace porl
