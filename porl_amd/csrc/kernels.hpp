// Non-GEMM kernels of the update step: batch packing, loss heads, reductions, Adam(+EMA) sweep, replay
// gather.  All HBM/L2-bound or latency-bound; written for 64-wide wavefronts (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace porl {

constexpr float LOG_STD_MIN = -5.0f;   // reference agent/policy.py:8-9
constexpr float LOG_STD_MAX = 2.0f;
constexpr float EXP_ADV_MAX = 100.0f;  // reference agent/por.py:12

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}

// ---------------------------------------------------------------------------------------------------
// pack: strided caller tensors -> dense, zero-padded, 16-byte-aligned step buffers.
// The reference hands the update column slices of one (B,row) tensor (por_train.py:74-78); s' starts
// at float offset S+1, i.e. is only 4-byte aligned, so it cannot feed 16-byte loads directly.
// ---------------------------------------------------------------------------------------------------
struct PackJob {
  const float* src; float* dst;
  long src_row_stride;   // floats between rows
  long src_col_stride;   // floats between columns (1 for matrices, unused for vectors)
  int cols;              // valid columns
  int ld;                // dst leading dimension (>= cols, padded with zeros)
};
struct PackArgs { int njobs; int rows; PackJob job[6]; };

__global__ void pack_kernel(const PackArgs a) {
  const PackJob& j = a.job[blockIdx.y];
  const long n = (long)a.rows * j.ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / j.ld;
    const int c = (int)(i - r * j.ld);
    j.dst[i] = c < j.cols ? j.src[r * j.src_row_stride + (long)c * j.src_col_stride] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------
// IQL value head: v = sum(parts) + b; TD target; expectile loss and dL/dv   (agent/por.py:81-87)
//   headparts layout: [net][part][B], nets = {target1, target2, online1, online2}
// One block; deterministic reduction order.
// ---------------------------------------------------------------------------------------------------
struct ValueLossArgs {
  const float* hp_t[2]; const float* hp_v[2];
  const float* b_t[2]; const float* b_v[2];     // scalar output biases (device pointers)
  const float* rew; const float* term;
  float* target_v; float* dv[2];
  float* db_out[2];    // gradient of the scalar output bias: sum_b dv[b]
  float* stats;        // stats[0] = v_loss (this rank's share)
  int B, parts;
  float tau, discount, inv_batch;
};

__global__ __launch_bounds__(1024) void value_loss_kernel(const ValueLossArgs a) {
  __shared__ float red[3][16];
  float lsum = 0.f, d0sum = 0.f, d1sum = 0.f;
  const float bt0 = a.b_t[0][0], bt1 = a.b_t[1][0], bv0 = a.b_v[0][0], bv1 = a.b_v[1][0];
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    float t0 = 0.f, t1 = 0.f, v0 = 0.f, v1 = 0.f;
#pragma unroll 8
    for (int p = 0; p < a.parts; ++p) {
      const size_t o = (size_t)p * a.B + b;
      t0 += a.hp_t[0][o]; t1 += a.hp_t[1][o]; v0 += a.hp_v[0][o]; v1 += a.hp_v[1][o];
    }
    t0 += bt0; t1 += bt1; v0 += bv0; v1 += bv1;
    const float next_v = fminf(t0, t1);
    const float tgt = a.rew[b] + (1.f - a.term[b]) * a.discount * next_v;
    a.target_v[b] = tgt;
    const float u0 = tgt - v0, u1 = tgt - v1;
    const float w0 = fabsf(a.tau - (u0 < 0.f ? 1.f : 0.f)), w1 = fabsf(a.tau - (u1 < 0.f ? 1.f : 0.f));
    lsum += 0.5f * (w0 * u0 * u0 + w1 * u1 * u1);
    const float d0 = -w0 * u0 * a.inv_batch, d1 = -w1 * u1 * a.inv_batch;
    a.dv[0][b] = d0;
    a.dv[1][b] = d1;
    d0sum += d0; d1sum += d1;
  }
  lsum = wave_sum(lsum); d0sum = wave_sum(d0sum); d1sum = wave_sum(d1sum);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = lsum; red[1][wave] = d0sum; red[2][wave] = d1sum; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[threadIdx.x][w];
    if (threadIdx.x == 0) a.stats[0] = s * a.inv_batch;
    else a.db_out[threadIdx.x - 1][0] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// advantage weight + diagonal-Gaussian NLL + its gradient   (agent/por.py:97-106, policy.py:18-23)
//   one wave per row, lane j handles columns j, j+64, ...
// ---------------------------------------------------------------------------------------------------
struct PolicyNllArgs {
  const float* hp_v[2]; const float* b_v[2]; int parts;   // updated twin V heads
  const float* target_v;
  const float* mean_slab; int nslab; long slab_stride;     // pre-bias mean = sum of split-K slabs
  const float* mean_bias; const float* log_std;
  const float* x; int ldx;                                 // regression target (s' or actions)
  float* dmean; int ldd;                                   // dL/d(pre-activation mean), (B, ldd)
  float* part_loss; float* part_min; float* part_dls;      // per-block partials; part_dls[blk][D]
  int B, D, ldm;
  int tanh_mean; int weight_mode;                          // 0: exp(adv/alpha), 1: exp(alpha*adv)
  float alpha, inv_batch;
  int rows_per_block;
};

constexpr int NLL_MAX_COLS_PER_LANE = 8;   // D <= 512
constexpr int NLL_FAST_SLABS = 16;

__global__ __launch_bounds__(256) void policy_nll_kernel(const PolicyNllArgs a) {
  __shared__ float sh_dls[4][NLL_MAX_COLS_PER_LANE * 64];
  __shared__ float sh_loss[4], sh_min[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ncl = (a.D + 63) >> 6;
  float sig[NLL_MAX_COLS_PER_LANE], isg[NLL_MAX_COLS_PER_LANE], mb[NLL_MAX_COLS_PER_LANE];
  float dls[NLL_MAX_COLS_PER_LANE];
  float lsig = 0.f;
#pragma unroll
  for (int c = 0; c < NLL_MAX_COLS_PER_LANE; ++c) {
    const int j = lane + 64 * c;
    sig[c] = 1.f; isg[c] = 1.f; mb[c] = 0.f; dls[c] = 0.f;
    if (c < ncl && j < a.D) {
      const float ls = fminf(fmaxf(a.log_std[j], LOG_STD_MIN), LOG_STD_MAX);
      sig[c] = expf(ls);
      isg[c] = 1.f / sig[c];
      lsig += logf(sig[c]);               // the reference takes log(exp(.))
      mb[c] = a.mean_bias[j];
    }
  }
  lsig = wave_sum(lsig);
  const float half_log2pi_D = 0.5f * (float)((double)a.D * 1.8378770664093454836);   // D*ln(2*pi)/2
  const float bv0 = a.b_v[0][0], bv1 = a.b_v[1][0];
  float loss_acc = 0.f, min_acc = INFINITY;
  const int row0 = blockIdx.x * a.rows_per_block;
  const int row1 = min(a.B, row0 + a.rows_per_block);
  // Common shape (D <= 64, at most NLL_FAST_SLABS mean slabs, at most 64 head partials): every load of a row is issued
  // before the first use — head partials, TD target, all slabs, the regression target: ONE round trip per row instead
  // of a chain of three (the kernel is one wave per row and nothing else hides the latency).  Same sums in the same
  // order as the general loop below.
  const bool fast = ncl == 1 && a.nslab <= NLL_FAST_SLABS && a.parts <= 64;
  for (int b = row0 + wave; fast && b < row1; b += 4) {
    const bool pv = lane < a.parts, cv = lane < a.D;
    const float h0 = a.hp_v[0][pv ? (size_t)lane * a.B + b : 0], h1 = a.hp_v[1][pv ? (size_t)lane * a.B + b : 0];
    const float tv = a.target_v[b];
    float ms[NLL_FAST_SLABS];
#pragma unroll
    for (int s = 0; s < NLL_FAST_SLABS; ++s) {
      const bool ok = cv && s < a.nslab;
      const float x = a.mean_slab[ok ? (size_t)s * a.slab_stride + (size_t)b * a.ldm + lane : 0];
      ms[s] = ok ? x : 0.f;
    }
    const float xv = a.x[cv ? (size_t)b * a.ldx + lane : 0];
    const float v0 = wave_sum(pv ? h0 : 0.f), v1 = wave_sum(pv ? h1 : 0.f);
    const float adv = tv - fminf(v0 + bv0, v1 + bv1);
    const float wgt = fminf(expf(a.weight_mode ? a.alpha * adv : adv / a.alpha), EXP_ADV_MAX);
    const float wb = wgt * a.inv_batch;
    float m = ms[0];
#pragma unroll
    for (int s = 1; s < NLL_FAST_SLABS; ++s)
      if (s < a.nslab) m += ms[s];
    m += mb[0];
    if (a.tanh_mean) m = tanhf(m);
    const float z = cv ? (xv - m) * isg[0] : 0.f;
    const float zz = wave_sum(z * z);
    const float nlp = half_log2pi_D + 0.5f * zz + lsig;
    loss_acc += wb * nlp;
    min_acc = fminf(min_acc, nlp);
    if (cv) {
      float dm = -wb * z * isg[0];
      if (a.tanh_mean) dm *= (1.f - m * m);
      a.dmean[(size_t)b * a.ldd + lane] = dm;
      dls[0] += wb * (1.f - z * z);
    }
  }
  for (int b = row0 + wave; !fast && b < row1; b += 4) {
    // head partial sums: lane p takes part p (parts <= 64), then a wave reduction — one load deep
    float v0 = 0.f, v1 = 0.f;
    for (int p = lane; p < a.parts; p += 64) {
      v0 += a.hp_v[0][(size_t)p * a.B + b];
      v1 += a.hp_v[1][(size_t)p * a.B + b];
    }
    v0 = wave_sum(v0); v1 = wave_sum(v1);
    const float adv = a.target_v[b] - fminf(v0 + bv0, v1 + bv1);
    const float wgt = fminf(expf(a.weight_mode ? a.alpha * adv : adv / a.alpha), EXP_ADV_MAX);
    const float wb = wgt * a.inv_batch;
    float zz = 0.f;
    float z[NLL_MAX_COLS_PER_LANE], mu[NLL_MAX_COLS_PER_LANE];
#pragma unroll
    for (int c = 0; c < NLL_MAX_COLS_PER_LANE; ++c) {
      const int j = lane + 64 * c;
      z[c] = 0.f; mu[c] = 0.f;
      if (c < ncl && j < a.D) {
        float m = a.mean_slab[(size_t)b * a.ldm + j];
#pragma unroll 8
        for (int s = 1; s < a.nslab; ++s) m += a.mean_slab[(size_t)s * a.slab_stride + (size_t)b * a.ldm + j];
        m += mb[c];
        if (a.tanh_mean) m = tanhf(m);
        mu[c] = m;
        z[c] = (a.x[(size_t)b * a.ldx + j] - m) * isg[c];
        zz += z[c] * z[c];
      }
    }
    zz = wave_sum(zz);
    const float nlp = half_log2pi_D + 0.5f * zz + lsig;
    loss_acc += wb * nlp;
    min_acc = fminf(min_acc, nlp);
#pragma unroll
    for (int c = 0; c < NLL_MAX_COLS_PER_LANE; ++c) {
      const int j = lane + 64 * c;
      if (c < ncl && j < a.D) {
        float dm = -wb * z[c] * isg[c];
        if (a.tanh_mean) dm *= (1.f - mu[c] * mu[c]);
        a.dmean[(size_t)b * a.ldd + j] = dm;
        dls[c] += wb * (1.f - z[c] * z[c]);
      }
    }
  }
  // zero the padding columns of dmean once per row (keeps the (B, ldd) buffer clean for 16-byte loads)
  for (int b = row0 + wave; b < row1; b += 4)
    for (int j = a.D + lane; j < a.ldd; j += 64) a.dmean[(size_t)b * a.ldd + j] = 0.f;

#pragma unroll
  for (int c = 0; c < NLL_MAX_COLS_PER_LANE; ++c) sh_dls[wave][c * 64 + lane] = dls[c];
  if (lane == 0) { sh_loss[wave] = loss_acc; sh_min[wave] = min_acc; }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < NLL_MAX_COLS_PER_LANE; ++c) {
      const int j = lane + 64 * c;
      if (c < ncl && j < a.D) {
        const float ls = a.log_std[j];          // clamp(log_std, -5, 2) passes no gradient outside its range
        const float inside = (ls >= LOG_STD_MIN && ls <= LOG_STD_MAX) ? 1.f : 0.f;
        a.part_dls[(size_t)blockIdx.x * a.D + j] = inside *
            (sh_dls[0][c * 64 + lane] + sh_dls[1][c * 64 + lane] + sh_dls[2][c * 64 + lane] + sh_dls[3][c * 64 + lane]);
      }
    }
    if (lane == 0) {
      a.part_loss[blockIdx.x] = sh_loss[0] + sh_loss[1] + sh_loss[2] + sh_loss[3];
      a.part_min[blockIdx.x] = fminf(fminf(sh_min[0], sh_min[1]), fminf(sh_min[2], sh_min[3]));
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// split-K / per-block-partial combine for several outputs: out = act(scale * sum_s slab_s + bias)
// ---------------------------------------------------------------------------------------------------
struct ReduceJob {
  float* out; const float* slab; const float* bias;
  long n; long stride; int nslab; int ncols; int act;
  int op;          // 0: sum of the slabs, 1: minimum
  float scale;     // applied to the sum (1 = none)
  long adam_off;   // fused into the Adam launch only: float offset of out[0] inside the parameter group, or -1
  int width;       // outputs per block: 32 (8 slab lanes) or 4 (64 slab lanes), set by the host from nslab
};
constexpr int MAX_REDUCE_JOBS = 12;
struct ReduceArgs { int njobs; ReduceJob job[MAX_REDUCE_JOBS]; };

// One chunk of `j.width` outputs of a reduce job by a 256-thread block: the block is width output columns x
// (256 / width) slab lanes; lane ty takes slabs ty, ty + L, ... with four loads in flight, then the L partial results
// are combined in lane order through LDS — the load chain is nslab / (4 L) round trips deep.  width = 32 (8 slab lanes)
// for short slab lists, 4 (64 lanes) for the long lists of per-block partials (256 loss partials: one round trip).
// Returns the combined value in the threads with ty == 0 (valid when i < j.n).  Fixed order: deterministic.
constexpr int REDUCE_LDS_FLOATS = 64 * 5;      // >= L * (width + 1) for (8, 32) and (64, 4)
__device__ __forceinline__ float reduce_chunk(const ReduceJob& j, long i0, float* sh) {
  const int W = j.width, L = 256 / W;
  const int tx = threadIdx.x % W, ty = threadIdx.x / W;
  const long i = i0 + tx;
  const float id = j.op ? INFINITY : 0.f;
  float s0 = id, s1 = id, s2 = id, s3 = id;
  if (i < j.n) {
    const float* p = j.slab + i;
    int k = ty;
    for (; k + 3 * L < j.nslab; k += 4 * L) {
      const float x0 = p[(long)k * j.stride], x1 = p[(long)(k + L) * j.stride];
      const float x2 = p[(long)(k + 2 * L) * j.stride], x3 = p[(long)(k + 3 * L) * j.stride];
      if (j.op) { s0 = fminf(s0, x0); s1 = fminf(s1, x1); s2 = fminf(s2, x2); s3 = fminf(s3, x3); }
      else { s0 += x0; s1 += x1; s2 += x2; s3 += x3; }
    }
    for (; k < j.nslab; k += L) {
      const float x = p[(long)k * j.stride];
      s0 = j.op ? fminf(s0, x) : s0 + x;
    }
  }
  const float s = j.op ? fminf(fminf(s0, s1), fminf(s2, s3)) : (s0 + s1) + (s2 + s3);
  sh[ty * (W + 1) + tx] = s;
  __syncthreads();
  if (W == 4) {
    // 64 slab lanes per output: wave w gathers the 64 partials of output w and combines them with the butterfly of
    // wave_sum / wave_min — a fixed tree, six steps deep, instead of a 63-long chain of dependent LDS reads and adds on
    // one thread (that chain, ~4 us, was the critical path of the Adam launches that fold these combines in)
    const int w = threadIdx.x >> 6, ln = threadIdx.x & 63;
    float v = sh[ln * (W + 1) + w];
    v = j.op ? wave_min(v) : wave_sum(v);
    __syncthreads();
    if (ln == 0) sh[w] = v;
    __syncthreads();
  }
  float t = 0.f;
  if (ty == 0 && i < j.n) {
    t = sh[tx];
    if (W != 4)
      for (int q = 1; q < L; ++q) t = j.op ? fminf(t, sh[q * (W + 1) + tx]) : t + sh[q * (W + 1) + tx];
    t *= j.scale;
    if (j.bias) t += j.bias[i % j.ncols];
    if (j.act == 1) t = fmaxf(t, 0.f);
    else if (j.act == 2) t = tanhf(t);
  }
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void multi_reduce_kernel(const ReduceArgs a) {
  __shared__ float sh[REDUCE_LDS_FLOATS];
  const ReduceJob& j = a.job[blockIdx.y];
  const int W = j.width, tx = threadIdx.x % W, ty = threadIdx.x / W;
  for (long i0 = (long)blockIdx.x * W; i0 < j.n; i0 += (long)gridDim.x * W) {
    const float t = reduce_chunk(j, i0, sh);
    if (ty == 0 && i0 + tx < j.n) j.out[i0 + tx] = t;
  }
}

// ---------------------------------------------------------------------------------------------------
// torch.optim.Adam (single-tensor arithmetic) over a flat parameter group, with the
// Polyak target update (util/util.py:54-56) fused in:  28 B/param (+8 B/param with the target).
//   step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) are host doubles rounded to fp32.
// The same launch can also FINISH the gradient: split-K slabs and per-block partial sums that the backward kernels
// left behind are combined here instead of in a launch of their own —
//   * short regions (nslab <= ADAM_SWEEP_SLABS, e.g. the split-K slabs of the (H, S) input-layer weights) inside the
//     float4 sweep: the thread that owns a float4 sums its slabs first (multi_reduce's order: 8 interleaved partial
//     sums, combined in index order, so the result is bit-identical to the separate combine launch);
//   * long ones (per-block partials of the loss heads: B/16 .. B/4 of them) by extra blocks that run reduce_chunk and
//     then apply Adam to their 32 outputs; jobs with adam_off < 0 (the loss statistics) just store.
// The combined gradient is also written to g, so the gradient buffer is complete after the launch.
// ---------------------------------------------------------------------------------------------------
struct AdamScalars { float omb1, beta2, omb2, eps, step_size, bc2_sqrt, ema_beta, omeb; };
constexpr int ADAM_SWEEP_SLABS = 16;
constexpr int MAX_ADAM_REGIONS = 8;
struct AdamRegion { long lo, hi; const float* slab; long stride; int nslab; };   // floats [lo, hi), multiples of 4;
                                                                                // slab indexed from lo
struct AdamArgs {
  float* p; float* g; float* m; float* v; float* tgt;
  long n4;              // float4 in the group
  long span4;           // float4 per sweep block (contiguous)
  int reduce_blocks;    // blocks [0, reduce_blocks) run reduce jobs (dispatched first: their load chains are the
                        // longest), the rest sweep
  AdamScalars s;
  int nregions; AdamRegion region[MAX_ADAM_REGIONS];
  int nskip; long skip_lo[MAX_REDUCE_JOBS], skip_hi[MAX_REDUCE_JOBS];   // float ranges owned by reduce blocks
  int job_block0[MAX_REDUCE_JOBS + 1];                                   // first reduce block of job j (prefix sums)
  ReduceArgs r;
};

// Polyak update of one element (util/util.py:54-56): target <- (1 - beta) * target + beta * source.  One definition
// for the fused sweep and the stand-alone kernel, so both round the same way.
__device__ __forceinline__ float ema1(float t, float p, float omeb, float ema_beta) { return t * omeb + ema_beta * p; }

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamScalars& s) {
  m = m + s.omb1 * (g - m);
  v = v * s.beta2 + s.omb2 * g * g;
  p = p - s.step_size * (m / (sqrtf(v) / s.bc2_sqrt + s.eps));
}

__global__ __launch_bounds__(256) void adam_ema_kernel(const AdamArgs a) {
  __shared__ float sh[REDUCE_LDS_FLOATS];
  const AdamScalars s = a.s;
  if ((int)blockIdx.x < a.reduce_blocks) {
    // ---- reduce block: one chunk of one job, then Adam on those outputs --------------------------
    const int rb = blockIdx.x;
    int ji = 0;
#pragma unroll
    for (int q = 1; q < MAX_REDUCE_JOBS; ++q)
      if (q < a.r.njobs && rb >= a.job_block0[q]) ji = q;
    const ReduceJob& j = a.r.job[ji];
    const long i0 = (long)(rb - a.job_block0[ji]) * j.width;
    const float t = reduce_chunk(j, i0, sh);
    const int tx = threadIdx.x % j.width, ty = threadIdx.x / j.width;
    if (ty == 0 && i0 + tx < j.n) {
      j.out[i0 + tx] = t;
      if (j.adam_off >= 0) {
        const long e = j.adam_off + i0 + tx;
        float pp = a.p[e], mm = a.m[e], vv = a.v[e];
        adam1(pp, t, mm, vv, s);
        a.p[e] = pp; a.m[e] = mm; a.v[e] = vv;
        if (a.tgt) a.tgt[e] = ema1(a.tgt[e], pp, s.omeb, s.ema_beta);
      }
    }
    return;
  }
  // ---- sweep block: a contiguous span of float4 ----------------------------------------------------------------
  float4* p4 = reinterpret_cast<float4*>(a.p);
  float4* g4 = reinterpret_cast<float4*>(a.g);
  float4* m4 = reinterpret_cast<float4*>(a.m);
  float4* v4 = reinterpret_cast<float4*>(a.v);
  float4* t4 = reinterpret_cast<float4*>(a.tgt);
  const long lo4 = (long)(blockIdx.x - a.reduce_blocks) * a.span4, hi4 = min(a.n4, lo4 + a.span4);
  // block-uniform: does this span touch a slab region or a range owned by reduce blocks at all?
  bool special = false;
  for (int q = 0; q < a.nregions; ++q) special = special || (lo4 * 4 < a.region[q].hi && hi4 * 4 > a.region[q].lo);
  for (int q = 0; q < a.nskip; ++q) special = special || (lo4 * 4 < a.skip_hi[q] && hi4 * 4 > a.skip_lo[q]);
  for (long i = lo4 + threadIdx.x; i < hi4; i += 256) {
    float4 gg;
    bool have_g = false;
    if (special) {
      const long e = i * 4;
      bool skip = false;
      for (int q = 0; q < a.nskip; ++q) skip = skip || (e >= a.skip_lo[q] && e < a.skip_hi[q]);
      if (skip) continue;
      for (int q = 0; q < a.nregions; ++q) {
        const AdamRegion& R = a.region[q];
        if (e >= R.lo && e < R.hi) {
          // 8 interleaved partial sums (slab k goes to partial k & 7), combined in index order: multi_reduce's order
          float4 part[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) part[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          const float* base = R.slab + (e - R.lo);
#pragma unroll
          for (int k = 0; k < ADAM_SWEEP_SLABS; ++k)
            if (k < R.nslab) {
              const float4 x = *reinterpret_cast<const float4*>(base + (long)k * R.stride);
              part[k & 7].x += x.x; part[k & 7].y += x.y; part[k & 7].z += x.z; part[k & 7].w += x.w;
            }
          gg = part[0];
#pragma unroll
          for (int u = 1; u < 8; ++u) { gg.x += part[u].x; gg.y += part[u].y; gg.z += part[u].z; gg.w += part[u].w; }
          g4[i] = gg;
          have_g = true;
        }
      }
    }
    if (!have_g) gg = g4[i];
    float4 pp = p4[i];
    float4 mm = m4[i], vv = v4[i];
    adam1(pp.x, gg.x, mm.x, vv.x, s); adam1(pp.y, gg.y, mm.y, vv.y, s);
    adam1(pp.z, gg.z, mm.z, vv.z, s); adam1(pp.w, gg.w, mm.w, vv.w, s);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (a.tgt) {
      float4 tt = t4[i];
      tt.x = ema1(tt.x, pp.x, s.omeb, s.ema_beta); tt.y = ema1(tt.y, pp.y, s.omeb, s.ema_beta);
      tt.z = ema1(tt.z, pp.z, s.omeb, s.ema_beta); tt.w = ema1(tt.w, pp.w, s.omeb, s.ema_beta);
      t4[i] = tt;
    }
  }
}

// Stand-alone Polyak sweep (12 B/param): the data-parallel path applies Adam to a 1/N slice per rank, all-gathers the
// parameters and then updates the whole target locally — cheaper than gathering the target as well.
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ tgt, const float* __restrict__ src, long n4,
                                                   float omeb, float ema_beta) {
  float4* t4 = reinterpret_cast<float4*>(tgt);
  const float4* p4 = reinterpret_cast<const float4*>(src);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 tt = t4[i];
  const float4 pp = p4[i];
  tt.x = ema1(tt.x, pp.x, omeb, ema_beta); tt.y = ema1(tt.y, pp.y, omeb, ema_beta);
  tt.z = ema1(tt.z, pp.z, omeb, ema_beta); tt.w = ema1(tt.w, pp.w, omeb, ema_beta);
  t4[i] = tt;
}

// ---------------------------------------------------------------------------------------------------
// Small-batch Linear (+bias, +ReLU/tanh): the inference path (SORL.select_action, GaussianPolicy.act, TwinV.both
// on a handful of states; reference agent/sorl.py:71-76, agent/policy.py:30-33).  At B <= SMALL_FWD_MAX_B a layer is
// a GEMV bound by reading W once: one wave per output row, 16-byte loads of the row, the B inputs staged in LDS,
// wave-shuffle reduction.  blockIdx.y selects one of up to 2 networks (the twins).  3 launches per 2-hidden-layer MLP
// instead of pack + 2 grouped GEMMs + split-K mean + finish.
// ---------------------------------------------------------------------------------------------------
constexpr int SMALL_FWD_MAX_B = 8;
struct SmallFwdArgs {
  const float* W[2]; const float* bias[2];    // (N, K) row-major, (N)
  const float* X[2]; long ldx[2];             // (B, K) inputs, row stride in floats
  float* Y[2]; long ldy[2];                   // (B, N) outputs
  int N, K, B, act;
};

__global__ __launch_bounds__(256) void small_fwd_kernel(const SmallFwdArgs a) {
  extern __shared__ float xs[];               // B * K
  const int net = blockIdx.y;
  const float* __restrict__ X = a.X[net];
  for (int i = threadIdx.x; i < a.B * a.K; i += 256) {
    const int b = i / a.K, k = i - b * a.K;
    xs[i] = X[(long)b * a.ldx[net] + k];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* __restrict__ W = a.W[net];
  const bool vec = (a.K & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  for (int n = blockIdx.x * 4 + wave; n < a.N; n += gridDim.x * 4) {
    const float* __restrict__ w = W + (size_t)n * a.K;
    float acc[SMALL_FWD_MAX_B];
#pragma unroll
    for (int b = 0; b < SMALL_FWD_MAX_B; ++b) acc[b] = 0.f;
    if (vec) {
      for (int k = lane * 4; k < a.K; k += 256) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
        for (int b = 0; b < SMALL_FWD_MAX_B; ++b)
          if (b < a.B) {
            const float4 xv = *reinterpret_cast<const float4*>(xs + b * a.K + k);
            acc[b] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[b]))));
          }
      }
    } else {
      for (int k = lane; k < a.K; k += 64) {
        const float wv = w[k];
#pragma unroll
        for (int b = 0; b < SMALL_FWD_MAX_B; ++b)
          if (b < a.B) acc[b] = fmaf(wv, xs[b * a.K + k], acc[b]);
      }
    }
    const float bv = a.bias[net] ? a.bias[net][n] : 0.f;
#pragma unroll
    for (int b = 0; b < SMALL_FWD_MAX_B; ++b)
      if (b < a.B) {
        float v = wave_sum(acc[b]) + bv;
        if (a.act == 1) v = fmaxf(v, 0.f);
        else if (a.act == 2) v = tanhf(v);
        if (lane == 0) a.Y[net][(long)b * a.ldy[net] + n] = v;
      }
  }
}

// v = sum(parts) + b for both twins (forward-only API: TwinV.both)
__global__ void head_finish_kernel(const float* __restrict__ hp0, const float* __restrict__ hp1,
                                   const float* __restrict__ b0, const float* __restrict__ b1, int parts, int B,
                                   float* __restrict__ out0, float* __restrict__ out1) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float v0 = 0.f, v1 = 0.f;
  for (int p = 0; p < parts; ++p) { v0 += hp0[(size_t)p * B + b]; v1 += hp1[(size_t)p * B + b]; }
  out0[b] = v0 + b0[0];
  out1[b] = v1 + b1[0];
}

// mean = act(sum of split-K slabs + bias), written to a caller view with arbitrary row stride
__global__ void mean_finish_kernel(const float* __restrict__ slab, int nslab, long stride, int B, int D, int ldm,
                                   const float* __restrict__ bias, int tanh_mean, float* __restrict__ out,
                                   long out_rs) {
  const long n = (long)B * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / D;
    const int j = (int)(i - b * D);
    float m = slab[b * ldm + j];
    for (int s = 1; s < nslab; ++s) m += slab[(long)s * stride + b * ldm + j];
    m += bias[j];
    if (tanh_mean) m = tanhf(m);
    out[b * out_rs + j] = m;
  }
}

// ---------------------------------------------------------------------------------------------------
// replay gather: rows[idx[i], :] -> out[i, :]   (device-resident packed-row store, K12)
// one wave per row; 16-byte lanes when the row width allows
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ rows, long row_stride,
                                                           const int64_t* __restrict__ idx, int n, int width,
                                                           float* __restrict__ out, long out_stride) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool vec = (width & 3) == 0 && (row_stride & 3) == 0 && (out_stride & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  for (int i = wave; i < n; i += nwaves) {
    const float* src = rows + idx[i] * row_stride;
    float* dst = out + (long)i * out_stride;
    if (vec) {
      for (int c = lane * 4; c < width; c += 256)
        *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
    } else {
      for (int c = lane; c < width; c += 64) dst[c] = src[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// device sampler: `batch` DISTINCT row indices in [0, n) per call (uniform without replacement, like
// np.random.choice(size, B, replace=False) in buffer/replay_buffer.py:64, but O(B) instead of O(N)):
// index i is the image of i under a keyed pseudo-random permutation of [0, n) — a 4-round Feistel
// network on 2*hb bits, cycle-walked into range.  Key = (seed, step).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ int64_t feistel_index(int64_t n, int64_t i, uint64_t seed, uint64_t step, int hb) {
  const uint32_t mask = (1u << hb) - 1u;
  uint32_t keys[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    keys[r] = mix32((uint32_t)(seed >> (r & 1 ? 32 : 0)) ^ mix32((uint32_t)step * 4u + r) ^ (uint32_t)(step >> 30));
  // Cycle walking: apply the bijection of [0, 4^hb) until the image lands in [0, n).  The walk follows the cycle
  // of i, which contains i < n itself, so it always ends (the domain is < 4n: fewer than 4 rounds on average) and
  // the restriction to [0, n) is again a bijection — every position maps to a distinct row.
  uint64_t x = (uint64_t)i;
  do {
    uint32_t L = (uint32_t)(x >> hb) & mask, R = (uint32_t)x & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t f = mix32(R ^ keys[r]) & mask;
      const uint32_t t = L ^ f;
      L = R; R = t;
    }
    x = ((uint64_t)L << hb) | R;
  } while ((int64_t)x >= n);
  return (int64_t)x;
}

// sampler + gather + split in one pass: batch row i <- packed replay row perm(i), written straight
// into the step's dense buffers ([s | r | s' | d | a] wire format of por_train.py:74-78).
// One wave per batch row.
struct SampledBatchArgs {
  const float* rows; long row_stride; int64_t n_rows;
  int batch, S, A, D, Sp, Dp, hb, target_is_action;
  uint64_t seed, step;
  float* xs; float* xn; float* xt; float* rew; float* term;
  int64_t* idx_out;      // optional: the drawn indices
};

__global__ __launch_bounds__(256) void sampled_batch_kernel(const SampledBatchArgs a) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= a.batch) return;
  const int64_t r = feistel_index(a.n_rows, i, a.seed, a.step, a.hb);
  const float* __restrict__ src = a.rows + r * a.row_stride;
  const int S = a.S;
  for (int c = lane; c < a.Sp; c += 64) {
    a.xs[(size_t)i * a.Sp + c] = c < S ? src[c] : 0.f;
    a.xn[(size_t)i * a.Sp + c] = c < S ? src[S + 1 + c] : 0.f;
  }
  const int toff = a.target_is_action ? 2 * S + 2 : S + 1;
  for (int c = lane; c < a.Dp; c += 64) a.xt[(size_t)i * a.Dp + c] = c < a.D ? src[toff + c] : 0.f;
  if (lane == 0) {
    a.rew[i] = src[S];
    a.term[i] = src[2 * S + 1];
    if (a.idx_out) a.idx_out[i] = r;
  }
}

// out[i] = base + perm_{seed,step}(first + i): `batch` consecutive positions of one keyed permutation of [0, n)
__global__ void sample_indices_kernel(int64_t n, int batch, uint64_t seed, uint64_t step, int hb,
                                      int64_t base, int64_t first, int64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  out[i] = base + feistel_index(n, first + (int64_t)i, seed, step, hb);
}

// ---------------------------------------------------------------------------------------------------
// Backward through the scalar V head and the last ReLU:  dZ[b,j] = dv[b] * w[j] * 1[H[b,j] > 0], plus
// per-block partial column sums of dv[b] * H[b,j] (the head's weight gradient).  One float4 of columns
// per thread and pass; rows_per_block rows per block.  grid (ceil(B/rows), nets), block 256.
// (Doing this inside the GEMM's operand loader costs ~20 VALU per staged float4, and VALU issued
// between f32 MFMAs is not free: a separate 16 MB pass is cheaper.)
// ---------------------------------------------------------------------------------------------------
struct HeadBwdArgs {
  const float* Hact[2]; const float* w[2];
  float* dZ[2]; float* part_dw[2];      // part_dw: [nblk][H]
  // fused TD-target / expectile head (agent/por.py:81-87): v = sum(parts) + b for the 4 nets
  const float* hp_t[2]; const float* hp_v[2]; const float* b_t[2]; const float* b_v[2];
  const float* rew; const float* term;
  float* target_v; float* dv[2];
  float* part_loss; float* part_db[2];  // [nblk] each: partial v_loss and partial sum of dv (= db_L)
  int B, H, ld, parts;
  float tau, discount, inv_batch;
};
constexpr int HEAD_ROWS = 8;             // rows per block (all row loads of a thread in flight together); a wave walks HEAD_ROWS/4 rows of the head in turn, so 8 instead of 16 halves that dependent chain and fills all 256 CUs at B = 1024

__global__ __launch_bounds__(256) void relu_head_bwd_kernel(const HeadBwdArgs a) {
  __shared__ float sh_dv[2][HEAD_ROWS];
  __shared__ float sh_loss[4];
  const int net = blockIdx.y;
  const float* __restrict__ Hm = a.Hact[net];
  float* __restrict__ dZ = a.dZ[net];
  const int row0 = blockIdx.x * HEAD_ROWS;
  const bool vec = (a.H & 3) == 0 && (a.ld & 3) == 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // -- loss head for this block's rows: wave w handles rows w, w+4, ...; lanes split the partial sums.
  //    Both nets' blocks compute the same numbers (cheap); the net-0 block publishes them. ---------------
  float loss = 0.f;
  for (int r = wave; r < HEAD_ROWS; r += 4) {
    const int b = row0 + r;
    float t0 = 0.f, t1 = 0.f, v0 = 0.f, v1 = 0.f;
    if (b < a.B)
      for (int p = lane; p < a.parts; p += 64) {
        const size_t o = (size_t)p * a.B + b;
        t0 += a.hp_t[0][o]; t1 += a.hp_t[1][o]; v0 += a.hp_v[0][o]; v1 += a.hp_v[1][o];
      }
    t0 = wave_sum(t0); t1 = wave_sum(t1); v0 = wave_sum(v0); v1 = wave_sum(v1);
    float d0 = 0.f, d1 = 0.f;
    if (b < a.B) {
      t0 += a.b_t[0][0]; t1 += a.b_t[1][0]; v0 += a.b_v[0][0]; v1 += a.b_v[1][0];
      const float tgt = a.rew[b] + (1.f - a.term[b]) * a.discount * fminf(t0, t1);
      const float u0 = tgt - v0, u1 = tgt - v1;
      const float w0 = fabsf(a.tau - (u0 < 0.f ? 1.f : 0.f)), w1 = fabsf(a.tau - (u1 < 0.f ? 1.f : 0.f));
      d0 = -w0 * u0 * a.inv_batch; d1 = -w1 * u1 * a.inv_batch;
      loss += 0.5f * (w0 * u0 * u0 + w1 * u1 * u1);
      if (net == 0 && lane == 0) { a.target_v[b] = tgt; a.dv[0][b] = d0; a.dv[1][b] = d1; }
    }
    if (lane == 0) { sh_dv[0][r] = d0; sh_dv[1][r] = d1; }
  }
  if (lane == 0) sh_loss[wave] = loss;
  __syncthreads();
  if (threadIdx.x == 0) {                     // fixed summation order
    float dsm = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) dsm += sh_dv[net][r];
    a.part_db[net][blockIdx.x] = dsm;
    if (net == 0) a.part_loss[blockIdx.x] = sh_loss[0] + sh_loss[1] + sh_loss[2] + sh_loss[3];
  }
  float d[HEAD_ROWS];
#pragma unroll
  for (int r = 0; r < HEAD_ROWS; ++r) d[r] = sh_dv[net][r];
  if (vec) {
    // 16-byte path: a float4 of columns per thread and pass; the 16 row loads are issued back to back
    for (int c0 = threadIdx.x * 4; c0 < a.H; c0 += 1024) {
      const float4 w4 = *reinterpret_cast<const float4*>(a.w[net] + c0);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 h[HEAD_ROWS];
#pragma unroll
      for (int r = 0; r < HEAD_ROWS; ++r) {
        const int b = min(row0 + r, a.B - 1);                     // clamped: rows past B re-read the last row
        h[r] = *reinterpret_cast<const float4*>(Hm + (size_t)b * a.ld + c0);
      }
#pragma unroll
      for (int r = 0; r < HEAD_ROWS; ++r) {
        const float dr = d[r];                                     // 0 for rows past B
        float4 z;
        z.x = h[r].x > 0.f ? dr * w4.x : 0.f; z.y = h[r].y > 0.f ? dr * w4.y : 0.f;
        z.z = h[r].z > 0.f ? dr * w4.z : 0.f; z.w = h[r].w > 0.f ? dr * w4.w : 0.f;
        if (row0 + r < a.B) *reinterpret_cast<float4*>(dZ + (size_t)(row0 + r) * a.ld + c0) = z;
        acc.x += dr * h[r].x; acc.y += dr * h[r].y; acc.z += dr * h[r].z; acc.w += dr * h[r].w;
      }
      *reinterpret_cast<float4*>(a.part_dw[net] + (size_t)blockIdx.x * a.H + c0) = acc;
    }
  } else {
    // odd widths: one column per thread and pass
    for (int c = threadIdx.x; c < a.H; c += 256) {
      const float wc = a.w[net][c];
      float acc = 0.f;
      for (int r = 0; r < HEAD_ROWS; ++r) {
        const int b = row0 + r;
        if (b >= a.B) break;
        const float hv = Hm[(size_t)b * a.ld + c];
        dZ[(size_t)b * a.ld + c] = hv > 0.f ? d[r] * wc : 0.f;
        acc += d[r] * hv;
      }
      a.part_dw[net][(size_t)blockIdx.x * a.H + c] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm (+ReLU) between a hidden Linear and its activation (util/util.py:36-37; nn.LayerNorm(H): biased
// variance, eps 1e-5 inside the sqrt, affine).  One block per row group; a block walks its rows one at a
// time, 256 threads striding over the H columns (H <= 256 * LN_MAX_COLS).
//   forward : Z (B, ld) -> H = relu(xhat * gamma + beta) written in place of Z; optionally keeps xhat and
//             rstd for the backward pass and accumulates the fused scalar head v[b] = sum_j H[b,j] w[j]
//   backward: dZ = (dxh - mean(dxh) - xhat * mean(dxh * xhat)) * rstd, dxh = dy * gamma, dy = dH . 1[H > 0]
//             with dH either read from memory or made on the fly as dv[b] * w[j] (top layer); per-block
//             partial column sums of dgamma = dy * xhat, dbeta = dy and (top layer) dw = dv * H
// ---------------------------------------------------------------------------------------------------
constexpr int LN_MAX_COLS = 8;
constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();                         // sh may still be read from the previous use
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

struct LnFwdArgs {
  float* Z[4]; float* xhat[4]; float* rstd[4];        // per net; xhat/rstd may be null (no backward)
  const float* gamma[4]; const float* beta[4];
  const float* headw[4]; const float* headb[4]; float* headout[4];   // fused head (may be null)
  int nnets, B, H, ld, rows_per_block;
};

__global__ __launch_bounds__(256) void ln_relu_fwd_kernel(const LnFwdArgs a) {
  __shared__ float sh[4];
  const int net = blockIdx.y;
  float* __restrict__ Z = a.Z[net];
  const float* __restrict__ g = a.gamma[net];
  const float* __restrict__ be = a.beta[net];
  const int row0 = blockIdx.x * a.rows_per_block, row1 = min(a.B, row0 + a.rows_per_block);
  const float invH = 1.f / (float)a.H;
  for (int b = row0; b < row1; ++b) {
    float z[LN_MAX_COLS];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_COLS; ++c) {
      const int j = threadIdx.x + 256 * c;
      z[c] = j < a.H ? Z[(size_t)b * a.ld + j] : 0.f;
      s += z[c];
    }
    const float mu = block_sum_256(s, sh) * invH;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_COLS; ++c) {
      const int j = threadIdx.x + 256 * c;
      const float d = j < a.H ? z[c] - mu : 0.f;
      q += d * d;
    }
    const float var = block_sum_256(q, sh) * invH;
    const float rs = 1.f / sqrtf(var + LN_EPS);
    float hv = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_COLS; ++c) {
      const int j = threadIdx.x + 256 * c;
      if (j < a.H) {
        const float xh = (z[c] - mu) * rs;
        const float h = fmaxf(xh * g[j] + be[j], 0.f);
        Z[(size_t)b * a.ld + j] = h;
        if (a.xhat[net]) a.xhat[net][(size_t)b * a.ld + j] = xh;
        if (a.headw[net]) hv += h * a.headw[net][j];
      }
    }
    if (a.rstd[net] && threadIdx.x == 0) a.rstd[net][b] = rs;
    if (a.headw[net]) {
      const float v = block_sum_256(hv, sh);
      if (threadIdx.x == 0) a.headout[net][b] = v + (a.headb[net] ? a.headb[net][0] : 0.f);
    }
  }
}

struct LnBwdArgs {
  const float* dH[2];            // (B, ld) upstream gradient, or null -> dv[b] * headw[j]
  const float* dv[2]; const float* headw[2];
  const float* Hact[2]; const float* xhat[2]; const float* rstd[2]; const float* gamma[2];
  float* dZ[2];                  // (B, ld)
  float* part_dgamma[2]; float* part_dbeta[2]; float* part_dhead[2];   // [nblk][H]; dhead only for the top layer
  int nnets, B, H, ld, rows_per_block;
};

__global__ __launch_bounds__(256) void ln_relu_bwd_kernel(const LnBwdArgs a) {
  __shared__ float sh[4];
  const int net = blockIdx.y;
  const float* __restrict__ Hm = a.Hact[net];
  const float* __restrict__ xh = a.xhat[net];
  const float* __restrict__ g = a.gamma[net];
  const bool top = a.dH[net] == nullptr;
  float gam[LN_MAX_COLS], hw[LN_MAX_COLS], dg[LN_MAX_COLS], db[LN_MAX_COLS], dh[LN_MAX_COLS];
#pragma unroll
  for (int c = 0; c < LN_MAX_COLS; ++c) {
    const int j = threadIdx.x + 256 * c;
    gam[c] = j < a.H ? g[j] : 0.f;
    hw[c] = (top && j < a.H) ? a.headw[net][j] : 0.f;
    dg[c] = db[c] = dh[c] = 0.f;
  }
  const int row0 = blockIdx.x * a.rows_per_block, row1 = min(a.B, row0 + a.rows_per_block);
  const float invH = 1.f / (float)a.H;
  for (int b = row0; b < row1; ++b) {
    const float dvb = top ? a.dv[net][b] : 0.f;
    float dxh[LN_MAX_COLS], x[LN_MAX_COLS];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_COLS; ++c) {
      const int j = threadIdx.x + 256 * c;
      dxh[c] = 0.f; x[c] = 0.f;
      if (j < a.H) {
        const size_t o = (size_t)b * a.ld + j;
        const float h = Hm[o];
        const float up = top ? dvb * hw[c] : a.dH[net][o];
        const float dy = h > 0.f ? up : 0.f;
        x[c] = xh[o];
        dg[c] += dy * x[c];
        db[c] += dy;
        if (top) dh[c] += dvb * h;
        dxh[c] = dy * gam[c];
        s1 += dxh[c];
        s2 += dxh[c] * x[c];
      }
    }
    const float m1 = block_sum_256(s1, sh) * invH;
    const float m2 = block_sum_256(s2, sh) * invH;
    const float rs = a.rstd[net][b];
#pragma unroll
    for (int c = 0; c < LN_MAX_COLS; ++c) {
      const int j = threadIdx.x + 256 * c;
      if (j < a.H) a.dZ[net][(size_t)b * a.ld + j] = (dxh[c] - m1 - x[c] * m2) * rs;
    }
  }
#pragma unroll
  for (int c = 0; c < LN_MAX_COLS; ++c) {
    const int j = threadIdx.x + 256 * c;
    if (j < a.H) {
      const size_t o = (size_t)blockIdx.x * a.H + j;
      a.part_dgamma[net][o] = dg[c];
      a.part_dbeta[net][o] = db[c];
      if (top) a.part_dhead[net][o] = dh[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// CQL(H) on a plain-DQN TD loss for discrete actions  (src/porl/train/cql_trainer.py:60-124)
//   y = r + gamma * max_a Q_tgt(s', a) * (1 - d);  td = mean((Q(s)[a] - y)^2)
//   pen = mean(logsumexp_a Q(s, a) - ln A - Q(s)[a]);  loss = td + alpha * pen
//   dL/dQ[b, j] = alpha/B * softmax_j + 1[j == a] * (2/B (Q[a] - y) - alpha/B)
// One thread per row (A <= 64), per-block partial sums, fixed-order finalize.
// ---------------------------------------------------------------------------------------------------
struct CqlLossArgs {
  const float* Q; const float* Qn; int ldq;
  const int64_t* actions; const float* rew; const float* done;
  float* dQ;                       // (B, ldq), padding columns zeroed
  float* part_td; float* part_pen; // per block
  int B, A;
  float gamma, alpha, inv_batch, log_A;
  // DQN variants (the arithmetic of qnet_fused.hpp:qf_loss_rows, so wide networks behave like narrow ones):
  const float* Qon;                // Double DQN (ddqn_trainer.py:58-99): Q_online(s', .), bootstrap action = its first argmax;
                                   // may alias dQ (a thread reads its row before it writes it); null = plain max
  const float* is_w;               // (B,) per-sample loss weights, or null
  const float* w_uniform;          // device scalar multiplying every sample's loss, or null
  float* td_abs;                   // (B,) |Q(s)[a] - target|, or null
  const float* next_mask;          // (B, A) 0/1: BCQ's allowed bootstrap actions (policy/bcq.py:59-74), or null
  int td_off;                      // 1: no TD term (behaviour-policy pre-training, bcq.py:23-47)
};

__global__ __launch_bounds__(256) void cql_loss_kernel(const CqlLossArgs a) {
  __shared__ float red[2][4];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  float td = 0.f, pen = 0.f;
  if (b < a.B) {
    const float* q = a.Q + (size_t)b * a.ldq;
    const float* qn = a.Qn + (size_t)b * a.ldq;
    float mx = -INFINITY, mxn = -INFINITY;
    for (int j = 0; j < a.A; ++j) { mx = fmaxf(mx, q[j]); mxn = fmaxf(mxn, qn[j]); }
    float se = 0.f;
    for (int j = 0; j < a.A; ++j) se += expf(q[j] - mx);
    const float lse = mx + logf(se);
    const int act = (int)a.actions[b];
    const float qa = q[act];
    float qnext = mxn;
    if (a.Qon) {                                  // Double DQN: the online network picks, the target network values
      const float* qo = a.Qon + (size_t)b * a.ldq;
      int am = 0;
      float best = qo[0];
      for (int j = 1; j < a.A; ++j) if (qo[j] > best) { best = qo[j]; am = j; }     // first maximum, like torch.argmax
      qnext = qn[am];
    }
    if (a.next_mask) {
      const float* mk = a.next_mask + (size_t)b * a.A;
      int best = 0;
      float bestv = qn[0] + (mk[0] - 1.f) * 1e10f;
      for (int j = 1; j < a.A; ++j) {
        const float v = qn[j] + (mk[j] - 1.f) * 1e10f;
        if (v > bestv) { bestv = v; best = j; }
      }
      qnext = qn[best];
    }
    const float y = a.rew[b] + a.gamma * qnext * (1.f - a.done[b]);
    const float diff = a.td_off ? 0.f : qa - y;
    float wgt = a.is_w ? a.is_w[b] : 1.f;
    if (a.w_uniform) wgt *= a.w_uniform[0];
    td = wgt * (diff * diff);
    pen = lse - a.log_A - qa;
    if (a.td_abs) a.td_abs[b] = fabsf(diff);
    float* dq = a.dQ + (size_t)b * a.ldq;
    const float ab = a.alpha * a.inv_batch;
    for (int j = 0; j < a.A; ++j) {
      float g = ab * expf(q[j] - lse);
      if (j == act) g += 2.f * a.inv_batch * wgt * diff - ab;
      dq[j] = g;
    }
    for (int j = a.A; j < a.ldq; ++j) dq[j] = 0.f;
  }
  td = wave_sum(td); pen = wave_sum(pen);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = td; red[1][wave] = pen; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.part_td[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    a.part_pen[blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// Minibatch gather of the multi-launch Q-network path: rows idx[b] of the replay arrays -> the engine's staging buffers
// (states / next states zero padded to ld columns).  grid (ceil(B * ld / 256)), the first B threads also move the
// row's action, reward and done flag.
struct QnetGatherArgs {
  const float* states; const float* next_states; long s_rs, n_rs;
  const int64_t* actions; const float* rew; const float* done; const int64_t* idx;
  float* xs; float* xn; int64_t* act_out; float* rew_out; float* done_out;
  int B, S, ld;
};
__global__ __launch_bounds__(256) void qnet_gather_kernel(const QnetGatherArgs a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)a.B * a.ld) return;
  const int b = (int)(i / a.ld), c = (int)(i - (long)b * a.ld);
  const long row = a.idx ? a.idx[b] : (long)b;
  a.xs[i] = c < a.S ? a.states[row * a.s_rs + c] : 0.f;
  a.xn[i] = c < a.S ? a.next_states[row * a.n_rs + c] : 0.f;
  if (c == 0) { a.act_out[b] = a.actions[row]; a.rew_out[b] = a.rew[row]; a.done_out[b] = a.done[row]; }
}

// stats[0] = loss, [1] = td, [2] = penalty (this rank's shares)
__global__ __launch_bounds__(64) void cql_finalize_kernel(const float* __restrict__ part_td,
                                                           const float* __restrict__ part_pen, int nblk,
                                                           float inv_batch, float alpha, float* __restrict__ stats) {
  float td = 0.f, pen = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 64) { td += part_td[k]; pen += part_pen[k]; }
  td = wave_sum(td); pen = wave_sum(pen);
  if (threadIdx.x == 0) {
    td *= inv_batch; pen *= inv_batch;
    stats[0] = td + alpha * pen; stats[1] = td; stats[2] = pen;
  }
}

// mask[b, j] = softmax(logits[b, :])[j] > threshold   (BehaviorPolicy.sample, src/porl/net/behavior_policy.py:41-55)
// (write_probs == 1: the probabilities themselves, BehaviorPolicy.forward, :30-39; == 2: log-probabilities, the
// log_softmax over the atoms that ends CategoricalQNetwork.forward, categorical_q_network.py:76-78)
__global__ void softmax_mask_kernel(const float* __restrict__ logits, long ld, int B, int A, float threshold,
                                    int write_probs, float* __restrict__ mask) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* z = logits + (long)b * ld;
  float mx = -INFINITY;
  for (int j = 0; j < A; ++j) mx = fmaxf(mx, z[j]);
  float se = 0.f;
  for (int j = 0; j < A; ++j) se += expf(z[j] - mx);
  for (int j = 0; j < A; ++j) {
    const float pj = expf(z[j] - mx) / se;
    mask[(long)b * A + j] = write_probs == 2 ? z[j] - mx - logf(se) : (write_probs ? pj : (pj > threshold ? 1.f : 0.f));
  }
}

// penalty only (compute_cql_penalty, cql_trainer.py:60-86): per-block partials of lse - ln A - Q[a]
__global__ __launch_bounds__(256) void cql_penalty_kernel(const float* __restrict__ Q, int ldq,
                                                           const int64_t* __restrict__ actions, int B, int A,
                                                           float log_A, float* __restrict__ part) {
  __shared__ float red[4];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  float pen = 0.f;
  if (b < B) {
    const float* q = Q + (size_t)b * ldq;
    float mx = -INFINITY;
    for (int j = 0; j < A; ++j) mx = fmaxf(mx, q[j]);
    float se = 0.f;
    for (int j = 0; j < A; ++j) se += expf(q[j] - mx);
    pen = mx + logf(se) - log_A - q[(int)actions[b]];
  }
  pen = wave_sum(pen);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pen;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------
// state2costmap (util/costmap.py:7-64): (B, n_ang + 2) lidar ranges + relative goal -> (B, 3, n_ang, n_dist)
// polar occupancy image.  One block per (angle row, sample), one thread per distance bin; the image is
// written directly in channel-major order (the reference builds (B, n_ang, n_dist, 3) and permutes).
//   * values > 8 count as 0 everywhere and are zeroed IN PLACE like the reference does (:17)
//   * channel 0: one-hot of int(range / d_inc) per beam, beams rolled by n_ang/2, bin 0 cleared
//   * all channels: 3-pixel cross at the goal's (angle bin, distance bin); index -1 wraps to the last bin
//     exactly as the reference's advanced indexing does (dist bin 0 lights bin n_dist-1)
// Ranges that would index past n_dist-1 make the reference raise; here they are dropped.
// ---------------------------------------------------------------------------------------------------
__global__ void costmap_kernel(float* __restrict__ state, long state_stride, int n_ang, int n_dist,
                               float dist_inc, float ang_inc, float deg_min, float deg_max, float dist_max,
                               float* __restrict__ out) {
  const int r = blockIdx.x, b = blockIdx.y;
  float* st = state + (long)b * state_stride;
  auto rd = [&](int i) { const float v = st[i]; return v > 8.f ? 0.f : v; };
  const int src = (r - n_ang / 2 + n_ang) % n_ang;              // torch.roll(idx, n_ang/2, 1)
  const long beam_bin = (long)(rd(src) / dist_inc);             // .to(torch.long): truncation
  const float gx = rd(n_ang), gy = rd(n_ang + 1);
  float deg = atan2f(gy, gx);
  deg = fminf(fmaxf(deg, deg_min), deg_max);
  const long deg_bin = (long)((deg + 3.14159265358979323846f) / ang_inc);
  const float cd = fminf(sqrtf(gx * gx + gy * gy), dist_max);
  const long dist_bin = (long)(cd / dist_inc);
  auto wrap = [](long i, int n) { return i < 0 ? i + n : i; };  // python negative-index semantics
  const bool row_is_deg = r == wrap(deg_bin, n_ang);
  const bool row_near_deg = row_is_deg || r == wrap(deg_bin - 1, n_ang) || r == wrap(deg_bin + 1, n_ang);
  const size_t plane = (size_t)n_ang * n_dist;
  float* o = out + (size_t)b * 3 * plane + (size_t)r * n_dist;
  for (int c = threadIdx.x; c < n_dist; c += blockDim.x) {
    const bool cross = (row_is_deg && (c == wrap(dist_bin - 1, n_dist) || c == wrap(dist_bin, n_dist) ||
                                       c == wrap(dist_bin + 1, n_dist))) ||
                       (row_near_deg && c == wrap(dist_bin, n_dist));
    const bool beam = c != 0 && (long)c == beam_bin;
    o[c] = (beam || cross) ? 1.f : 0.f;
    o[plane + c] = cross ? 1.f : 0.f;
    o[2 * plane + c] = cross ? 1.f : 0.f;
  }
}

// the in-place side effect: state[state > 8] = 0 (runs after costmap_kernel on the same stream)
__global__ void clamp_gt8_kernel(float* __restrict__ state, long state_stride, int cols, int rows) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float* p = state + (i / cols) * state_stride + (i % cols);
    if (*p > 8.f) *p = 0.f;
  }
}

// int64 copy with stride (actions hand-over)
__global__ void pack_i64_kernel(const int64_t* __restrict__ src, long stride, int n, int64_t* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[(long)i * stride];
}

}  // namespace porl
