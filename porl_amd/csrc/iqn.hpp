// Implicit Quantile Network pieces that are not plain Linear layers (gfx950 only).
//
// Reference: src/porl/net/iqn_network.py:35-91 (IQNNetwork.forward / get_quantile_embedding) and
// src/porl/train/iqn_trainer.py:92-134 (IQNTrainer.learn).  The Linear layers of the network run on the grouped fp32-MFMA
// GEMM (gemm_f32.hpp) and the quantile-Huber head on iqn_loss_kernel (dist_losses.hpp); what is here is the HBM-bound glue
// between them, each a single pass over its tensor with 16-byte accesses where the rows allow it:
//   * cosine features cos(pi * i * tau), i = 1..E                                     (iqn_network.py:74-91)
//   * the Hadamard product of state features (B, H) with quantile embeddings (B*N, H)   (iqn_network.py:58-62) + backward
//   * gather / scatter of the taken action's quantile values along the action axis     (iqn_trainer.py:101-103)
//   * Double-DQN action choice on the tau-mean of the online net + Bellman targets     (iqn_trainer.py:108-121)
//   * torch.nn.utils.clip_grad_norm_ on a flat gradient buffer                         (iqn_trainer.py:131)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace porl {

// out[r, i] = cos(pi * (i + 1) * taus[r]) in the reference's fp32 operation order: (fl32(pi) * idx) * tau, then cos
__global__ __launch_bounds__(256) void iqn_cos_embed_kernel(const float* __restrict__ taus, long n, int E,
                                                            float* __restrict__ out) {
  const long total = n * E;
  const float pi = 3.14159265358979323846f;
  for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < total; f += (long)gridDim.x * 256) {
    const long r = f / E;
    const int i = (int)(f - r * E);
    out[f] = cosf(__fmul_rn(__fmul_rn(pi, (float)(i + 1)), taus[r]));
  }
}

// out[(b*N + n), :] = feat[b, :] * emb[(b*N + n), :]
__global__ __launch_bounds__(256) void iqn_hadamard_kernel(const float* __restrict__ feat, long ldf,
                                                           const float* __restrict__ emb, int batch, int n_tau, int H,
                                                           float* __restrict__ out, int aligned16) {
  const long rows = (long)batch * n_tau;
  if (aligned16 && (H & 3) == 0 && (ldf & 3) == 0) {
    const int H4 = H >> 2;
    const long total = rows * H4;
    for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < total; f += (long)gridDim.x * 256) {
      const long r = f / H4;
      const int c = (int)(f - r * H4);
      const long b = r / n_tau;
      const float4 s = *reinterpret_cast<const float4*>(feat + b * ldf + 4 * c);
      const float4 e = *reinterpret_cast<const float4*>(emb + r * H + 4 * c);
      *reinterpret_cast<float4*>(out + r * H + 4 * c) = make_float4(s.x * e.x, s.y * e.y, s.z * e.z, s.w * e.w);
    }
  } else {
    const long total = rows * H;
    for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < total; f += (long)gridDim.x * 256) {
      const long r = f / H;
      const int c = (int)(f - r * H);
      out[f] = feat[(r / n_tau) * ldf + c] * emb[f];
    }
  }
}

// demb[(b*N + n), :] = dout[(b*N + n), :] * feat[b, :];   dfeat[b, :] = sum_n dout[(b*N + n), :] * emb[(b*N + n), :]
// (n ascending: the order of torch's sum over the expanded dimension is not specified; fp32, sequential).
// One thread per (b, column); either output may be null.
__global__ __launch_bounds__(256) void iqn_hadamard_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ feat,
                                                               long ldf, const float* __restrict__ emb, int batch, int n_tau,
                                                               int H, float* __restrict__ dfeat, float* __restrict__ demb) {
  const long total = (long)batch * H;
  for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < total; f += (long)gridDim.x * 256) {
    const long b = f / H;
    const int c = (int)(f - b * H);
    const float s = feat[b * ldf + c];
    float acc = 0.f;
    for (int n = 0; n < n_tau; ++n) {
      const long o = (b * n_tau + n) * H + c;
      const float d = dout[o];
      if (dfeat) acc = fmaf(d, emb[o], acc);
      if (demb) demb[o] = d * s;
    }
    if (dfeat) dfeat[f] = acc;
  }
}

// out[b, n] = z[b, n, actions[b]]   (an action outside 0..A-1 gives NaN: the reference's gather raises there)
__global__ __launch_bounds__(256) void iqn_select_kernel(const float* __restrict__ z, const int64_t* __restrict__ actions,
                                                         int batch, int n_tau, int A, float* __restrict__ out) {
  const long total = (long)batch * n_tau;
  for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < total; f += (long)gridDim.x * 256) {
    const int64_t a = actions[f / n_tau];
    out[f] = (a >= 0 && a < A) ? z[f * A + a] : __builtin_nanf("");
  }
}

// dz[b, n, a] = (a == actions[b]) ? dsel[b, n] : 0
__global__ __launch_bounds__(256) void iqn_scatter_kernel(const float* __restrict__ dsel, const int64_t* __restrict__ actions,
                                                          int batch, int n_tau, int A, float* __restrict__ dz) {
  const long total = (long)batch * n_tau * A;
  for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < total; f += (long)gridDim.x * 256) {
    const long r = f / A;
    const int a = (int)(f - r * A);
    dz[f] = (int64_t)a == actions[r / n_tau] ? dsel[r] : 0.f;
  }
}

// Per sample b: a* = argmax_a mean_n z_online[b, n, a] (first maximum, like torch.argmax; the mean is the fp32 sum over n
// ascending divided by N), td[b, n] = r[b] + gamma * z_target[b, n, a*] * (1 - done[b])        (iqn_trainer.py:108-121)
__global__ __launch_bounds__(256) void iqn_target_kernel(const float* __restrict__ z_online, const float* __restrict__ z_target,
                                                         const float* __restrict__ rewards, const float* __restrict__ dones,
                                                         float gamma, int batch, int n_tau, int A, float* __restrict__ td,
                                                         int64_t* __restrict__ next_actions) {
  for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < batch; b += (long)gridDim.x * 256) {
    int best = 0;
    float best_q = 0.f;
    for (int a = 0; a < A; ++a) {
      float s = 0.f;
      for (int n = 0; n < n_tau; ++n) s += z_online[(b * n_tau + n) * A + a];
      const float q = s / (float)n_tau;
      if (a == 0 || q > best_q) { best = a; best_q = q; }
    }
    if (next_actions) next_actions[b] = best;
    const float r = rewards[b], nd = 1.f - dones[b];
    for (int n = 0; n < n_tau; ++n)
      td[b * n_tau + n] = __fadd_rn(r, __fmul_rn(__fmul_rn(gamma, z_target[(b * n_tau + n) * A + best]), nd));
  }
}

// ---- clip_grad_norm_ --------------------------------------------------------------------------------------------------
// total = sqrt(sum g^2) (fp64 partial sums per block in a fixed order, one block adds the partials), coefficient
// min(1, max_norm / (total + 1e-6)) as torch forms it in fp32, applied in a second pass.
constexpr int CLIP_BLOCKS = 256;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
  __shared__ double red[256];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double v = (double)g[i];
    s += v * v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ partial, int nblocks, float max_norm,
                                                        float* __restrict__ norm_coef) {
  __shared__ double red[256];
  red[threadIdx.x] = (int)threadIdx.x < nblocks ? partial[threadIdx.x] : 0.0;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(red[0]);
    const float coef = max_norm / (total + 1e-6f);
    norm_coef[0] = total;
    norm_coef[1] = coef < 1.f ? coef : 1.f;
  }
}
__global__ __launch_bounds__(256) void scale_by_kernel(float* __restrict__ g, long n, const float* __restrict__ norm_coef) {
  const float c = norm_coef[1];
  if (c == 1.f) return;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] *= c;
}

}  // namespace porl
