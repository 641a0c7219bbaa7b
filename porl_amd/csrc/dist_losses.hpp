// Distributional-RL loss heads on the Q-network engine's general path:
//   * QR-DQN quantile-Huber loss   (reference src/porl/train/qr_dqn_trainer.py:97-205)
//   * C51 categorical projection + cross-entropy   (reference src/porl/train/c51_trainer.py:52-174)
// Each kernel takes the network outputs of one minibatch (online net on s, online / target net on s') and leaves
// dL/d(online output on s) for the engine's backward pass plus per-row loss terms.  One 64-lane wave per minibatch
// row; everything a row needs (N quantiles / atoms, A actions) is small, so a row lives in registers + a little LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace porl {

constexpr int DIST_MAX_N = 256;     // quantiles / atoms per action

// ---------------------------------------------------------------------------------------------------
// QR-DQN.  Z* are (B, ld) rows holding (A, N) quantile values.
//   a*      = argmax_a mean_j Z_online(s')[a, j]                         (Double-DQN style selection, :132-137)
//   T_i     = r + gamma * Z_target(s')[a*, i] * (1 - d)                   (:155-158)
//   u_ij    = T_i - theta_j,  theta = Z_online(s)[a_taken]                (:166)
//   rho_ij  = |tau_i - 1[u_ij < 0]| * Huber_kappa(u_ij)                   (:170-188; tau_i = (2i+1)/(2N) is indexed by the
//             TARGET quantile i — `self.tau.unsqueeze(-1)` broadcasts over dim 1 — exactly as the reference does)
//   loss    = mean over (b, i) of sum_j rho_ij                            (:205)
//   dL/dtheta_j = -(1/(B N)) sum_i |tau_i - 1[u_ij < 0]| * (|u_ij| <= kappa ? u_ij : kappa * sign(u_ij))
// ---------------------------------------------------------------------------------------------------
struct QrLossArgs {
  const float* z_cur; const float* z_next_online; const float* z_next_target; long ld;
  const int64_t* actions; const float* rew; const float* done;
  float* dz;            // (B, ld): gradient w.r.t. z_cur, zero outside the taken action's quantiles
  float* row_loss;      // (B,): sum_i sum_j rho_ij / N  (the batch mean is taken by the caller's reduce)
  int B, A, N;
  float gamma, kappa, inv_batch;
};

__global__ __launch_bounds__(256) void qr_loss_kernel(const QrLossArgs a) {
  __shared__ float sh_t[4][DIST_MAX_N];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  const int N = a.N, A = a.A;
  const float* zc = a.z_cur + (long)b * a.ld;
  const float* zo = a.z_next_online + (long)b * a.ld;
  const float* zt = a.z_next_target + (long)b * a.ld;
  float* dz = a.dz + (long)b * a.ld;
  // next action: first maximum of the per-action quantile means (torch.mean then argmax)
  int best = 0;
  float bestv = -INFINITY;
  for (int act = 0; act < A; ++act) {
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += zo[act * N + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float m = s / (float)N;
    if (m > bestv) { bestv = m; best = act; }
  }
  const float r = a.rew[b], nd = 1.f - a.done[b];
  float* T = sh_t[wave];
  for (int i = lane; i < N; i += 64) T[i] = r + a.gamma * zt[best * N + i] * nd;
  __builtin_amdgcn_wave_barrier();
  const int64_t at64 = a.actions[b];
  if (at64 < 0 || at64 >= A) {
    // an action outside [0, A): the reference's gather raises IndexError (qr_dqn_trainer.py:147).  No access is made with
    // it here: the row contributes no gradient and a NaN loss term, which the host turns into that IndexError
    for (int j = lane; j < a.ld; j += 64) dz[j] = 0.f;
    if (lane == 0) a.row_loss[b] = __builtin_nanf("");
    return;
  }
  const int at = (int)at64;
  for (int j = lane; j < a.ld; j += 64)                       // every element is written exactly once
    if (j < at * N || j >= at * N + N) dz[j] = 0.f;
  float loss = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float th = zc[at * N + j];
    float g = 0.f;
    for (int i = 0; i < N; ++i) {
      const float u = T[i] - th;
      const float tau = __fdiv_rn(2.f * (float)i + 1.f, 2.f * (float)N);     // (2i + 1) / (2N), one fp32 division like torch
      const float w = fabsf(tau - (u < 0.f ? 1.f : 0.f));
      const float au = fabsf(u);
      const bool quad = au <= a.kappa;
      loss += w * (quad ? 0.5f * u * u : a.kappa * (au - 0.5f * a.kappa));
      g += w * (quad ? u : (u > 0.f ? a.kappa : -a.kappa));
    }
    dz[at * N + j] = -g * a.inv_batch / (float)N;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
  if (lane == 0) a.row_loss[b] = loss / (float)N;
}

// ---------------------------------------------------------------------------------------------------
// IQN quantile-Huber loss (reference src/porl/train/iqn_trainer.py:136-149, `quantile_huber_loss`):
//   u_ij = target_j - current_i   (td_target.unsqueeze(1) - current.unsqueeze(2): i over the N' current quantiles,
//          j over the N'' target quantiles), rho_ij = |tau_i - 1[u_ij < 0]| * Huber_kappa(u_ij), tau = the sampled
//          fractions of the CURRENT quantiles; loss = mean over (b, i, j).
// Only this loss head is provided: upstream's IQNTrainer cannot run (it builds IQNNetwork with five positionals for a
// four-parameter constructor and calls a get_q_values the class does not define, iqn_trainer.py:58-64,89 vs
// iqn_network.py:10), so its learn() has no reference output to pin against.
// ---------------------------------------------------------------------------------------------------
struct IqnLossArgs {
  const float* cur; const float* target; const float* taus;   // (B, Np), (B, Npp), (B, Np)
  float* dcur;          // (B, Np)
  float* row_loss;      // (B,): mean over (i, j) of rho
  int B, Np, Npp;
  float kappa, inv_batch;
};

__global__ __launch_bounds__(256) void iqn_loss_kernel(const IqnLossArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  const float* T = a.target + (long)b * a.Npp;
  float loss = 0.f;
  for (int i = lane; i < a.Np; i += 64) {
    const float th = a.cur[(long)b * a.Np + i], tau = a.taus[(long)b * a.Np + i];
    float g = 0.f;
    for (int j = 0; j < a.Npp; ++j) {
      const float u = T[j] - th, au = fabsf(u);
      const float w = fabsf(tau - (u < 0.f ? 1.f : 0.f));
      const bool quad = au <= a.kappa;
      loss += w * (quad ? 0.5f * u * u : a.kappa * (au - 0.5f * a.kappa));
      g += w * (quad ? u : (u > 0.f ? a.kappa : -a.kappa));
    }
    a.dcur[(long)b * a.Np + i] = -g * a.inv_batch / ((float)a.Np * (float)a.Npp);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
  if (lane == 0) a.row_loss[b] = loss / ((float)a.Np * (float)a.Npp);
}

// ---------------------------------------------------------------------------------------------------
// C51.  logits* are (B, ld) rows holding (A, N) PRE-softmax outputs (the reference's network ends in log_softmax over
// the atoms, categorical_q_network.py:76-78; the engine's network stops at the Linear layer and the log_softmax lives
// here, on both sides).
//   p'      = softmax(logits_target(s'))                                   (:63-64)
//   a*      = argmax_a sum_n p'[a, n] z_n,  z_n = v_min + n * delta         (:67-70)
//   Tz_n    = clamp(r + gamma z_n (1 - d), v_min, v_max), b_n = (Tz_n - v_min) / delta, l = floor, u = ceil   (:78-96)
//   m       = projection of p'[a*] onto the support: l != u: m[l] += p (u - b), m[u] += p (b - l); l == u: m[l] += p
//             (three scatter_add_ passes in the reference, :120-137; here one pass per atom in ascending n)
//   loss    = -mean_b sum_n m_n log(clamp(exp(logp_n), 1e-8)),  logp = log_softmax(logits_online(s)[a_taken])   (:160-166)
//   dL/dlogit_k = -(1/B) [ c_k m_k - softmax_k sum_n c_n m_n ],  c_n = 1[exp(logp_n) >= 1e-8]  (clamp passes gradient
//             inside its range)
// ---------------------------------------------------------------------------------------------------
struct C51LossArgs {
  const float* logits_cur; const float* logits_next_target; long ld;
  const int64_t* actions; const float* rew; const float* done;
  const float* support; // (N,) atom values z_n — torch.linspace(v_min, v_max, N) as the reference builds it (:50)
  float* dlogits;       // (B, ld)
  float* row_loss;      // (B,)
  int B, A, N;
  float gamma, v_min, v_max, delta_z, inv_batch;
};

__global__ __launch_bounds__(256) void c51_loss_kernel(const C51LossArgs a) {
  __shared__ float sh_p[4][DIST_MAX_N];      // p'[a*] then reused
  __shared__ float sh_m[4][DIST_MAX_N];      // projected distribution
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  const int N = a.N, A = a.A;
  const float* lt = a.logits_next_target + (long)b * a.ld;
  const float* lc = a.logits_cur + (long)b * a.ld;
  float* dl = a.dlogits + (long)b * a.ld;
  float* P = sh_p[wave];
  float* M = sh_m[wave];
  auto wsum = [](float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  auto wmax = [](float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
  };
  // expected value of every action under the target net; keep the best action's probabilities
  int best = 0;
  float bestq = -INFINITY;
  for (int act = 0; act < A; ++act) {
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, lt[act * N + n]);
    mx = wmax(mx);
    float se = 0.f;
    for (int n = lane; n < N; n += 64) se += expf(lt[act * N + n] - mx);
    se = wsum(se);
    const float lse = mx + logf(se);
    float q = 0.f;
    for (int n = lane; n < N; n += 64) q += expf(lt[act * N + n] - lse) * a.support[n];
    q = wsum(q);
    if (q > bestq) { bestq = q; best = act; }
  }
  {
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, lt[best * N + n]);
    mx = wmax(mx);
    float se = 0.f;
    for (int n = lane; n < N; n += 64) se += expf(lt[best * N + n] - mx);
    se = wsum(se);
    const float lse = mx + logf(se);
    for (int n = lane; n < N; n += 64) { P[n] = expf(lt[best * N + n] - lse); M[n] = 0.f; }
  }
  __builtin_amdgcn_wave_barrier();
  // projection: one lane walks the atoms in ascending order (the order of a sequential scatter_add_)
  if (lane == 0) {
    const float r = a.rew[b], nd = 1.f - a.done[b];
    for (int n = 0; n < N; ++n) {
      // r + gamma * z * (1 - d) with the reference's rounding points (no fused multiply-add: b sits next to floor/ceil)
      float tz = __fadd_rn(r, __fmul_rn(__fmul_rn(a.gamma, a.support[n]), nd));
      tz = fminf(fmaxf(tz, a.v_min), a.v_max);
      const float bb = __fdiv_rn(__fsub_rn(tz, a.v_min), a.delta_z);
      const float fl = floorf(bb), ce = ceilf(bb);
      int l = (int)fl, u = (int)ce;
      const int lcl = min(max(l, 0), N - 1), ucl = min(max(u, 0), N - 1);
      if (l != u) {
        M[lcl] += P[n] * (ce - bb);
        M[ucl] += P[n] * (bb - fl);
      } else {
        M[lcl] += P[n];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int64_t at64 = a.actions[b];
  if (at64 < 0 || at64 >= A) {                                 // as in qr_loss_kernel: no access through a bad action index
    for (int j = lane; j < a.ld; j += 64) dl[j] = 0.f;
    if (lane == 0) a.row_loss[b] = __builtin_nanf("");
    return;
  }
  const int at = (int)at64;
  float mx = -INFINITY;
  for (int n = lane; n < N; n += 64) mx = fmaxf(mx, lc[at * N + n]);
  mx = wmax(mx);
  float se = 0.f;
  for (int n = lane; n < N; n += 64) se += expf(lc[at * N + n] - mx);
  se = wsum(se);
  const float lse = mx + logf(se);
  float loss = 0.f, cm = 0.f;
  for (int n = lane; n < N; n += 64) {
    const float logp = lc[at * N + n] - lse;
    const float p = expf(logp);
    const bool in = p >= 1e-8f;
    loss -= M[n] * logf(fmaxf(p, 1e-8f));
    if (in) cm += M[n];
  }
  loss = wsum(loss);
  cm = wsum(cm);
  for (int j = lane; j < a.ld; j += 64)                       // every element is written exactly once
    if (j < at * N || j >= at * N + N) dl[j] = 0.f;
  for (int n = lane; n < N; n += 64) {
    const float p = expf(lc[at * N + n] - lse);
    const float c = p >= 1e-8f ? 1.f : 0.f;
    dl[at * N + n] = -(c * M[n] - p * cm) * a.inv_batch;
  }
  if (lane == 0) a.row_loss[b] = loss;
}

}  // namespace porl
