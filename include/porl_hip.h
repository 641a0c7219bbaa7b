/*
 * porl_hip.h — C ABI of libporl_hip.so: the MI355X (gfx950) engine for porl's batched offline-RL
 * update step.  Plain pointers and sizes only; no torch types.  All device pointers are fp32 unless
 * noted; every call enqueues on `stream` (a hipStream_t passed as void*) and returns without
 * synchronising.  Return value: 0 on success, a negative PORL_ERR_* or a positive hipError_t.
 *
 * The reference (/root/reference, Python) has no FFI layer; each entry point below names the
 * reference Python interface it replaces, and INTEGRATION.md shows the ctypes stub a maintainer
 * would add on the reference side.
 */
#ifndef PORL_HIP_H
#define PORL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PORL_ABI_VERSION 5
#define PORL_MAX_HIDDEN 8

#define PORL_OK 0
#define PORL_ERR_INVALID (-1)      /* bad argument / shape */
#define PORL_ERR_UNSUPPORTED (-2)  /* configuration not implemented on device yet */
#define PORL_ERR_UNBOUND (-3)      /* engine used before porl_iql_bind() */

int porl_abi_version(void);
/* Human-readable text for the last failing call on this thread. */
const char* porl_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * IQL-family engine: POR (agent/por.py:20-112) and SORL (agent/sorl.py:20-152) share one value
 * step (twin-V expectile regression against an EMA target) and an advantage-weighted Gaussian
 * regression step; they differ in the regression target, the mean's output activation and the
 * weight formula.
 * --------------------------------------------------------------------------------------------- */
typedef struct porl_iql_cfg {
  int32_t obs_dim;      /* S: input width of vf / policy (args.state_size or args.feature_dim)       */
  int32_t pol_out_dim;  /* POR: S (goal policy predicts s', por.py:36-39); SORL: args.action_size     */
  int32_t hidden_dim;   /* args.hidden_dim                                                            */
  int32_t n_hidden;     /* args.n_hidden (>= 1)                                                       */
  int32_t layer_norm;   /* args.layer_norm: LayerNorm in TwinV only (value_functions.py:35-36)        */
  int32_t pol_tanh;     /* 1 = BoundedGaussianPolicy (policy.py:35-59), 0 = GaussianPolicy (:12-33)   */
  int32_t weight_mode;  /* 0 = exp(adv/alpha) (por.py:100), 1 = exp(alpha*adv) (sorl.py:104)          */
  int32_t max_batch;    /* workspace is sized for this many rows per call                             */
} porl_iql_cfg;

typedef struct porl_iql porl_iql;   /* opaque host-side handle; owns no device memory */

int porl_iql_create(const porl_iql_cfg* cfg, porl_iql** out);
void porl_iql_destroy(porl_iql* h);

/* Parameter groups.  Each group is ONE flat fp32 range so Adam/EMA/all-reduce are single sweeps;
 * tensors start on 16-byte boundaries, padding floats stay zero.
 *   group 0 "vf"     : TwinV  v1 then v2; per net, per Linear: weight (out,in), bias (out)
 *                       [, LayerNorm weight, bias after each hidden Linear when layer_norm]
 *   group 1 "policy" : log_std (pol_out_dim) first, then net Linear weights/biases
 * The EMA target network uses the layout of group 0.  Tensor order inside a group equals
 * nn.Module.named_parameters() order of the reference modules. */
int64_t porl_iql_group_floats(const porl_iql* h, int group);
int32_t porl_iql_group_tensors(const porl_iql* h, int group);
/* rows == 0 marks a 1-D tensor of `cols` elements. */
int porl_iql_tensor_info(const porl_iql* h, int group, int index, int64_t* offset, int32_t* rows,
                         int32_t* cols);

int64_t porl_iql_workspace_floats(const porl_iql* h);

typedef struct porl_iql_buffers {
  float* params_vf;   /* group 0 floats */
  float* params_tgt;  /* group 0 floats (v_target / v_tgt) */
  float* params_pol;  /* group 1 floats */
  float* grads_vf;
  float* grads_pol;
  float* adam_m_vf;   /* exp_avg    */
  float* adam_v_vf;   /* exp_avg_sq */
  float* adam_m_pol;
  float* adam_v_pol;
  float* workspace;   /* porl_iql_workspace_floats() floats, 16-byte aligned */
  float* stats;       /* >= 8 floats: [0] v_loss, [1] g_loss, [2] min NLL of the batch (por.py:104) */
} porl_iql_buffers;

int porl_iql_bind(porl_iql* h, const porl_iql_buffers* bufs);

/* Minibatch hand-over (replaces the tensor arguments of POR.por_residual_update, por.py:73, and
 * SORL.update, sorl.py:78).  Inputs may be strided column slices of one packed (B,row) tensor
 * (por_train.py:74-78): *_rs are row strides in floats, matrices have unit column stride, vectors
 * have element stride *_rs.  `pol_target` is s' for POR and the action matrix for SORL; it may be NULL
 * when only the value step is run.  Caller keeps ownership; inputs are not modified. */
int porl_iql_load_batch(porl_iql* h, int32_t batch,
                        const float* obs, int64_t obs_rs,
                        const float* next_obs, int64_t next_rs,
                        const float* rew, int64_t rew_rs,
                        const float* term, int64_t term_rs,
                        const float* pol_target, int64_t pt_rs,
                        void* stream);

/* Same hand-over, but the minibatch is DRAWN on the device from a resident packed-row replay store
 * (rows of [s(S) | r | s'(S) | d | a(A)], the wire format of por_train.py:74-78): row i of the batch is
 * store row perm_{seed,step}(i) for a keyed bijection perm of [0, n_rows) — `batch` distinct rows, i.e.
 * np.random.choice(size, B, replace=False) semantics (buffer/replay_buffer.py:64) without the O(N)
 * host permutation or any host->device copy.  One kernel does draw + gather + split.  The policy target
 * is s' (target_is_action = 0, POR) or the action columns (1, SORL).  idx_out (batch int64) may be NULL. */
int porl_iql_load_batch_sampled(porl_iql* h, int32_t batch, const float* rows, int64_t row_stride,
                                int64_t n_rows, int32_t act_dim, int32_t target_is_action,
                                uint64_t seed, uint64_t step, int64_t* idx_out, void* stream);

/* Execution mode of the phase calls (default 0).
 *   PORL_IQL_MODE_TWO_SLOTS   : the minibatch staging buffers the policy phase reads (s, policy target, TD target)
 *       exist PORL_IQL_SLOTS times and every porl_iql_load_batch* call moves to the next copy.  The policy phase of
 *       update t may then run on a second stream while updates t+1 .. t+PORL_IQL_SLOTS-1 are loaded and their value
 *       phases run (a load must only wait for the policy phase PORL_IQL_SLOTS updates back): the caller orders
 *       value_apply(t) -> policy_backward(t) and policy_apply(t) -> value_apply(t+1) with events (the policy phase
 *       reads the value parameters of update t; everything else it touches is private to it).  The reference runs
 *       the two phases back to back (agent/por.py:81-110); results are identical, only completion order differs.
 *   PORL_IQL_MODE_FOLD_COMBINE: *_backward leaves its split-K slabs / per-block partial sums uncombined and
 *       *_apply combines them inside the Adam launch (one launch less per phase).  grads_* are complete only after
 *       *_apply; a data-parallel caller that all-reduces grads_* between the two calls must not set it. */
#define PORL_IQL_SLOTS 3           /* copies of the staging buffers in PORL_IQL_MODE_TWO_SLOTS (name kept from ABI 3 drafts) */
#define PORL_IQL_MODE_TWO_SLOTS 1
#define PORL_IQL_MODE_FOLD_COMBINE 2
/*   PORL_IQL_MODE_SHORT_BLOCKS : the big products use 64x64 tiles (3 short blocks per CU) instead of 64x128 / 128x128.
 *       Alone on the chip they are ~10 % slower, but kernels of a second stream are only placed when blocks retire, so
 *       this is what lets the policy phase of a pipelined caller actually run beside the value phase. */
#define PORL_IQL_MODE_SHORT_BLOCKS 4
int porl_iql_set_mode(porl_iql* h, int32_t mode);

/* Redirect where the next updates write their 3 loss statistics (>= 8 floats, 16-byte aligned): lets a
 * training loop keep a device-side loss history without copies or host syncs. */
int porl_iql_set_stats(porl_iql* h, float* stats);

typedef struct porl_iql_hyper {
  float tau;         /* expectile                                   */
  float discount;    /* gamma                                       */
  float alpha;       /* advantage temperature                       */
  float inv_batch;   /* 1/B_global: a data-parallel shard passes 1/(world*B_local) */
  int32_t value_step;   /* Adam step counter t >= 1 of the value optimizer for this update  */
  int32_t policy_step;  /* ditto for the policy optimizer                                   */
  int32_t reserved;
  /* torch keeps these as Python doubles and rounds derived quantities (1-beta, lr/bias_correction) to
   * fp32 once; they are doubles here so that the same roundings happen */
  double ema_beta;   /* Polyak coefficient (por.py:31, beta=0.005)  */
  double value_lr;   /* constant (no schedule on the value optimizer) */
  double policy_lr;  /* CosineAnnealingLR value for THIS update (host-computed, Appendix A.3) */
  double adam_beta1, adam_beta2, adam_eps;   /* torch defaults 0.9, 0.999, 1e-8 */
} porl_iql_hyper;

/* por.py:81-89 — target-V forward, TD target, twin forward, expectile loss, backward.
 * Leaves dL/dtheta in grads_vf and stats[0] = this rank's share of v_loss.  (A data-parallel
 * caller all-reduces grads_vf and stats[0] here.) */
int porl_iql_value_backward(porl_iql* h, const porl_iql_hyper* hp, void* stream);
/* por.py:90-93 — Adam on vf, then v_target <- (1-beta) v_target + beta vf, one fused sweep. */
int porl_iql_value_apply(porl_iql* h, const porl_iql_hyper* hp, void* stream);
/* por.py:97-108 — second twin forward with the updated vf, advantage weights, policy forward,
 * weighted NLL, backward.  Leaves grads_pol, stats[1] = g_loss share, stats[2] = min NLL. */
/* optional, before porl_iql_policy_backward on the same batch: the policy MLP's forward (independent of the value
 * networks), e.g. while the value-gradient all-reduce is in flight; policy_backward then skips it */
int porl_iql_policy_prefetch(porl_iql* h, void* stream);
/* optional: the forward half of porl_iql_policy_backward on its own — second twin forward, policy forward, weights,
 * NLL and dL/dmean, i.e. every read of the VALUE parameters the policy phase makes.  A pipelined caller records its
 * "value parameters may change again" event right after it; porl_iql_policy_backward then only runs the rest. */
int porl_iql_policy_forward(porl_iql* h, const porl_iql_hyper* hp, void* stream);
int porl_iql_policy_backward(porl_iql* h, const porl_iql_hyper* hp, void* stream);
/* por.py:109 — Adam on the policy (lr = hp->policy_lr). */
int porl_iql_policy_apply(porl_iql* h, const porl_iql_hyper* hp, void* stream);
/* The four phases back to back (single-GPU update; batch must have been loaded). */
int porl_iql_step(porl_iql* h, const porl_iql_hyper* hp, void* stream);

/* Forward-only paths: TwinV.both (value_functions.py:38-39) on vf (which=0) or the target (which=1),
 * and the policy mean (policy.py:19 / sorl.py:71-76).  x is (batch, obs_dim) with row stride x_rs. */
int porl_iql_forward_value(porl_iql* h, int which, const float* x, int64_t x_rs, int32_t batch,
                           float* v1_out, float* v2_out, void* stream);
int porl_iql_forward_policy(porl_iql* h, const float* x, int64_t x_rs, int32_t batch, float* mean_out,
                            int64_t mean_rs, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Discrete-action Q-network engine: CQL(H) on a plain-DQN TD target
 * (CQLTrainer.learn / compute_cql_penalty, src/porl/train/cql_trainer.py:60-124; QNetwork,
 * src/porl/net/q_network.py:8-30; hard target sync, src/porl/train/dqn_trainer.py:195-196).
 * --------------------------------------------------------------------------------------------- */
typedef struct porl_qnet_cfg {
  int32_t state_dim;
  int32_t n_actions;                 /* <= 64 */
  int32_t n_hidden;                  /* QNetwork default: 3 */
  int32_t hidden[PORL_MAX_HIDDEN];   /* QNetwork default: 64, 128, 64 */
  int32_t max_batch;
} porl_qnet_cfg;

typedef struct porl_qnet porl_qnet;

typedef struct porl_qnet_buffers {
  float* params;      /* online net, porl_qnet_param_floats() floats: per Linear weight (out,in), bias (out) */
  float* params_tgt;  /* target net, same layout */
  float* grads;
  float* adam_m;
  float* adam_v;
  float* workspace;   /* porl_qnet_workspace_floats() floats */
  float* stats;       /* >= 8 floats: [0] loss, [1] td loss, [2] cql penalty */
} porl_qnet_buffers;

typedef struct porl_qnet_hyper {
  float gamma;        /* discount */
  float alpha;        /* penalty weight (cql_trainer.py:42, default 1) */
  float inv_batch;    /* 1/B_global */
  int32_t step;       /* Adam step counter t >= 1 */
  double lr;          /* dqn_trainer.py:71: 5e-4 */
  double adam_beta1, adam_beta2, adam_eps;
} porl_qnet_hyper;

int porl_qnet_create(const porl_qnet_cfg* cfg, porl_qnet** out);
void porl_qnet_destroy(porl_qnet* h);
/* Flat parameter group = per Linear layer an IMAGE of round32(out) rows x (round16(in) + 4) floats with the (out, in)
 * weight matrix in its top-left corner and zeros elsewhere, followed by round32(out) bias floats (zero padded): the
 * step kernel stages a layer into LDS with one linear copy.  tensor_info gives, for tensor `index` (weights and
 * biases alternate), its float offset, shape (rows == 0: a vector of `cols`) and the row stride of a weight matrix. */
int64_t porl_qnet_param_floats(const porl_qnet* h);
int32_t porl_qnet_tensors(const porl_qnet* h);
int porl_qnet_tensor_info(const porl_qnet* h, int index, int64_t* offset, int32_t* rows, int32_t* cols,
                          int32_t* row_stride);
int64_t porl_qnet_workspace_floats(const porl_qnet* h);
int porl_qnet_bind(porl_qnet* h, const porl_qnet_buffers* bufs);
/* The five tensors ReplayBuffer.sample returns (buffer/replay_buffer.py:53-75); *_rs = row / element
 * strides; actions are int64.  Any of actions / rewards / next_states / dones may be NULL for
 * forward-only use. */
int porl_qnet_load_batch(porl_qnet* h, int32_t batch, const float* states, int64_t s_rs,
                         const int64_t* actions, int64_t a_rs, const float* rewards, int64_t r_rs,
                         const float* next_states, int64_t n_rs, const float* dones, int64_t d_rs,
                         void* stream);
/* cql_trainer.py:94-111: both forwards, TD + penalty, backward -> grads, stats[0..2]. */
int porl_qnet_cql_backward(porl_qnet* h, const porl_qnet_hyper* hp, void* stream);
/* cql_trainer.py:112-113: Adam step. */
/* Building blocks for losses computed outside the engine (QR-DQN, C51 — porl_qr_loss / porl_c51_loss below): forward of
 * the online (which_params 0) / target (1) network on the loaded batch's states (which_input 0) / next states (1) into
 * out (batch, n_actions) [here n_actions = all outputs, e.g. actions x quantiles]; keep != 0 retains the activations
 * for porl_qnet_backward, which back-propagates a caller-supplied dL/d(output) and leaves the complete gradient in
 * `grads` (porl_qnet_apply then runs Adam). */
int porl_qnet_forward_loaded(porl_qnet* h, int which_params, int which_input, int keep, float* out, int64_t out_rs,
                             void* stream);
int porl_qnet_backward(porl_qnet* h, const float* dout, int64_t dout_rs, void* stream);

/* QR-DQN quantile-Huber loss (src/porl/train/qr_dqn_trainer.py:97-205): rows of (n_actions, n_quantiles) quantile values
 * with row stride ld; dz_out = dL/d(z_cur) for loss = mean_b row_loss[b] (1/batch already applied), row_loss (batch,). */
int porl_qr_loss(const float* z_cur, const float* z_next_online, const float* z_next_target, int64_t ld,
                 const int64_t* actions, const float* rewards, const float* dones, int32_t batch, int32_t n_actions,
                 int32_t n_quantiles, float gamma, float kappa, float* dz_out, float* row_loss, void* stream);
/* IQN quantile-Huber loss head only (src/porl/train/iqn_trainer.py:136-149): current (batch, n_current) quantile values
 * of the taken actions at fractions taus (batch, n_current), target (batch, n_target) Bellman targets; dcurrent_out =
 * dL/dcurrent for loss = mean_b row_loss[b].  (Upstream's IQNTrainer / IQNNetwork pair cannot run as shipped:
 * porl_amd/train/iqn_trainer.py states how learn() is read.) */
int porl_iqn_quantile_huber(const float* current, const float* target, const float* taus, int32_t batch, int32_t n_current,
                            int32_t n_target, float kappa, float* dcurrent_out, float* row_loss, void* stream);
/* The parts of an Implicit Quantile Network step that are not Linear layers (src/porl/net/iqn_network.py:35-91,
 * src/porl/train/iqn_trainer.py:92-134; csrc/iqn.hpp).  Row-major fp32, `n_tau` quantile fractions per sample.
 *   cos_embed        out (n, E): cos(pi * i * taus[r]), i = 1..E                                   iqn_network.py:74-91
 *   hadamard         out (batch*n_tau, H) = feat[b, :] (row stride ldf) * emb (batch*n_tau, H)     iqn_network.py:58-62
 *   hadamard_backward  dfeat (batch, H) = sum_n dout * emb, demb = dout * feat; either may be null
 *   select / scatter   out (batch, n_tau) = z[b, n, actions[b]] and its adjoint dz (batch, n_tau, A) iqn_trainer.py:101-103
 *   target           a* = argmax_a mean_n z_online_next[b, n, a]; td (batch, n_tau) = r + gamma * z_target_next[b, n, a*] *
 *                    (1 - done); next_actions (batch,) optional                                     iqn_trainer.py:108-121 */
int porl_iqn_cos_embed(const float* taus, int64_t n, int32_t embedding_dim, float* out, void* stream);
int porl_iqn_hadamard(const float* feat, int64_t ldf, const float* emb, int32_t batch, int32_t n_tau, int32_t width,
                      float* out, void* stream);
int porl_iqn_hadamard_backward(const float* dout, const float* feat, int64_t ldf, const float* emb, int32_t batch,
                               int32_t n_tau, int32_t width, float* dfeat, float* demb, void* stream);
int porl_iqn_select(const float* z, const int64_t* actions, int32_t batch, int32_t n_tau, int32_t n_actions, float* out,
                    void* stream);
int porl_iqn_scatter(const float* dsel, const int64_t* actions, int32_t batch, int32_t n_tau, int32_t n_actions, float* dz,
                     void* stream);
int porl_iqn_target(const float* z_online_next, const float* z_target_next, const float* rewards, const float* dones,
                    float gamma, int32_t batch, int32_t n_tau, int32_t n_actions, float* td, int64_t* next_actions,
                    void* stream);
/* torch.nn.utils.clip_grad_norm_(params, max_norm) on one flat gradient buffer (iqn_trainer.py:131): norm_coef[0] = the
 * total 2-norm, norm_coef[1] = min(1, max_norm / (norm + 1e-6)), grads scaled in place.  workspace: >= 256 doubles. */
int porl_grad_clip(float* grads, int64_t n, float max_norm, float* norm_coef, double* workspace, void* stream);
/* C51 projection + cross-entropy (src/porl/train/c51_trainer.py:52-174) on PRE-softmax outputs (the log_softmax of
 * categorical_q_network.py:76-78 is applied inside, to both networks' rows). */
int porl_c51_loss(const float* logits_cur, const float* logits_next_target, int64_t ld, const int64_t* actions,
                  const float* rewards, const float* dones, const float* support, int32_t batch, int32_t n_actions,
                  int32_t n_atoms, float gamma, float v_min, float v_max, float* dlogits_out, float* row_loss, void* stream);
/* out[0] = mean(x[0..n)) with a fixed summation order (loss reporting). */
int porl_reduce_mean(const float* x, int32_t n, float* out, void* stream);

int porl_qnet_apply(porl_qnet* h, const porl_qnet_hyper* hp, void* stream);
int porl_qnet_learn(porl_qnet* h, const porl_qnet_hyper* hp, void* stream);   /* the two above */
/* target_network.load_state_dict(q_network.state_dict()) */
/* 1 when the network fits the one-launch step kernel (porl_qnet_learn_indexed and the fast path of
 * porl_qnet_cql_backward), else 0 */
int32_t porl_qnet_one_launch(const porl_qnet* h);
/* learn() on the minibatch { row idx[b] (or b when idx is NULL) of the given replay arrays : b < batch } without
 * materialising it: the gather of ReplayBuffer.sample (buffer/replay_buffer.py:64-73) happens inside the step
 * kernel.  Only for networks the one-launch kernel covers (every layer <= 128 wide, <= 5 Linear layers);
 * PORL_ERR_UNSUPPORTED otherwise.  actions int64, dones fp32 (1.0 = terminal). */
int porl_qnet_learn_indexed(porl_qnet* h, const float* states, int64_t s_rs, const int64_t* actions,
                            const float* rewards, const float* next_states, int64_t n_rs, const float* dones,
                            const int64_t* idx, int32_t batch, const porl_qnet_hyper* hp, void* stream);
/* porl_qnet_learn_indexed with the loss of src/porl/train/dqn_per_trainer.py:75-123: Double-DQN target
 * (argmax by the online net, value by the target net), importance-sampling weights, |TD error| per sample for the
 * priority write-back.  Use hp->alpha = 0 for plain (un-regularised) DQN.  The reference multiplies a (B,1) weight
 * tensor with a (B,) error tensor, i.e. its loss is mean(w) * mean(td^2): pass that mean as `uniform_weight`
 * (device scalar) to reproduce it, or `is_weights` (B,) for per-sample weighting; either may be NULL. */
typedef struct porl_qnet_variant {
  int32_t double_dqn;
  const float* is_weights;
  const float* uniform_weight;
  float* td_abs;
  /* BCQ (src/porl/policy/bcq.py:59-74): (batch, n_actions) fp32 0/1 mask of the actions the behaviour policy allows in
   * s', row b for minibatch position b; the bootstrap action is argmax_a [Q_target(s',a) + (mask - 1) * 1e10] and is
   * valued by Q_target.  NULL = off. */
  const float* next_mask;
  /* 1: drop the TD term; with hyper.alpha = 1 the loss is the cross-entropy of softmax(Q(s)) against the taken
   * action minus ln(n_actions) — the behaviour-policy pre-training step of bcq.py:23-47 on a "Q" network that holds
   * the behaviour policy's logits. */
  int32_t td_off;
} porl_qnet_variant;
/* learn() on B distinct rows drawn INSIDE the step kernel: batch row b is row perm_{seed,draw}(b) of the n_rows-row replay
 * arrays, the keyed permutation of porl_sample_indices (same indices, no sampler launch, no index buffer).  Replaces
 * ReplayBuffer.sample + learn (buffer/replay_buffer.py:64, cql_trainer.py:88-124) for device-resident buffers.
 * porl_qnet_can_sample: 1 when the engine's network runs on the two-group one-launch kernel. */
int32_t porl_qnet_can_sample(const porl_qnet* h);
int porl_qnet_learn_sampled(porl_qnet* h, const float* states, int64_t s_rs, const int64_t* actions, const float* rewards,
                            const float* next_states, int64_t n_rs, const float* dones, int64_t n_rows, uint64_t seed,
                            uint64_t draw, int32_t batch, const porl_qnet_hyper* hp, void* stream);
int porl_qnet_learn_variant(porl_qnet* h, const float* states, int64_t s_rs, const int64_t* actions,
                            const float* rewards, const float* next_states, int64_t n_rs, const float* dones,
                            const int64_t* idx, int32_t batch, const porl_qnet_hyper* hp,
                            const porl_qnet_variant* variant, void* stream);
int porl_qnet_sync_target(porl_qnet* h, void* stream);
/* q_network(states) (which=0) or target_network(states) (which=1) -> (batch, n_actions) */
int porl_qnet_forward(porl_qnet* h, int which, const float* states, int64_t s_rs, int32_t batch,
                      float* q_out, int64_t q_rs, void* stream);
/* compute_cql_penalty(states, actions) -> out[0] (device float) */
int porl_qnet_penalty(porl_qnet* h, const float* states, int64_t s_rs, const int64_t* actions,
                      int64_t a_rs, int32_t batch, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Building blocks, exported for tests, the replay buffer and other trainers
 * --------------------------------------------------------------------------------------------- */
/* C = epilogue(op(A) * op(B)); mode 0 "NT": A (M,K), B (N,K); 1 "NN": A (M,K), B (K,N);
 * 2 "TN": A (K,M), B (K,N).  act: 0 none, 1 relu, 2 tanh.  bias (N) / mask (M,N; ld=ldmask) may be
 * NULL.  tile: 0 128x128, 1 128x64, 2 64x128, 3 64x64, -1 auto.  splitk > 1 needs `slab`
 * (splitk*M*ldc floats) and is combined in fixed order. */
int porl_gemm_f32(int mode, int tile, int32_t M, int32_t N, int32_t K,
                  const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                  const float* bias, int act, const float* mask, int32_t ldmask,
                  int splitk, float* slab, void* stream);

/* torch.optim.Adam single-tensor arithmetic over a flat range (n multiple of 4, 16-byte aligned),
 * optionally fused with target <- (1-ema_beta) target + ema_beta p (target may be NULL). */
int porl_adam_ema(float* p, const float* g, float* m, float* v, float* target, int64_t n,
                  double lr, int32_t step, double beta1, double beta2, double eps, double ema_beta,
                  void* stream);
/* F.softmax(logits, -1) > threshold as a 0/1 fp32 mask (src/porl/net/behavior_policy.py:41-55), or the probabilities
 * themselves when write_probs == 1 (:30-39), or log_softmax when write_probs == 2: logits (batch, ld) rows, n_actions
 * columns used; mask_out
 * (batch, n_actions) dense. */
int porl_softmax_mask(const float* logits, int64_t ld, int32_t batch, int32_t n_actions, float threshold,
                      int32_t write_probs, float* mask_out, void* stream);

/* util/util.py:54-56 on its own: target <- (1 - ema_beta) * target + ema_beta * source over n floats (n % 4 == 0).
 * Same rounding as the sweep fused into porl_adam_ema. */
int porl_ema(float* target, const float* source, int64_t n, double ema_beta, void* stream);

/* out[i,:] = rows[idx[i],:] — minibatch gather from a device-resident packed-row replay store
 * (replaces the numpy fancy-index + H2D copies of ReplayBuffer.sample, buffer/replay_buffer.py:64-73). */
int porl_gather_rows(const float* rows, int64_t row_stride, const int64_t* idx, int32_t n,
                     int32_t width, float* out, int64_t out_stride, void* stream);

/* `batch` distinct indices base + perm_{seed,step}(i), i < batch, perm a keyed bijection of [0, n_rows):
 * uniform sampling WITHOUT replacement on the device (semantics of np.random.choice(size, B,
 * replace=False), buffer/replay_buffer.py:64; the stream differs from numpy's). */
int porl_sample_indices(int64_t n_rows, int32_t batch, uint64_t seed, uint64_t step, int64_t base,
                        int64_t* out, void* stream);
/* Positions first .. first+count-1 of the keyed permutation (seed, epoch) of [0, n_rows): consecutive calls walk
 * one shuffled epoch — DataLoader(shuffle=True, drop_last=False) of por_train.py:61-63 without host work; the
 * last call of an epoch simply asks for fewer rows.  out[i] = base + perm(first + i), int64. */
int porl_epoch_indices(int64_t n_rows, int64_t first, int32_t count, uint64_t seed, uint64_t epoch, int64_t base,
                       int64_t* out, void* stream);

/* state2costmap (util/costmap.py:7-64; called by FasterNet.forward_cls, agent/fasternet.py:431):
 * (batch, n_ang + 2) lidar ranges + relative goal (x, y), row stride state_rs -> out (batch, 3, n_ang, n_dist)
 * contiguous.  The reference uses n_ang = 360, n_dist = 256.  Like the reference, entries > 8 of `state` are
 * zeroed in place. */
int porl_state2costmap(float* state, int64_t state_rs, int32_t batch, int32_t n_ang, int32_t n_dist,
                       float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Costmap encoder engine: FasterNet.forward_cls (agent/fasternet.py:428-438) as built by
 * sorl_train.py:29 `FasterNet(3, args.feature_dim)` — state2costmap, PatchEmbed 4x4s4 + BN (:234-246),
 * depth0 MLPBlocks (:141-194, Partial_conv3 :110-138), PatchMerging 2x2s2 + BN (:249-261), depth1
 * MLPBlocks, AdaptiveAvgPool2d(1) + 1x1 conv + ReLU (:367-371), Linear head (:372).  Forward only: the
 * reference keeps the backbone out of every optimizer (agent/sorl.py:58-64), so its backward never
 * changes a result.  BatchNorm follows nn.BatchNorm2d: batch statistics + running-stat update when
 * `training`, running statistics otherwise.
 * --------------------------------------------------------------------------------------------- */
#define PORL_ENC_MAX_BLOCKS 8
typedef struct porl_enc porl_enc;
typedef struct porl_enc_cfg {
  int32_t n_ang, n_dist;      /* costmap image height x width (reference: 360 x 256) */
  int32_t embed_dim;          /* 96 */
  int32_t depth0, depth1;     /* MLPBlocks per stage (reference depths=(1,2)) */
  int32_t n_div;              /* PConv acts on embed_dim/n_div channels (4) */
  int32_t feature_dim;        /* width of the pre-head 1x1 conv (1280) */
  int32_t num_classes;        /* output features per sample (sorl_train.py:29 -> 256) */
  int32_t max_batch;
  float mlp_ratio;            /* 2.0 */
  float bn_eps, bn_momentum;  /* nn.BatchNorm2d defaults 1e-5, 0.1 */
  /* 0 (default, the parity path): fp32 operands on the fp32-input MFMA.  1: the 1x1 / merge convolutions round their
   * operands to bf16 on the way into LDS and multiply on the bf16 matrix pipe, fp32 accumulate; activations, BatchNorm
   * statistics, the partial 3x3 conv, the patch embedding and the two head products stay fp32.  The reference has no
   * bf16 path (fp32 everywhere): results then differ from it by bf16 rounding (tests state the tolerance).
   * 2: bf16 ACTIVATIONS in HBM behind the patch embedding; partial 3x3 conv, MLP blocks (fused: one kernel per pass over
   * x) and the 2x2s2 merge on the bf16 matrix pipe, fp32 accumulate, fp32 BatchNorm statistics and head
   * (csrc/encoder_bf16.hpp).  Instantiated for the reference architecture (embed 96, mlp_ratio 2, n_div 4); other shapes
   * run mode 1. */
  int32_t bf16_operands;
} porl_enc_cfg;

int porl_enc_create(const porl_enc_cfg* cfg, porl_enc** out);
void porl_enc_destroy(porl_enc* h);
int64_t porl_enc_param_floats(const porl_enc* h);      /* flat parameter buffer, state_dict order + layouts */
int64_t porl_enc_stat_floats(const porl_enc* h);       /* BatchNorm running_mean / running_var buffer */
int64_t porl_enc_workspace_floats(const porl_enc* h);
int32_t porl_enc_tensors(const porl_enc* h);
int32_t porl_enc_norms(const porl_enc* h);
int32_t porl_enc_blocks(const porl_enc* h);            /* MLPBlocks = rows of drop_scale */
/* name = the reference's state_dict key ("stages.0.blocks.0.mlp.0.weight", ...) */
int porl_enc_tensor_info(const porl_enc* h, int32_t index, int64_t* offset, int64_t* numel, char* name,
                         int32_t name_len);
/* name = module prefix ("patch_embed.norm"); offsets into the stat buffer */
int porl_enc_norm_info(const porl_enc* h, int32_t index, int64_t* mean_offset, int64_t* var_offset,
                       int32_t* channels, char* name, int32_t name_len);
int porl_enc_bind(porl_enc* h, float* params, float* bn_stats, float* workspace);
/* The encoder keeps permuted copies of its convolution weights in the workspace and refreshes them on the first
 * forward after porl_enc_bind.  Call this after writing new values into `params` (load_state_dict, an optimizer step on
 * the backbone) so the next forward refreshes them again. */
int porl_enc_weights_changed(porl_enc* h);
/* state (batch, n_ang + 2) row stride state_rs, entries > 8 zeroed in place like the reference
 * (util/costmap.py:17); drop_scale (blocks, batch) = DropPath keep mask / keep_prob per MLPBlock and
 * sample (fasternet.py:76-93) or NULL for none; features (batch, num_classes), row stride feat_rs. */
int porl_enc_forward(porl_enc* h, float* state, int64_t state_rs, int32_t batch, int32_t training,
                     const float* drop_scale, float* features, int64_t feat_rs, void* stream);

/* Prioritized replay on the device (src/porl/buffer/sum_tree.py:4-77, prioritized_replay_buffer.py:36-108).
 * `tree`: 2*capacity-1 doubles in the reference's heap layout (leaf of data slot d at d + capacity - 1).
 * porl_per_update: SumTree.update for a batch (add() and update_priorities()): leaf <- (|td_error| + eps)^alpha,
 *   a leaf named twice keeps the last value, ancestors are recomputed from their children.  `stamp`: capacity
 *   zero-initialised int32 of scratch that the call leaves zeroed.
 * porl_per_sample: PrioritizedReplayBuffer.sample: segment i of the total priority, s = a + (b-a)*u[i] with the
 *   caller's uniforms u (random.random() of the reference's generator), tree walk -> out_idx (tree indices);
 *   out_prio needs 2*batch doubles (priorities, then scratch); out_w = (n_entries * p/total)^-beta / max. */
int porl_per_update(double* tree, int64_t capacity, const int64_t* tree_idx, const double* td_error, int32_t n,
                    double eps, double alpha, int32_t* stamp, void* stream);
int porl_per_sample(const double* tree, int64_t capacity, const double* u, int32_t batch, int64_t n_entries,
                    double beta, int64_t* out_idx, double* out_prio, float* out_w, void* stream);

/* Experiment knobs (scheduling only, never the mathematics).  "gemm_lds_pad": extra dynamic LDS bytes per GEMM
 * block, limiting how many blocks share a CU.  "qnet_fused": 0 forces the multi-launch CQL path. */
int porl_tune_set(const char* key, int value);
/* Diagnostics taking a device pointer.  "qnet_stamps": >= 32 uint64 receiving block 0's shader-clock stamps at
 * the phase boundaries of the one-launch CQL kernel (NULL switches it off). */
int porl_tune_set_ptr(const char* key, void* ptr);

/* Stream signals (no reference counterpart; used by the pipelined update of porl_amd/agent/_iql.py): a 64-bit counter
 * in device signal memory.  porl_signal_write enqueues "counter = value" on `stream` (after everything enqueued before
 * it), porl_signal_wait_ge holds `stream` until counter >= value.  Values must only grow. */
int porl_signal_create(void** out);
int porl_signal_destroy(void* sig);
int porl_signal_write(void* sig, uint64_t value, void* stream);
int porl_signal_wait_ge(void* sig, uint64_t value, void* stream);
/* One pipelined update (sampled minibatch) from one call: value phase on main_stream, policy phase on side_stream,
 * ordered by the three counters (value Adam done / policy forward half done / policy phase done) exactly as
 * porl_amd/agent/_iql.py issues the phase calls — reference agent/por.py:73-112 (one update), por_train.py:71-82 (the
 * loop that calls it).  seq = update number (>= 1, growing); wait_policy_seq / wait_fwd_seq = counter values the main
 * stream waits for before loading the batch / before the value Adam (0 = no wait); write_policy != 0 publishes seq on
 * sig_policy at the end.  The engine must be in PORL_IQL_MODE_TWO_SLOTS | PORL_IQL_MODE_FOLD_COMBINE mode. */
int porl_iql_update_pipelined(porl_iql* h, const porl_iql_hyper* hp, int32_t batch, const float* rows, int64_t row_stride,
                              int64_t n_rows, int32_t act_dim, int32_t target_is_action, uint64_t seed, uint64_t step,
                              void* sig_value, void* sig_fwd, void* sig_policy, uint64_t seq, uint64_t wait_policy_seq,
                              uint64_t wait_fwd_seq, int32_t write_policy, void* main_stream, void* side_stream);

/* Per-launch timing with HIP events on the launch stream (off by default; adds two event records per
 * kernel).  porl_prof_read synchronises the device and returns the number of entries filled. */
typedef struct porl_prof_entry {
  char name[96];
  int64_t launches;
  double total_ms;
  double flops;   /* algorithmic, summed over launches (GEMMs: 2*M*N*K per problem) */
  double bytes;   /* algorithmic operand + result bytes, summed over launches */
} porl_prof_entry;
int porl_prof_enable(int on);
int porl_prof_read(porl_prof_entry* out, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* PORL_HIP_H */
