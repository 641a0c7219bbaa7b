// Driver of tests/test_native_abi.py:test_host_planner_under_sanitizers — walks the REJECTED-argument paths and the
// pure host paths (handle creation, layout queries, planning) of the C ABI on a HOST-ONLY build of csrc/porl_api.hip
// compiled with -fsanitize=address,undefined.  No kernel is launched: every call below must fail validation before
// it reaches a launch (or is host-only by construction), so the program runs without a GPU.  Exit code 0 = every
// rejection was a clean error return and the sanitizers stayed silent (they abort the process otherwise).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/porl_hip.h"

static int g_checks = 0, g_bad = 0;
#define REJECT(expr)                                                                          \
  do {                                                                                        \
    ++g_checks;                                                                               \
    const long _r = (long)(expr);                                                             \
    if (_r == 0) { ++g_bad; std::fprintf(stderr, "accepted: %s\n", #expr); }                  \
    else if (!porl_last_error() || !porl_last_error()[0]) { ++g_bad; std::fprintf(stderr, "no message: %s\n", #expr); } \
  } while (0)
#define ACCEPT(expr)                                                                          \
  do {                                                                                        \
    ++g_checks;                                                                               \
    const long _r = (long)(expr);                                                             \
    if (_r != 0) { ++g_bad; std::fprintf(stderr, "rejected (%ld, %s): %s\n", _r, porl_last_error(), #expr); } \
  } while (0)

int main() {
  if (porl_abi_version() < 1) return 2;
  // ---- IQL engine ---------------------------------------------------------------------------------------------------
  porl_iql* h = nullptr;
  porl_iql_cfg c{60, 60, 64, 2, 0, 0, 0, 128};
  REJECT(porl_iql_create(nullptr, &h));
  REJECT(porl_iql_create(&c, nullptr));
  { porl_iql_cfg b = c; b.obs_dim = 0; REJECT(porl_iql_create(&b, &h)); }
  { porl_iql_cfg b = c; b.n_hidden = 0; REJECT(porl_iql_create(&b, &h)); }
  { porl_iql_cfg b = c; b.n_hidden = 1000; REJECT(porl_iql_create(&b, &h)); }
  { porl_iql_cfg b = c; b.pol_out_dim = 1 << 20; REJECT(porl_iql_create(&b, &h)); }
  { porl_iql_cfg b = c; b.layer_norm = 1; b.hidden_dim = 1 << 20; REJECT(porl_iql_create(&b, &h)); }
  { porl_iql_cfg b = c; b.max_batch = -3; REJECT(porl_iql_create(&b, &h)); }
  ACCEPT(porl_iql_create(&c, &h));
  const int64_t nv = porl_iql_group_floats(h, 0), np_ = porl_iql_group_floats(h, 1), nw = porl_iql_workspace_floats(h);
  if (nv <= 0 || np_ <= 0 || nw <= 0 || porl_iql_group_floats(nullptr, 0) != 0) ++g_bad;
  // every tensor of both groups lies inside its group and starts on a 16-byte boundary
  for (int g = 0; g < 2; ++g) {
    const int nt = porl_iql_group_tensors(h, g);
    int64_t off = 0; int32_t r = 0, cc = 0, prev_end = 0;
    for (int i = 0; i < nt; ++i) {
      ACCEPT(porl_iql_tensor_info(h, g, i, &off, &r, &cc));
      const int64_t n = (int64_t)(r ? r : 1) * cc;
      if (off % 4 || off < prev_end || off + n > (g == 0 ? nv : np_)) { ++g_bad; std::fprintf(stderr, "layout: group %d tensor %d\n", g, i); }
      prev_end = (int32_t)(off + n);
    }
    REJECT(porl_iql_tensor_info(h, g, nt, &off, &r, &cc));
    REJECT(porl_iql_tensor_info(h, g, -1, &off, &r, &cc));
  }
  REJECT(porl_iql_tensor_info(h, 2, 0, nullptr, nullptr, nullptr));
  float x[64] = {0};
  porl_iql_hyper hp{};
  hp.tau = 0.9f; hp.discount = 0.99f; hp.alpha = 10.f; hp.ema_beta = 0.005f; hp.inv_batch = 1.f / 32; hp.value_lr = 1e-4;
  hp.policy_lr = 1e-4; hp.value_step = 1; hp.policy_step = 1; hp.adam_beta1 = 0.9; hp.adam_beta2 = 0.999; hp.adam_eps = 1e-8;
  // unbound engine: every compute entry refuses
  REJECT(porl_iql_load_batch(h, 32, x, 60, x, 60, x, 1, x, 1, x, 60, nullptr));
  REJECT(porl_iql_value_backward(h, &hp, nullptr));
  REJECT(porl_iql_step(h, &hp, nullptr));
  REJECT(porl_iql_set_stats(h, x));
  REJECT(porl_iql_set_mode(nullptr, 0));
  REJECT(porl_iql_set_mode(h, 64));
  ACCEPT(porl_iql_set_mode(h, PORL_IQL_MODE_TWO_SLOTS | PORL_IQL_MODE_FOLD_COMBINE | PORL_IQL_MODE_SHORT_BLOCKS));
  ACCEPT(porl_iql_set_mode(h, 0));
  porl_iql_buffers bufs{};
  REJECT(porl_iql_bind(h, nullptr));
  REJECT(porl_iql_bind(h, &bufs));                                  // null buffers
  // host memory standing in for device buffers: bind only records the pointers (they are never dereferenced on the host)
  auto mk = [](int64_t n) { return static_cast<float*>(std::aligned_alloc(64, ((size_t)n * 4 + 63) / 64 * 64)); };
  float* P[11] = {mk(nv), mk(nv), mk(np_), mk(nv), mk(np_), mk(nv), mk(nv), mk(np_), mk(np_), mk(nw), mk(8)};
  bufs = porl_iql_buffers{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], P[9], P[10]};
  { porl_iql_buffers b2 = bufs; b2.grads_vf = P[3] + 1; REJECT(porl_iql_bind(h, &b2)); }     // misaligned
  ACCEPT(porl_iql_bind(h, &bufs));
  REJECT(porl_iql_load_batch(h, 0, x, 60, x, 60, x, 1, x, 1, x, 60, nullptr));
  REJECT(porl_iql_load_batch(h, 129, x, 60, x, 60, x, 1, x, 1, x, 60, nullptr));
  REJECT(porl_iql_load_batch(h, 32, nullptr, 60, x, 60, x, 1, x, 1, x, 60, nullptr));
  REJECT(porl_iql_load_batch_sampled(h, 0, x, 124, 1000, 2, 0, 1, 0, nullptr, nullptr));
  REJECT(porl_iql_load_batch_sampled(h, 32, nullptr, 124, 1000, 2, 0, 1, 0, nullptr, nullptr));
  REJECT(porl_iql_load_batch_sampled(h, 32, x, 124, 16, 2, 0, 1, 0, nullptr, nullptr));        // fewer rows than the batch
  REJECT(porl_iql_load_batch_sampled(h, 32, x, 100, 1000, 2, 0, 1, 0, nullptr, nullptr));       // row shorter than 2S+2+A
  REJECT(porl_iql_load_batch_sampled(h, 32, x, 124, (int64_t)1 << 41, 2, 0, 1, 0, nullptr, nullptr));
  REJECT(porl_iql_load_batch_sampled(h, 32, x, 124, 1000, 2, 1, 1, 0, nullptr, nullptr));       // target width mismatch
  REJECT(porl_iql_value_backward(h, &hp, nullptr));                 // no minibatch loaded
  REJECT(porl_iql_value_backward(h, nullptr, nullptr));
  REJECT(porl_iql_policy_backward(h, &hp, nullptr));
  REJECT(porl_iql_policy_forward(h, &hp, nullptr));
  REJECT(porl_iql_step(h, &hp, nullptr));
  REJECT(porl_iql_forward_value(h, 0, nullptr, 60, 4, x, x, nullptr));
  REJECT(porl_iql_forward_policy(h, x, 60, 100000, x, 60, nullptr));
  {
    void* s1 = reinterpret_cast<void*>(0x10); void* s2 = reinterpret_cast<void*>(0x20);
    REJECT(porl_iql_update_pipelined(h, &hp, 32, x, 124, 1000, 2, 0, 1, 0, nullptr, nullptr, nullptr, 1, 0, 0, 1, s1, s2));   // no signals
    REJECT(porl_iql_update_pipelined(h, &hp, 32, x, 124, 1000, 2, 0, 1, 0, x, x, x, 0, 0, 0, 1, s1, s2));                     // seq 0
    REJECT(porl_iql_update_pipelined(h, &hp, 32, x, 124, 1000, 2, 0, 1, 0, x, x, x, 1, 0, 0, 1, s1, s1));                     // one stream
    REJECT(porl_iql_update_pipelined(h, &hp, 32, x, 124, 1000, 2, 0, 1, 0, x, x, x, 1, 0, 0, 1, s1, s2));                     // mode bits missing
    ACCEPT(porl_iql_set_mode(h, PORL_IQL_MODE_TWO_SLOTS | PORL_IQL_MODE_FOLD_COMBINE | PORL_IQL_MODE_SHORT_BLOCKS));
    REJECT(porl_iql_update_pipelined(h, &hp, 1000, x, 124, 1000, 2, 0, 1, 0, x, x, x, 1, 0, 0, 1, s1, s2));                   // batch > max_batch
    REJECT(porl_iql_update_pipelined(h, &hp, 32, x, 124, (int64_t)1 << 41, 2, 0, 1, 0, x, x, x, 1, 0, 0, 1, s1, s2));         // n_rows > 2^40
    REJECT(porl_iql_update_pipelined(h, nullptr, 32, x, 124, 1000, 2, 0, 1, 0, x, x, x, 1, 0, 0, 1, s1, s2));
  }
  porl_iql_destroy(h);
  porl_iql_destroy(nullptr);
  // ---- stateless building blocks ------------------------------------------------------------------------------------------
  REJECT(porl_adam_ema(nullptr, x, x, x, nullptr, 16, 1e-3, 1, 0.9, 0.999, 1e-8, 0.0, nullptr));
  REJECT(porl_adam_ema(P[0] + 1, P[1], P[2], P[3], nullptr, 16, 1e-3, 1, 0.9, 0.999, 1e-8, 0.0, nullptr));   // misaligned
  REJECT(porl_adam_ema(P[0], P[1], P[2], P[3], nullptr, 18, 1e-3, 1, 0.9, 0.999, 1e-8, 0.0, nullptr));        // n % 4
  REJECT(porl_adam_ema(P[0], P[1], P[2], P[3], nullptr, 16, 1e-3, 0, 0.9, 0.999, 1e-8, 0.0, nullptr));        // step 0
  REJECT(porl_adam_ema(P[0], P[1], P[2], P[3], nullptr, -4, 1e-3, 1, 0.9, 0.999, 1e-8, 0.0, nullptr));
  ACCEPT(porl_adam_ema(P[0], P[1], P[2], P[3], nullptr, 0, 1e-3, 1, 0.9, 0.999, 1e-8, 0.0, nullptr));         // empty sweep
  REJECT(porl_ema(nullptr, P[0], 16, 0.005, nullptr));
  REJECT(porl_ema(P[0], P[1], 18, 0.005, nullptr));
  ACCEPT(porl_ema(P[0], P[1], 0, 0.005, nullptr));
  REJECT(porl_gather_rows(nullptr, 4, nullptr, 4, 4, x, 4, nullptr));
  REJECT(porl_sample_indices(10, 11, 1, 0, 0, reinterpret_cast<int64_t*>(P[0]), nullptr));
  REJECT(porl_sample_indices(10, 4, 1, 0, 0, nullptr, nullptr));
  REJECT(porl_epoch_indices(10, 8, 4, 1, 0, 0, reinterpret_cast<int64_t*>(P[0]), nullptr));
  REJECT(porl_state2costmap(nullptr, 362, 4, 360, 256, x, nullptr));
  REJECT(porl_state2costmap(x, 362, 70000, 360, 256, x, nullptr));
  REJECT(porl_softmax_mask(nullptr, 8, 4, 8, 0.1f, 0, x, nullptr));
  REJECT(porl_reduce_mean(nullptr, 4, x, nullptr));
  REJECT(porl_qr_loss(nullptr, x, x, 8, nullptr, x, x, 4, 2, 4, 0.99f, 1.f, x, x, nullptr));
  REJECT(porl_qr_loss(x, x, x, 8, reinterpret_cast<int64_t*>(P[0]), x, x, 4, 2, 100000, 0.99f, 1.f, x, x, nullptr));
  REJECT(porl_c51_loss(x, x, 8, reinterpret_cast<int64_t*>(P[0]), x, x, x, 4, 2, 100000, 0.99f, -1.f, 1.f, x, x, nullptr));
  REJECT(porl_iqn_quantile_huber(nullptr, x, x, 4, 4, 4, 1.f, x, x, nullptr));
  REJECT(porl_iqn_cos_embed(nullptr, 4, 8, x, nullptr));
  REJECT(porl_iqn_cos_embed(x, 0, 8, x, nullptr));
  REJECT(porl_iqn_cos_embed(x, (int64_t)1 << 41, 8, x, nullptr));
  REJECT(porl_iqn_hadamard(x, 4, x, 2, 2, 8, x, nullptr));                 // row stride below the width
  REJECT(porl_iqn_hadamard(x, 8, nullptr, 2, 2, 8, x, nullptr));
  REJECT(porl_iqn_hadamard_backward(x, x, 8, x, 2, 2, 8, nullptr, nullptr, nullptr));   // neither output
  REJECT(porl_iqn_hadamard_backward(x, x, 8, x, 0, 2, 8, x, x, nullptr));
  REJECT(porl_iqn_select(x, nullptr, 2, 2, 3, x, nullptr));
  REJECT(porl_iqn_select(x, reinterpret_cast<int64_t*>(P[0]), 2, 0, 3, x, nullptr));
  REJECT(porl_iqn_scatter(x, reinterpret_cast<int64_t*>(P[0]), 2, 2, 0, x, nullptr));
  REJECT(porl_iqn_target(x, x, x, nullptr, 0.9f, 2, 2, 3, x, nullptr, nullptr));
  REJECT(porl_iqn_target(x, x, x, x, 0.9f, -1, 2, 3, x, nullptr, nullptr));
  REJECT(porl_grad_clip(x, 8, 0.f, x, reinterpret_cast<double*>(P[0]), nullptr));      // max_norm must be positive
  REJECT(porl_grad_clip(x, -1, 1.f, x, reinterpret_cast<double*>(P[0]), nullptr));
  REJECT(porl_grad_clip(x, 8, 1.f, x, nullptr, nullptr));
  REJECT(porl_gemm_f32(7, -1, 8, 8, 8, x, 8, x, 8, x, 8, nullptr, 0, nullptr, 0, 1, nullptr, nullptr));
  REJECT(porl_gemm_f32(0, -1, 8, 8, 8, nullptr, 8, x, 8, x, 8, nullptr, 0, nullptr, 0, 1, nullptr, nullptr));
  REJECT(porl_tune_set(nullptr, 1));
  REJECT(porl_tune_set("no_such_key", 1));
  ACCEPT(porl_tune_set("skinny", 7));
  REJECT(porl_tune_set_ptr("no_such_key", nullptr));
  REJECT(porl_signal_write(nullptr, 1, nullptr));
  REJECT(porl_signal_wait_ge(nullptr, 1, nullptr));
  // ---- Q-network engine ---------------------------------------------------------------------------------------------------
  porl_qnet* q = nullptr;
  porl_qnet_cfg qc{};
  qc.state_dim = 60; qc.n_actions = 10; qc.n_hidden = 3; qc.hidden[0] = 64; qc.hidden[1] = 256; qc.hidden[2] = 64; qc.max_batch = 64;
  REJECT(porl_qnet_create(nullptr, &q));
  { porl_qnet_cfg b = qc; b.n_hidden = 0; REJECT(porl_qnet_create(&b, &q)); }
  { porl_qnet_cfg b = qc; b.n_actions = 0; REJECT(porl_qnet_create(&b, &q)); }
  { porl_qnet_cfg b = qc; b.hidden[1] = -5; REJECT(porl_qnet_create(&b, &q)); }
  ACCEPT(porl_qnet_create(&qc, &q));
  if (porl_qnet_one_launch(q) != 0) { ++g_bad; std::fprintf(stderr, "a 256-wide layer cannot be on the one-launch kernel\n"); }
  if (porl_qnet_param_floats(q) <= 0 || porl_qnet_workspace_floats(q) <= 0) ++g_bad;
  { int64_t off; int32_t r, cc, ld; REJECT(porl_qnet_tensor_info(q, 99, &off, &r, &cc, &ld)); ACCEPT(porl_qnet_tensor_info(q, 0, &off, &r, &cc, &ld)); }
  porl_qnet_hyper qh{0.99f, 1.f, 1.f / 64, 1, 5e-4, 0.9, 0.999, 1e-8};
  REJECT(porl_qnet_learn(q, &qh, nullptr));                         // unbound
  REJECT(porl_qnet_bind(q, nullptr));
  REJECT(porl_qnet_forward(q, 0, x, 60, 4, x, 10, nullptr));
  porl_qnet_destroy(q);
  // ---- costmap encoder ------------------------------------------------------------------------------------------------------
  porl_enc* e = nullptr;
  porl_enc_cfg ec{};
  ec.n_ang = 360; ec.n_dist = 256; ec.embed_dim = 96; ec.depth0 = 1; ec.depth1 = 2; ec.n_div = 4; ec.feature_dim = 1280;
  ec.num_classes = 256; ec.max_batch = 8; ec.mlp_ratio = 2.f; ec.bn_eps = 1e-5f; ec.bn_momentum = 0.1f; ec.bf16_operands = 2;
  REJECT(porl_enc_create(nullptr, &e));
  { porl_enc_cfg b = ec; b.n_ang = 361; REJECT(porl_enc_create(&b, &e)); }
  { porl_enc_cfg b = ec; b.embed_dim = 100; REJECT(porl_enc_create(&b, &e)); }
  { porl_enc_cfg b = ec; b.depth0 = 0; REJECT(porl_enc_create(&b, &e)); }
  { porl_enc_cfg b = ec; b.max_batch = 70000; REJECT(porl_enc_create(&b, &e)); }
  ACCEPT(porl_enc_create(&ec, &e));
  if (porl_enc_param_floats(e) != 1032960 + 0 && porl_enc_param_floats(e) < 1032960) { ++g_bad; std::fprintf(stderr, "encoder parameter count\n"); }
  if (porl_enc_tensors(e) <= 0 || porl_enc_norms(e) != 5 || porl_enc_blocks(e) != 3 || porl_enc_workspace_floats(e) <= 0) ++g_bad;
  { int64_t off, n; char name[128]; REJECT(porl_enc_tensor_info(e, 10000, &off, &n, name, 128)); ACCEPT(porl_enc_tensor_info(e, 0, &off, &n, name, 128)); }
  REJECT(porl_enc_forward(e, x, 362, 4, 1, nullptr, x, 256, nullptr));      // unbound
  REJECT(porl_enc_bind(e, nullptr, x, x));
  porl_enc_destroy(e);
  for (float* p : P) std::free(p);
  std::printf("abi_reject: %d checks, %d unexpected\n", g_checks, g_bad);
  return g_bad ? 1 : 0;
}
