// libporl_hip.so — C ABI (include/porl_hip.h) over the gfx950 kernels.  Host-side planning only:
// parameter/workspace layouts, the launch sequence of one update, and the group/tile/split-K choices
// that keep 256 CUs busy with 1024-wide MLPs.  No device allocation, no synchronisation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/porl_hip.h"
#include "gemm_f32.hpp"
#include "gemm_bf16.hpp"
#include "kernels.hpp"
#include "l0_fwd.hpp"
#include "skinny.hpp"
#include "iqn.hpp"
#include "qnet_fused.hpp"
#include "per_tree.hpp"
#include "dist_losses.hpp"

using namespace porl;

namespace {

thread_local std::string g_err;

#define PORL_FAIL(code, ...)                         \
  do {                                               \
    char _b[512];                                    \
    snprintf(_b, sizeof _b, __VA_ARGS__);            \
    g_err = _b;                                      \
    return (code);                                   \
  } while (0)

#define PORL_HIP(expr)                                                        \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      g_err = std::string(#expr) + ": " + hipGetErrorString(_e);              \
      return (int)_e;                                                         \
    }                                                                         \
  } while (0)

#define PORL_TRY(expr)          \
  do {                          \
    int _r = (expr);            \
    if (_r != 0) return _r;     \
  } while (0)

// Every entry point launches on the device that owns its buffers, whatever the calling thread's current device is:
// handles remember the ordinal of their workspace at bind time, stateless entry points look it up from a pointer.
struct DevGuard {
  int prev = -1;
  bool switched = false;
  explicit DevGuard(int dev) {
    if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
  DevGuard(const DevGuard&) = delete;
  DevGuard& operator=(const DevGuard&) = delete;
};
int device_of(const void* p) {
  hipPointerAttribute_t a;
  if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return a.type == hipMemoryTypeDevice ? a.device : -1;
}

// ---- optional per-launch timing with HIP events (bench.py's roofline leg) ---------------------------
struct ProfRec { int label; hipEvent_t e0, e1; double flops, bytes; };
struct ProfState {
  bool on = false;
  std::vector<std::string> labels;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  size_t pool_used = 0;
  hipEvent_t get() {
    if (pool_used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
    return pool[pool_used++];
  }
  int label_id(const std::string& s) {
    for (size_t i = 0; i < labels.size(); ++i) if (labels[i] == s) return (int)i;
    labels.push_back(s);
    return (int)labels.size() - 1;
  }
} g_prof;

const char* g_phase = "";     // prefix of the profile labels: which launch of the update step this is (profiling only)

struct ProfScope {
  bool active; ProfRec r; hipStream_t s;
  ProfScope(const std::string& label, hipStream_t st, double flops, double bytes) : active(g_prof.on), s(st) {
    if (!active) return;
    r.label = g_prof.label_id(std::string(g_phase) + label); r.flops = flops; r.bytes = bytes;
    r.e0 = g_prof.get(); r.e1 = g_prof.get();
    (void)hipEventRecord(r.e0, s);
  }
  ~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(r.e1, s);
    g_prof.recs.push_back(r);
  }
};

// porl_gemm_f32 diagnostics (scripts/bench_gemm_enc.py): the encoder's operand prologue / epilogue on a bare product
const float* g_dbg_a_scale = nullptr; const float* g_dbg_a_shift = nullptr; const float* g_dbg_resid = nullptr;
float* g_dbg_cstat = nullptr;
unsigned long long* g_qnet_stamps = nullptr;   // porl_tune_set_ptr("qnet_stamps", device buffer of >= 32 u64)
int g_enc_tile_n96 = 0;
int g_enc_gemm_sb = 1;        // porl_tune_set("enc_gemm_sb", 0): the encoder's 64x64 row products on the double-buffered schedule (A/B)
int g_enc_patch_rows = 0;     // porl_tune_set("enc_patch_rows", n): patch rows per block of patch_bn_kernel (0 = by grid size; A/B)
int g_enc_s2d = 0;           // porl_tune_set("enc_s2d", 1): materialise the 2x2 patches before the merge GEMM (cross-check)
// What pick_tile returns where its occupancy rule selects tile i (porl_tune_set("tile_map<i>", t) / "tile_map_short<i>").
// Measured on the POR step (gpurun_out/r02, bench.py PORL_TILE_MAP): with two co-resident 64x128 blocks per CU the
// 4 x 1024^3 launches take 71 instead of 75 us (each block's prologue / C store under the other's MFMA loop), and the
// 3-net forward / policy backward take 55 / 41 us on 64x64 tiles (768 / 512 blocks = exactly 3 / 2 per CU) instead of
// 69 / 45 on 128x64 (384 blocks: 1.5 per CU).  SHORT-BLOCK mode (PORL_IQL_MODE_SHORT_BLOCKS, the pipelined update):
// 64x64 everywhere — alone those launches are slower (77 / 84 us), but a second stream's kernels only get CUs when
// blocks retire, and 20 us blocks retire often: 2 990 -> 3 110 updates/s.
int g_tile_map[4] = {TILE_64x128, TILE_64x64, TILE_64x128, TILE_64x64};
int g_tile_map_short[4] = {TILE_64x64, TILE_64x64, TILE_64x128, TILE_64x64};
thread_local bool g_short_blocks = false;       // set by the IQL entry points from the handle's mode
int g_vbwd_tile_short = -1;   // porl_tune_set("vbwd_tile_short", t): tile of the value nets' hidden-layer backward in short-block mode (A/B)
int g_l0_kernel = 1;        // porl_tune_set("l0_kernel", 0): input layers (K <= 64) through the grouped GEMM instead of l0_fwd_kernel (A/B, bit-identical)
int g_l0_tile = -1;          // porl_tune_set("l0_tile", t): tile override for the K <= 128 forward layers of the IQL step (A/B)
// porl_tune_set("skinny", mask): which of the <= 64-wide products of the IQL step run on skinny.hpp instead of the grouped
// GEMM — bit 0: input-layer weight gradients dW0, bit 1: policy mean, bit 2: policy output-layer backward; 0 = round 2's
// path (A/B; same sums, other order inside a 64-chunk), 1 is read as "all" (7)
int g_skinny = 7;
// The same mask for the PIPELINED update (PORL_IQL_MODE_SHORT_BLOCKS), porl_tune_set("skinny_pipelined", mask).  Default 0:
// there the policy phase is not the critical stream, and shortening its launches moved the two queues against each other
// — same-box A/B (gpurun_out/r03/ab3.log, sustained updates/s, two runs each): mask 0: 3 306 / 3 309; 7: 3 206 / 3 204; 4:
// 3 233 / 3 238; 6: 3 221 / 3 218 — while the one-stream update gains 10-12 us (380 -> 369 us event-timed).
// Round 3, later: that holds for H = 1024, where the big products dominate.  At H = 256 (B = 1024) the update is a chain
// of short launches and the skinny kernels are worth +10 % pipelined as well (8 600 -> 9 530 updates/s), at H = 512
// +0.5 ... +5 % (6 570-6 860 -> 6 890-6 950), at H = 768 +2 % (4 490 -> 4 570), two runs each: -1 = by width — on up to
// `g_skinny_pipelined_max_h`, off above.
int g_skinny_pipelined = -1;
int g_skinny_pipelined_max_h = 768;
thread_local int g_cur_hidden = 0;              // hidden width of the engine whose entry point is running (check_ready)
inline int skinny_mask() {
  if (!g_short_blocks) return g_skinny;
  if (g_skinny_pipelined >= 0) return g_skinny_pipelined;
  return g_cur_hidden <= g_skinny_pipelined_max_h ? g_skinny : 0;
}
int g_dw0_slabs = 16;        // porl_tune_set("dw0_slabs", n <= SK_MAX): slabs of the skinny dW0 kernel (A/B: fewer slabs = fewer partial bytes, fewer blocks)
int g_iql_fold = 1;          // porl_tune_set("iql_fold", 0): porl_iql_step keeps the slab combines as launches of their own (A/B)
int g_enc_bf16_operands_only = 0;   // porl_tune_set("enc_bf16_operands_only", 1): compute_dtype="bf16" runs round 2's bf16-OPERAND mode (fp32 tensors in memory) instead of encoder_bf16.hpp (A/B)
int g_enc_pconv_mfma = 0;    // porl_tune_set("enc_pconv_mfma", 1): the fp32 partial 3x3 conv as an implicit GEMM on the matrix pipe (round 3) instead of the direct vector-ALU kernel.  Measured SLOWER (5.8 vs 3.7 ms per update at 360x256 B=512): the step moves ~2.5 GB per stage-1 call (input tile, result, the copy of the untouched channels) and is bound by that, not by the 184 GFLOP; the MFMA form parks more LDS per block (77 / 117 KB) and hides less latency.  Kept as a cross-check (tests/test_fasternet_gpu.py)
int g_enc_bn_sweep = 0;      // porl_tune_set("enc_bn_sweep", 1): BatchNorm + ReLU of the MLP blocks as a separate sweep (cross-check)
int g_enc_dense_patch = 0;   // porl_tune_set("enc_dense_patch", 1): rasterise + dense patch embedding (cross-check)
int g_qnet_fused = 1;     // porl_tune_set("qnet_fused", 0) forces the multi-launch CQL path (A/B measurements)
// 16 rows per block (v_mfma_f32_16x16x4_f32 tiles) while 32-row blocks would leave CUs idle (config 3 at B = 4096: 128 blocks
// on 256 CUs).  No faster in round 2; with round 3's loss stage on eight lanes per row it is: 22 800 -> 25 300 updates/s,
// step kernel 41.4 -> 34.6 us event-timed (gpurun_out/r03: same box, two runs each).  porl_tune_set("qnet_rows16", 0) = A/B.
int g_qnet_rows16 = 1;
int g_qnet_wgrad_share = 11;   // porl_tune_set("qnet_wgrad_share", s): sixteenths of a 32-tile layer's dW tiles the dW group keeps (16 = all: round 2's split)
int g_qnet_two_groups = 1;  // porl_tune_set("qnet_two_groups", 0): the one-group (256-thread) step kernel (A/B, bit-identical)

constexpr int NUM_CU = 256;
constexpr int SK_MAX = 16;
constexpr int NLL_ROWS_PER_BLOCK = 4;    // one row per wave
constexpr int LN_ROWS_PER_BLOCK = 8;
constexpr int HEAD_ROWS_PER_BLOCK = HEAD_ROWS;

inline int64_t ru4(int64_t x) { return (x + 3) & ~int64_t(3); }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

struct TensorInfo { int64_t off; int rows, cols; };

struct MlpLayout {
  int n_lin = 0;                       // Linear layers = n_hidden + 1
  int dims[PORL_MAX_HIDDEN + 2] = {0};
  int64_t w[PORL_MAX_HIDDEN + 1] = {0}, b[PORL_MAX_HIDDEN + 1] = {0};
  int64_t lnw[PORL_MAX_HIDDEN] = {0}, lnb[PORL_MAX_HIDDEN] = {0};
  bool ln = false;
};

int64_t add_tensor(std::vector<TensorInfo>& v, int64_t& cur, int rows, int cols) {
  const int64_t off = cur;
  v.push_back({off, rows, cols});
  cur += ru4((int64_t)(rows ? rows : 1) * cols);
  return off;
}

// named_parameters() order of util.mlp: Linear(w,b) [LayerNorm(w,b)] per hidden layer, final Linear
void layout_mlp(MlpLayout& m, std::vector<TensorInfo>& v, int64_t& cur, int in, int hid, int L, int out, bool ln) {
  m.n_lin = L + 1;
  m.ln = ln;
  m.dims[0] = in;
  for (int l = 0; l < L; ++l) m.dims[l + 1] = hid;
  m.dims[L + 1] = out;
  for (int l = 0; l <= L; ++l) {
    m.w[l] = add_tensor(v, cur, m.dims[l + 1], m.dims[l]);
    m.b[l] = add_tensor(v, cur, 0, m.dims[l + 1]);
    if (ln && l < L) {
      m.lnw[l] = add_tensor(v, cur, 0, hid);
      m.lnb[l] = add_tensor(v, cur, 0, hid);
    }
  }
}

struct Workspace {
  // xs / xt / target_v exist PORL_IQL_SLOTS times ("batch slots"): with PORL_IQL_MODE_TWO_SLOTS every load moves to the
  // next slot, so the policy phase of update t (on its own stream) can still read its minibatch while updates t+1 and
  // t+2 are being loaded
  int64_t xs_slot[PORL_IQL_SLOTS], xt_slot[PORL_IQL_SLOTS], target_v_slot[PORL_IQL_SLOTS];
  int64_t xn, rew, term;
  int64_t act_v[2][PORL_MAX_HIDDEN], act_t[2][2], act_p[PORL_MAX_HIDDEN];
  int64_t dz_v[2][2], dz_p[2];
  int64_t hp_t[2], hp_v[2];
  int64_t act_q[2][2], hp_q[2];      // policy phase only: scratch of the second twin forward and its head partials
  int64_t dv[2], dmu;
  int64_t slab_mean, slab_a, slab_b, slab_pa;   // slab_pa: the policy phase's own split-K slabs for dW0
  int64_t part_loss, part_min, part_dls;
  int64_t xhat_v[2][PORL_MAX_HIDDEN], rstd_v[2][PORL_MAX_HIDDEN];   // LayerNorm only
  int64_t ln_dg[2], ln_db[2], ln_dh[2];
  int64_t head_part[2], head_loss, head_db[2];                       // relu_head_bwd partials
  int64_t total;
};

}  // namespace

struct porl_iql {
  porl_iql_cfg cfg;
  MlpLayout v[2], pol;
  int64_t logstd_off = 0, n_vf = 0, n_pol = 0;
  std::vector<TensorInfo> t_vf, t_pol;
  porl_iql_buffers buf{};
  bool bound = false;
  int device = -1;                  // ordinal of the device that owns the bound buffers
  int mode = 0;                     // PORL_IQL_MODE_* bits
  int slot = 0;                     // batch slot of the most recent load
  ReduceArgs fin_v{}, fin_p{};      // PORL_IQL_MODE_FOLD_COMBINE: combines left pending by *_backward for *_apply
  bool fin_v_pending = false, fin_p_pending = false;
  int batch = 0;
  bool have_pol_target = false;
  bool pol_prefetched = false;      // porl_iql_policy_prefetch ran for the loaded batch: policy forward is done
  bool pol_fwd_done = false;        // porl_iql_policy_forward ran for the loaded batch
  int pol_nslab = 1;
  int Sp = 0, Dp = 0, Hp = 0, parts_max = 0;
  Workspace ws{};
};

namespace {

// ---- launch helpers -------------------------------------------------------------------------------
int pick_tile(const GemmGroup& g) {
  int minM = 1 << 30, minN = 1 << 30, maxK = 0;
  for (int i = 0; i < g.nprob; ++i) {
    minM = std::min(minM, g.p[i].M); minN = std::min(minN, g.p[i].N); maxK = std::max(maxK, g.p[i].K);
  }
  // a short K loop cannot hide its own prologue/epilogue: many small blocks per CU overlap them instead
  if (maxK <= 128) return TILE_64x64;
  auto blocks = [&](int tile) {
    int bm, bn, n = 0;
    tile_dims(tile, bm, bn);
    for (int i = 0; i < g.nprob; ++i) n += cdiv(g.p[i].M, bm) * cdiv(g.p[i].N, bn) * std::max(1, g.p[i].splitk);
    return n;
  };
  int cand[3], nc = 0;
  if (minN <= 64 && minM <= 64) { cand[nc++] = TILE_64x64; }
  else if (minN <= 64) { cand[nc++] = TILE_128x64; cand[nc++] = TILE_64x64; }
  else if (minM <= 64) { cand[nc++] = TILE_64x128; cand[nc++] = TILE_64x64; }
  else { cand[nc++] = TILE_128x128; cand[nc++] = TILE_128x64; cand[nc++] = TILE_64x64; }
  for (int i = 0; i < nc; ++i)
    if (blocks(cand[i]) >= NUM_CU) return (g_short_blocks ? g_tile_map_short : g_tile_map)[cand[i]];
  return (g_short_blocks ? g_tile_map_short : g_tile_map)[cand[nc - 1]];
}

// split-K factor for an output too small to fill the chip on its own
int pick_splitk(int M, int N, int K, int nprob, int bm, int bn) {
  const int tiles = nprob * cdiv(M, bm) * cdiv(N, bn);
  int sk = cdiv(NUM_CU, tiles);
  sk = std::min(sk, std::max(1, K / 64));
  return std::max(1, std::min(sk, SK_MAX));
}

// Short-block mode (the pipelined update): how much dynamic LDS a GEMM block of the value ('V') / policy ('P') phase
// claims on top of its 36 KB of staging buffers, and from how many blocks per launch on.  18 KB -> 55 KB per block ->
// TWO 64x64 blocks per CU instead of three.  Two effects, both measured (bench.py PORL_IQL_PAD="value,policy,min_blocks",
// PORL_GEMM_LDS_PAD; updates/s): (1) wave quantisation — the value nets' 4 x 1024^3 launches have 1 024 blocks: at three
// per CU that is one full round of 768 and a ragged one of 256, at two per CU two full rounds (backward launch alone on
// the chip: 84.6 -> 73.4 us; the 768-block 3-net forward gets WORSE at two per CU, 55.1 -> 59.1 us); (2) room for the
// other stream — a third of every CU's registers and 50 KB of LDS stay free for the policy stream's small kernels
// (with a 45 KB pad, i.e. two blocks and NO LDS left over, the update is slower than unpadded: 3 044 vs 3 106).
// none 3 106; every GEMM launch 3 256; value phase only 3 248; policy phase only 3 065; only launches of >= 1 000 blocks
// 3 270.  Default: the value phase's launches of at least 4 blocks per CU.
int g_iql_pad_value = 18432, g_iql_pad_policy = 0, g_iql_pad_min_blocks = 4 * NUM_CU, g_iql_pad_min_k = 0;
struct PadScope {
  int saved, saved_min;
  explicit PadScope(const GemmGroup& g) : saved(gemm_lds_pad()), saved_min(gemm_lds_pad_min_blocks()) {
    if (g_short_blocks && saved == 0 && g.nprob > 0 && g.p[0].K >= g_iql_pad_min_k) {
      gemm_lds_pad() = g_phase[0] == 'P' ? g_iql_pad_policy : g_iql_pad_value;
      gemm_lds_pad_min_blocks() = g_iql_pad_min_blocks;
    }
  }
  ~PadScope() { gemm_lds_pad() = saved; gemm_lds_pad_min_blocks() = saved_min; }
};

int launch_group(GemmGroup& g, int tile, hipStream_t s) {
  PadScope _pad(g);
  double flops = 0.0, bytes = 0.0;
  std::string label;
  if (g_prof.on) {
    bool vec = true, apro = false;
    for (int i = 0; i < g.nprob; ++i) {
      const GemmProb& p = g.p[i];
      flops += 2.0 * p.M * p.N * p.K;
      bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (p.store_c ? (double)p.M * p.N : 0.0));
      vec = vec && p.a_vec && p.b_vec;
      apro = apro || p.apro != APRO_NONE;
    }
    int bm, bn;
    tile_dims(tile, bm, bn);
    const TileCfg tc = tile_cfg(tile);
    label = "gemm_f32_kernel<" + std::to_string(bm) + "," + std::to_string(bn) + "," + std::to_string(GEMM_BK) + "," +
            std::to_string(tc.wm) + "," + std::to_string(tc.wn) + "," + (vec ? "true" : "false") + "," +
            (apro ? "true" : "false") + (g.single_buffer && vec && tile == TILE_64x64 ? ",true>" : ">");   // ONEBUF
  }
  ProfScope ps(label, s, flops, bytes);
  hipError_t e = launch_gemm_group(tile, g, s);
  if (e != hipSuccess) {
    g_err = std::string("gemm launch: ") + hipGetErrorString(e);
    return (int)e;
  }
  return 0;
}

// bf16-operand / fp32-accumulate variant (gemm_bf16.hpp): the costmap encoder's opt-in mode
int launch_group_bf16(GemmGroup& g, int tile, hipStream_t s) {
  double flops = 0.0, bytes = 0.0;
  std::string label;
  if (g_prof.on) {
    for (int i = 0; i < g.nprob; ++i) {
      const GemmProb& p = g.p[i];
      flops += 2.0 * p.M * p.N * p.K;
      bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N * (p.resid ? 2.0 : 1.0));
    }
    int bm, bn;
    tile_dims(tile, bm, bn);
    label = "gemm_bf16_kernel<" + std::to_string(bm) + "," + std::to_string(bn) + "," +
            (g.p[0].apro != APRO_NONE ? "true" : "false") + ">";
  }
  ProfScope ps(label, s, flops, bytes);
  hipError_t e = launch_gemm_bf16_group(tile, g, s);
  if (e != hipSuccess) {
    g_err = std::string("bf16 gemm launch: ") + hipGetErrorString(e);
    return (int)e;
  }
  return 0;
}

int launch_reduce(ReduceArgs& r, hipStream_t s) {
  if (r.njobs == 0) return 0;
  long maxn = 0;
  for (int i = 0; i < r.njobs; ++i) maxn = std::max(maxn, r.job[i].n);
  dim3 grid((unsigned)std::min<long>((maxn + 31) / 32, 2048), r.njobs);
  ProfScope ps("multi_reduce_kernel", s, 0.0, 0.0);
  hipLaunchKernelGGL(multi_reduce_kernel, grid, dim3(256), 0, s, r);
  PORL_HIP(hipGetLastError());
  return 0;
}

void add_reduce(ReduceArgs& r, float* out, const float* slab, long n, long stride, int nslab, int op = 0,
                float scale = 1.f) {
  ReduceJob& j = r.job[r.njobs++];
  j.out = out; j.slab = slab; j.bias = nullptr; j.n = n; j.stride = stride; j.nslab = nslab; j.ncols = 1; j.act = 0;
  j.op = op; j.scale = scale; j.adam_off = -1;
  j.width = nslab > 64 ? 4 : 32;        // long lists of partials: 64 slab lanes per output instead of 8
}

int check_ready(const porl_iql* h, bool need_batch) {
  if (!h) PORL_FAIL(PORL_ERR_INVALID, "null engine");
  if (!h->bound) PORL_FAIL(PORL_ERR_UNBOUND, "porl_iql_bind() has not been called");
  if (need_batch && h->batch <= 0) PORL_FAIL(PORL_ERR_INVALID, "no minibatch loaded (porl_iql_load_batch)");
  g_short_blocks = (h->mode & PORL_IQL_MODE_SHORT_BLOCKS) != 0;
  g_cur_hidden = h->cfg.hidden_dim;
  return 0;
}

// One hidden layer of up to 4 MLPs as one grouped launch (+ one LayerNorm launch for the nets that have it).
struct FwdNet {
  const float* in; int ldin;       // (B, K)
  const float* W; const float* b;  // (H, K), (H)
  float* out;                      // (B, Hp); may be null without LayerNorm when only the head is needed
  const float* headw; float* headout;
  const float* ln_g = nullptr; const float* ln_b = nullptr;   // LayerNorm affine (null = no LayerNorm)
  float* xhat = nullptr; float* rstd = nullptr;              // kept for backward when non-null
};

int fwd_hidden_layer(porl_iql* h, const FwdNet* nets, int nnets, int B, int K, bool last, int* parts_out,
                     hipStream_t s) {
  const int H = h->cfg.hidden_dim;
  // input layer (K = obs_dim <= 64), plain Linear + ReLU with a stored output: its own one-round kernel (l0_fwd.hpp)
  if (g_l0_kernel && K <= L0_KP && nnets <= L0_MAX_NETS) {
    L0Args a{};
    a.nnets = nnets; a.B = B; a.H = H; a.K = K; a.ldw = K; a.ldo = h->Hp;
    bool plain = true;
    for (int n = 0; n < nnets; ++n) {
      plain = plain && !nets[n].ln_g && nets[n].out && !(last && nets[n].headw);
      a.net[n] = L0Net{nets[n].in, nets[n].W, nets[n].b, nets[n].out, nets[n].ldin};
    }
    if (plain && l0_fwd_supported(a)) {
      if (parts_out) *parts_out = head_parts(H, 0);
      ProfScope ps("l0_fwd_kernel", s, 2.0 * nnets * B * (double)H * K, 4.0 * nnets * ((double)B * K + (double)H * K + (double)B * H));
      PORL_HIP(launch_l0_fwd(a, s));
      return 0;
    }
  }
  GemmGroup g{};
  g.nprob = nnets;
  bool any_ln = false;
  for (int n = 0; n < nnets; ++n) {
    const bool ln = nets[n].ln_g != nullptr;
    any_ln = any_ln || ln;
    GemmProb p = make_prob(GEMM_NT, nets[n].in, nets[n].ldin, nets[n].W, K, nets[n].out, h->Hp, B, H, K);
    p.bias = nets[n].b;
    p.act = ln ? ACT_NONE : ACT_RELU;
    p.store_c = nets[n].out != nullptr;
    if (ln && !nets[n].out) PORL_FAIL(PORL_ERR_INVALID, "LayerNorm needs a pre-activation buffer");
    if (!ln && last && nets[n].headw) { p.headw = nets[n].headw; p.headout = nets[n].headout; }
    g.p[n] = p;
  }
  const int tile = (K <= 128 && g_l0_tile >= 0) ? g_l0_tile : pick_tile(g);
  if (parts_out) *parts_out = any_ln ? 1 : head_parts(H, tile);
  PORL_TRY(launch_group(g, tile, s));
  if (any_ln) {
    LnFwdArgs a{};
    for (int n = 0; n < nnets; ++n) {
      if (!nets[n].ln_g) continue;
      const int k = a.nnets++;
      a.Z[k] = nets[n].out; a.xhat[k] = nets[n].xhat; a.rstd[k] = nets[n].rstd;
      a.gamma[k] = nets[n].ln_g; a.beta[k] = nets[n].ln_b;
      a.headw[k] = last ? nets[n].headw : nullptr; a.headb[k] = nullptr; a.headout[k] = nets[n].headout;
    }
    a.B = B; a.H = H; a.ld = h->Hp; a.rows_per_block = LN_ROWS_PER_BLOCK;
    hipLaunchKernelGGL(ln_relu_fwd_kernel, dim3(cdiv(B, LN_ROWS_PER_BLOCK), a.nnets), dim3(256), 0, s, a);
    PORL_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace

// =====================================================================================================
extern "C" {

int porl_abi_version(void) { return PORL_ABI_VERSION; }
const char* porl_last_error(void) { return g_err.c_str(); }

int porl_iql_create(const porl_iql_cfg* c, porl_iql** out) {
  if (!c || !out) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  if (c->obs_dim < 1 || c->pol_out_dim < 1 || c->hidden_dim < 1 || c->max_batch < 1)
    PORL_FAIL(PORL_ERR_INVALID, "dimensions must be positive");
  if (c->n_hidden < 1 || c->n_hidden > PORL_MAX_HIDDEN) PORL_FAIL(PORL_ERR_INVALID, "n_hidden must be in [1,%d]", PORL_MAX_HIDDEN);
  if (c->pol_out_dim > 64 * NLL_MAX_COLS_PER_LANE) PORL_FAIL(PORL_ERR_UNSUPPORTED, "pol_out_dim > %d", 64 * NLL_MAX_COLS_PER_LANE);
  if (c->layer_norm && c->hidden_dim > 256 * LN_MAX_COLS)
    PORL_FAIL(PORL_ERR_UNSUPPORTED, "layer_norm with hidden_dim > %d", 256 * LN_MAX_COLS);
  porl_iql* h = new porl_iql();
  h->cfg = *c;
  const int S = c->obs_dim, D = c->pol_out_dim, H = c->hidden_dim, L = c->n_hidden, B = c->max_batch;
  int64_t cur = 0;
  layout_mlp(h->v[0], h->t_vf, cur, S, H, L, 1, c->layer_norm != 0);
  layout_mlp(h->v[1], h->t_vf, cur, S, H, L, 1, c->layer_norm != 0);
  h->n_vf = cur;
  cur = 0;
  h->logstd_off = add_tensor(h->t_pol, cur, 0, D);
  layout_mlp(h->pol, h->t_pol, cur, S, H, L, D, false);
  h->n_pol = cur;

  h->Sp = (int)ru4(S); h->Dp = (int)ru4(D); h->Hp = (int)ru4(H);
  h->parts_max = 0;
  for (int tile = 0; tile < TILE_COUNT; ++tile) h->parts_max = std::max(h->parts_max, head_parts(H, tile));
  Workspace& w = h->ws;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += ru4(n); return r; };
  const int64_t BH = (int64_t)B * h->Hp;
  for (int k = 0; k < PORL_IQL_SLOTS; ++k) {
    w.xs_slot[k] = take((int64_t)B * h->Sp); w.xt_slot[k] = take((int64_t)B * h->Dp); w.target_v_slot[k] = take(B);
  }
  w.xn = take((int64_t)B * h->Sp);
  w.rew = take(B); w.term = take(B);
  for (int i = 0; i < 2; ++i) {
    for (int l = 0; l < L; ++l) w.act_v[i][l] = take(BH);
    w.act_t[i][0] = take(BH); w.act_t[i][1] = take(BH);
    w.act_q[i][0] = take(BH); w.act_q[i][1] = take(BH);
    w.hp_q[i] = take((int64_t)h->parts_max * B);
    w.dz_v[i][0] = take(BH); w.dz_v[i][1] = take(BH);
    w.hp_t[i] = take((int64_t)h->parts_max * B); w.hp_v[i] = take((int64_t)h->parts_max * B);
    w.dv[i] = take(B);
  }
  for (int l = 0; l < L; ++l) w.act_p[l] = take(BH);
  w.dz_p[0] = take(BH); w.dz_p[1] = take(BH);
  w.dmu = take((int64_t)B * h->Dp);
  w.slab_mean = take((int64_t)SK_MAX * B * h->Dp);
  w.slab_a = take((int64_t)SK_MAX * 2 * ((int64_t)H * S + 2 * H + 8));
  w.slab_b = take((int64_t)SK_MAX * ((int64_t)D * H + D + 8));
  w.slab_pa = take((int64_t)SK_MAX * ((int64_t)H * S + 2 * H + 8));
  const int nblk = cdiv(B, NLL_ROWS_PER_BLOCK);
  w.part_loss = take(nblk); w.part_min = take(nblk); w.part_dls = take((int64_t)nblk * D);
  for (int i = 0; i < 2; ++i) { w.head_part[i] = take((int64_t)cdiv(B, HEAD_ROWS_PER_BLOCK) * H); w.head_db[i] = take(cdiv(B, HEAD_ROWS_PER_BLOCK)); }
  w.head_loss = take(cdiv(B, HEAD_ROWS_PER_BLOCK));
  if (c->layer_norm) {
    const int nln = cdiv(B, LN_ROWS_PER_BLOCK);
    for (int i = 0; i < 2; ++i) {
      for (int l = 0; l < L; ++l) { w.xhat_v[i][l] = take(BH); w.rstd_v[i][l] = take(B); }
      w.ln_dg[i] = take((int64_t)nln * H); w.ln_db[i] = take((int64_t)nln * H); w.ln_dh[i] = take((int64_t)nln * H);
    }
  }
  w.total = o;
  *out = h;
  return PORL_OK;
}

void porl_iql_destroy(porl_iql* h) { delete h; }

int64_t porl_iql_group_floats(const porl_iql* h, int group) { return !h ? 0 : (group == 0 ? h->n_vf : h->n_pol); }
int32_t porl_iql_group_tensors(const porl_iql* h, int group) {
  return !h ? 0 : (int32_t)(group == 0 ? h->t_vf.size() : h->t_pol.size());
}
int porl_iql_tensor_info(const porl_iql* h, int group, int index, int64_t* offset, int32_t* rows, int32_t* cols) {
  if (!h || group < 0 || group > 1) PORL_FAIL(PORL_ERR_INVALID, "bad group");
  const auto& v = group == 0 ? h->t_vf : h->t_pol;
  if (index < 0 || index >= (int)v.size()) PORL_FAIL(PORL_ERR_INVALID, "tensor index out of range");
  if (offset) *offset = v[index].off;
  if (rows) *rows = v[index].rows;
  if (cols) *cols = v[index].cols;
  return PORL_OK;
}
int64_t porl_iql_workspace_floats(const porl_iql* h) { return h ? h->ws.total : 0; }

int porl_iql_bind(porl_iql* h, const porl_iql_buffers* b) {
  if (!h || !b) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  const void* ptrs[] = {b->params_vf, b->params_tgt, b->params_pol, b->grads_vf, b->grads_pol, b->adam_m_vf,
                        b->adam_v_vf, b->adam_m_pol, b->adam_v_pol, b->workspace, b->stats};
  for (const void* p : ptrs) {
    if (!p) PORL_FAIL(PORL_ERR_INVALID, "null buffer");
    if (!aligned16(p)) PORL_FAIL(PORL_ERR_INVALID, "buffers must be 16-byte aligned");
  }
  h->buf = *b;
  h->bound = true;
  h->device = device_of(b->workspace);
  h->batch = 0;
  return PORL_OK;
}

int porl_iql_load_batch(porl_iql* h, int32_t batch, const float* obs, int64_t obs_rs, const float* next_obs,
                        int64_t next_rs, const float* rew, int64_t rew_rs, const float* term, int64_t term_rs,
                        const float* pol_target, int64_t pt_rs, void* stream) {
  PORL_TRY(check_ready(h, false)); DevGuard _dg(h->device);
  if (batch < 1 || batch > h->cfg.max_batch) PORL_FAIL(PORL_ERR_INVALID, "batch %d outside [1,%d]", batch, h->cfg.max_batch);
  if (!obs || !next_obs || !rew || !term) PORL_FAIL(PORL_ERR_INVALID, "null batch tensor");
  float* W = h->buf.workspace;
  if (h->mode & PORL_IQL_MODE_TWO_SLOTS) h->slot = (h->slot + 1) % PORL_IQL_SLOTS;
  PackArgs a{};
  a.rows = batch;
  auto job = [&](const float* src, int64_t rs, float* dst, int cols, int ld) {
    PackJob& j = a.job[a.njobs++];
    j.src = src; j.dst = dst; j.src_row_stride = rs; j.src_col_stride = 1; j.cols = cols; j.ld = ld;
  };
  job(obs, obs_rs, W + h->ws.xs_slot[h->slot], h->cfg.obs_dim, h->Sp);
  job(next_obs, next_rs, W + h->ws.xn, h->cfg.obs_dim, h->Sp);
  job(rew, rew_rs, W + h->ws.rew, 1, 1);
  job(term, term_rs, W + h->ws.term, 1, 1);
  if (pol_target) job(pol_target, pt_rs, W + h->ws.xt_slot[h->slot], h->cfg.pol_out_dim, h->Dp);
  h->have_pol_target = pol_target != nullptr;
  h->pol_prefetched = false;
  h->pol_fwd_done = false;
  const long n = (long)batch * std::max(h->Sp, h->Dp);
  dim3 grid((unsigned)std::min<long>((n + 255) / 256, 1024), a.njobs);
  hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  PORL_HIP(hipGetLastError());
  h->batch = batch;
  return PORL_OK;
}

// LayerNorm variant of the value backward: dZ of every hidden layer is materialised by the LayerNorm
// backward kernel (row statistics), gamma/beta/(head) gradients come from its per-block partial sums.
static int value_backward_ln(porl_iql* h, const porl_iql_hyper* hp, hipStream_t s, ReduceArgs* defer) {
  (void)hp;
  const int B = h->batch, S = h->cfg.obs_dim, H = h->cfg.hidden_dim, L = h->cfg.n_hidden, Hp = h->Hp;
  float* W = h->buf.workspace;
  const Workspace& ws = h->ws;
  const float* Pv = h->buf.params_vf;
  float* Gv = h->buf.grads_vf;
  const int nln = cdiv(B, LN_ROWS_PER_BLOCK);
  for (int l = L - 1; l >= 0; --l) {
    const bool top = l == L - 1;
    const int Kin = l == 0 ? S : H;
    {
      LnBwdArgs a{};
      a.nnets = 2;
      for (int i = 0; i < 2; ++i) {
        a.dH[i] = top ? nullptr : W + ws.dz_v[i][0];
        a.dv[i] = W + ws.dv[i]; a.headw[i] = Pv + h->v[i].w[L];
        a.Hact[i] = W + ws.act_v[i][l]; a.xhat[i] = W + ws.xhat_v[i][l]; a.rstd[i] = W + ws.rstd_v[i][l];
        a.gamma[i] = Pv + h->v[i].lnw[l];
        a.dZ[i] = W + ws.dz_v[i][1];
        a.part_dgamma[i] = W + ws.ln_dg[i]; a.part_dbeta[i] = W + ws.ln_db[i]; a.part_dhead[i] = W + ws.ln_dh[i];
      }
      a.B = B; a.H = H; a.ld = Hp; a.rows_per_block = LN_ROWS_PER_BLOCK;
      hipLaunchKernelGGL(ln_relu_bwd_kernel, dim3(nln, 2), dim3(256), 0, s, a);
      PORL_HIP(hipGetLastError());
      ReduceArgs r{};
      for (int i = 0; i < 2; ++i) {
        add_reduce(r, Gv + h->v[i].lnw[l], W + ws.ln_dg[i], H, H, nln);
        add_reduce(r, Gv + h->v[i].lnb[l], W + ws.ln_db[i], H, H, nln);
        if (top) add_reduce(r, Gv + h->v[i].w[L], W + ws.ln_dh[i], H, H, nln);
      }
      PORL_TRY(launch_reduce(r, s));
    }
    ReduceArgs red{};
    GemmGroup g{};
    for (int i = 0; i < 2; ++i) {
      const float* dz = W + ws.dz_v[i][1];
      const float* in = l == 0 ? W + ws.xs_slot[h->slot] : W + ws.act_v[i][l - 1];
      const int ldin = l == 0 ? h->Sp : Hp;
      GemmProb p = make_prob(GEMM_TN, dz, Hp, in, ldin, Gv + h->v[i].w[l], Kin, H, Kin, B);
      p.colsum = Gv + h->v[i].b[l];
      g.p[g.nprob++] = p;
      if (l > 0)   // unmasked: the ReLU mask of layer l-1 is applied by its own LayerNorm backward
        g.p[g.nprob++] = make_prob(GEMM_NN, dz, Hp, Pv + h->v[i].w[l], Kin, W + ws.dz_v[i][0], Hp, B, Kin, H);
    }
    const int tile = pick_tile(g);
    if (l == 0) {
      int bm, bn;
      tile_dims(tile, bm, bn);
      const int sk = pick_splitk(H, Kin, B, 2, bm, bn);
      if (sk > 1) {
        const int64_t per = (int64_t)H * Kin, perc = H;
        for (int i = 0; i < 2; ++i) {
          float* slabW = W + ws.slab_a + (int64_t)i * SK_MAX * (per + 2 * perc + 8);
          float* slabC = slabW + (int64_t)SK_MAX * per;
          g.p[i].splitk = sk; g.p[i].C = slabW; g.p[i].colsum = slabC;
          add_reduce(red, Gv + h->v[i].w[0], slabW, per, per, sk);
          add_reduce(red, Gv + h->v[i].b[0], slabC, perc, perc, sk);
        }
      }
    }
    PORL_TRY(launch_group(g, tile, s));
    if (l == 0 && defer) *defer = red;       // only the last combine can wait for the Adam launch: the LayerNorm
    else PORL_TRY(launch_reduce(red, s));    // partial buffers are reused layer by layer
  }
  return PORL_OK;
}

static int feistel_half_bits(int64_t n_rows) {
  int bits = 1;
  while ((int64_t(1) << bits) < n_rows) ++bits;
  return std::max(1, (bits + 1) / 2);
}

int porl_iql_load_batch_sampled(porl_iql* h, int32_t batch, const float* rows, int64_t row_stride, int64_t n_rows,
                                int32_t act_dim, int32_t target_is_action, uint64_t seed, uint64_t step,
                                int64_t* idx_out, void* stream) {
  PORL_TRY(check_ready(h, false)); DevGuard _dg(h->device);
  if (batch < 1 || batch > h->cfg.max_batch) PORL_FAIL(PORL_ERR_INVALID, "batch %d outside [1,%d]", batch, h->cfg.max_batch);
  if (!rows || n_rows < batch || n_rows > (int64_t(1) << 40)) PORL_FAIL(PORL_ERR_INVALID, "need batch <= n_rows");
  const int S = h->cfg.obs_dim, D = h->cfg.pol_out_dim;
  if (row_stride < 2 * (int64_t)S + 2 + act_dim) PORL_FAIL(PORL_ERR_INVALID, "row stride shorter than 2*S+2+A");
  if (target_is_action ? D != act_dim : D != S) PORL_FAIL(PORL_ERR_INVALID, "policy target width mismatch");
  float* W = h->buf.workspace;
  if (h->mode & PORL_IQL_MODE_TWO_SLOTS) h->slot = (h->slot + 1) % PORL_IQL_SLOTS;
  SampledBatchArgs a{};
  a.rows = rows; a.row_stride = (long)row_stride; a.n_rows = n_rows;
  a.batch = batch; a.S = S; a.A = act_dim; a.D = D; a.Sp = h->Sp; a.Dp = h->Dp;
  a.hb = feistel_half_bits(n_rows); a.target_is_action = target_is_action;
  a.seed = seed; a.step = step;
  a.xs = W + h->ws.xs_slot[h->slot]; a.xn = W + h->ws.xn; a.xt = W + h->ws.xt_slot[h->slot]; a.rew = W + h->ws.rew; a.term = W + h->ws.term;
  a.idx_out = idx_out;
  {
    g_phase = "V1.";
    ProfScope ps("sampled_batch_kernel", (hipStream_t)stream, 0.0, 8.0 * batch * (2 * S + 2 + act_dim));
    hipLaunchKernelGGL(sampled_batch_kernel, dim3(cdiv(batch, 4)), dim3(256), 0, (hipStream_t)stream, a);
    g_phase = "";
  }
  PORL_HIP(hipGetLastError());
  h->batch = batch;
  h->have_pol_target = true;
  h->pol_prefetched = false;
  h->pol_fwd_done = false;
  return PORL_OK;
}

int porl_iql_set_mode(porl_iql* h, int32_t mode) {
  if (!h) PORL_FAIL(PORL_ERR_INVALID, "null engine");
  if (mode & ~(PORL_IQL_MODE_TWO_SLOTS | PORL_IQL_MODE_FOLD_COMBINE | PORL_IQL_MODE_SHORT_BLOCKS)) PORL_FAIL(PORL_ERR_INVALID, "unknown mode bits");
  if (h->fin_v_pending || h->fin_p_pending) PORL_FAIL(PORL_ERR_INVALID, "a backward pass is waiting for its apply call");
  h->mode = mode;
  return PORL_OK;
}

int porl_iql_set_stats(porl_iql* h, float* stats) {
  PORL_TRY(check_ready(h, false)); DevGuard _dg(h->device);
  if (!stats) PORL_FAIL(PORL_ERR_INVALID, "null stats");
  h->buf.stats = stats;
  return PORL_OK;
}


// Input-layer weight gradients dW0 = dZ0^T X (+ db0) of up to 4 nets on wgrad_skinny_kernel (skinny.hpp): slabs over
// batch chunks in the split-K layout, combine jobs appended to `red`.  Returns false when the shapes do not qualify
// (the caller then takes the grouped-GEMM path).
struct Dw0Net { const float* dz; float* gw; float* gb; float* slabW; float* slabC; };
static bool dw0_skinny(const Dw0Net* nets, int nnets, const float* X, int ldx, int B, int H, int S, int ldz, ReduceArgs& red,
                       const char* phase, hipStream_t s, int* rc) {
  *rc = 0;
  if (!(skinny_mask() & 1) || S > SKN_T || nnets > SKN_MAX_NETS || H % 4 || !skn_ok4(X, ldx) || ldz % 4) return false;
  for (int i = 0; i < nnets; ++i) if (!skn_ok4(nets[i].dz, ldz)) return false;
  WgradSkinnyArgs a{};
  a.nnets = nnets; a.B = B; a.H = H; a.S = S; a.ldz = ldz; a.ldx = ldx; a.ldo = S;
  a.tiles_n = cdiv(H, SKN_T);
  skn_split(cdiv(B, SKN_T), g_dw0_slabs, a.nslab, a.rtiles);
  a.slabW_stride = (long)H * S; a.slabC_stride = H;
  for (int i = 0; i < nnets; ++i) {
    a.net[i] = WgradSkinnyNet{nets[i].dz, X, nets[i].slabW, nets[i].slabC};
    add_reduce(red, nets[i].gw, nets[i].slabW, (long)H * S, (long)H * S, a.nslab);
    add_reduce(red, nets[i].gb, nets[i].slabC, H, H, a.nslab);
  }
  g_phase = phase;
  {
    ProfScope ps("wgrad_skinny_kernel", s, 2.0 * nnets * B * (double)H * S,
                 4.0 * nnets * ((double)B * H + (double)B * S + (double)a.nslab * H * (S + 1)));
    hipLaunchKernelGGL(wgrad_skinny_kernel, dim3(nnets * a.nslab * a.tiles_n), dim3(256), 0, s, a);
  }
  g_phase = "";
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_err = std::string("wgrad_skinny_kernel: ") + hipGetErrorString(e); *rc = (int)e; }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// `defer` (optional): the final combine launch is not issued; its jobs are handed to the caller, who folds them into
// the Adam launch (porl_iql_step).
static int value_backward_impl(porl_iql* h, const porl_iql_hyper* hp, hipStream_t s, ReduceArgs* defer) {
  const int B = h->batch, S = h->cfg.obs_dim, H = h->cfg.hidden_dim, L = h->cfg.n_hidden, Hp = h->Hp;
  float* W = h->buf.workspace;
  const Workspace& ws = h->ws;
  const float* Pv = h->buf.params_vf;
  const float* Pt = h->buf.params_tgt;
  float* Gv = h->buf.grads_vf;
  int parts = 0;
  const bool LN = h->cfg.layer_norm != 0;

  // -- forward: target twins on s', online twins on s — 4 nets per launch ---------------------------
  for (int l = 0; l < L; ++l) {
    FwdNet nets[4];
    const int K = l == 0 ? S : H;
    for (int n = 0; n < 4; ++n) {
      const bool tgt = n < 2;
      const int i = n & 1;
      const float* P = tgt ? Pt : Pv;
      FwdNet& f = nets[n];
      if (l == 0) { f.in = W + (tgt ? ws.xn : ws.xs_slot[h->slot]); f.ldin = h->Sp; }
      else { f.in = W + (tgt ? ws.act_t[i][(l - 1) & 1] : ws.act_v[i][l - 1]); f.ldin = Hp; }
      f.W = P + h->v[i].w[l]; f.b = P + h->v[i].b[l];
      const bool last = l == L - 1;
      f.out = tgt ? ((last && !LN) ? nullptr : W + ws.act_t[i][l & 1]) : W + ws.act_v[i][l];
      f.headw = P + h->v[i].w[L];
      f.headout = W + (tgt ? ws.hp_t[i] : ws.hp_v[i]);
      if (LN) {
        f.ln_g = P + h->v[i].lnw[l]; f.ln_b = P + h->v[i].lnb[l];
        if (!tgt) { f.xhat = W + ws.xhat_v[i][l]; f.rstd = W + ws.rstd_v[i][l]; }
      }
    }
    g_phase = l == 0 ? "V2.L0fwd:" : "V3.fwd:";
    PORL_TRY(fwd_hidden_layer(h, nets, 4, B, K, l == L - 1, &parts, s));
  }
  g_phase = "V4.head:";

  // -- TD target, expectile loss, dL/dv: with LayerNorm a kernel of its own; otherwise fused into the head
  //    backward below --------------------------------------------------------------------------------------
  if (LN) {
    ValueLossArgs a{};
    for (int i = 0; i < 2; ++i) {
      a.hp_t[i] = W + ws.hp_t[i]; a.hp_v[i] = W + ws.hp_v[i];
      a.b_t[i] = Pt + h->v[i].b[L]; a.b_v[i] = Pv + h->v[i].b[L];
      a.dv[i] = W + ws.dv[i];
      a.db_out[i] = Gv + h->v[i].b[L];
    }
    a.rew = W + ws.rew; a.term = W + ws.term; a.target_v = W + ws.target_v_slot[h->slot]; a.stats = h->buf.stats;
    a.B = B; a.parts = parts; a.tau = hp->tau; a.discount = hp->discount; a.inv_batch = hp->inv_batch;
    hipLaunchKernelGGL(value_loss_kernel, dim3(1), dim3(1024), 0, s, a);
    PORL_HIP(hipGetLastError());
  }

  ReduceArgs red{};
  if (LN) return value_backward_ln(h, hp, s, defer);
  // -- head + last ReLU backward: dZ_{L-1} = dv w_L^T . 1[H_{L-1} > 0], dW_L = dv^T H_{L-1} (partials) -----
  {
    const int nhb = cdiv(B, HEAD_ROWS_PER_BLOCK);
    HeadBwdArgs a{};
    for (int i = 0; i < 2; ++i) {
      a.Hact[i] = W + ws.act_v[i][L - 1]; a.w[i] = Pv + h->v[i].w[L];
      a.dZ[i] = W + ws.dz_v[i][(L - 1) & 1]; a.part_dw[i] = W + ws.head_part[i];
      a.hp_t[i] = W + ws.hp_t[i]; a.hp_v[i] = W + ws.hp_v[i];
      a.b_t[i] = Pt + h->v[i].b[L]; a.b_v[i] = Pv + h->v[i].b[L];
      a.dv[i] = W + ws.dv[i]; a.part_db[i] = W + ws.head_db[i];
      add_reduce(red, Gv + h->v[i].w[L], W + ws.head_part[i], H, H, nhb);
      add_reduce(red, Gv + h->v[i].b[L], W + ws.head_db[i], 1, 1, nhb);
    }
    a.rew = W + ws.rew; a.term = W + ws.term; a.target_v = W + ws.target_v_slot[h->slot]; a.part_loss = W + ws.head_loss;
    add_reduce(red, h->buf.stats, W + ws.head_loss, 1, 1, nhb, 0, hp->inv_batch);     // stats[0] = v_loss
    a.B = B; a.H = H; a.ld = Hp; a.parts = parts;
    a.tau = hp->tau; a.discount = hp->discount; a.inv_batch = hp->inv_batch;
    ProfScope ps("relu_head_bwd_kernel", s, 0.0, 16.0 * B * H);
    hipLaunchKernelGGL(relu_head_bwd_kernel, dim3(nhb, 2), dim3(256), 0, s, a);
    PORL_HIP(hipGetLastError());
  }
  for (int l = L - 1; l >= 0; --l) {
    const int Kin = l == 0 ? S : H;
    if (l == 0) {
      // skinny (H x S) weight gradient on its own kernel: one 64 x 64 tile of dZ0 per block, slabs over batch chunks
      Dw0Net dn[2];
      const int64_t per = (int64_t)H * S, perc = H;
      for (int i = 0; i < 2; ++i) {
        float* slabW = W + ws.slab_a + (int64_t)i * SK_MAX * (per + 2 * perc + 8);
        dn[i] = Dw0Net{W + ws.dz_v[i][0], Gv + h->v[i].w[0], Gv + h->v[i].b[0], slabW, slabW + (int64_t)SK_MAX * per};
      }
      int rc = 0;
      if (dw0_skinny(dn, 2, W + ws.xs_slot[h->slot], h->Sp, B, H, S, Hp, red, "V6.dW0:", s, &rc)) {
        if (rc) return rc;
        break;
      }
    }
    GemmGroup g{};
    for (int i = 0; i < 2; ++i) {
      const float* dz = W + ws.dz_v[i][l & 1];
      const float* in = l == 0 ? W + ws.xs_slot[h->slot] : W + ws.act_v[i][l - 1];
      const int ldin = l == 0 ? h->Sp : Hp;
      GemmProb p = make_prob(GEMM_TN, dz, Hp, in, ldin, Gv + h->v[i].w[l], Kin, H, Kin, B);
      p.colsum = Gv + h->v[i].b[l];
      g.p[g.nprob++] = p;
      if (l > 0) {
        GemmProb q = make_prob(GEMM_NN, dz, Hp, Pv + h->v[i].w[l], Kin, W + ws.dz_v[i][(l - 1) & 1], Hp, B, Kin, H);
        q.mask = W + ws.act_v[i][l - 1]; q.ldmask = Hp;
        g.p[g.nprob++] = q;
      }
    }
    int tile = pick_tile(g);
    if (l > 0 && g_short_blocks && g_vbwd_tile_short >= 0) tile = g_vbwd_tile_short;
    if (l == 0) {
      // skinny (H x S) weight gradient: split the batch (K) dimension to fill the chip
      int bm, bn;
      tile_dims(tile, bm, bn);
      const int sk = pick_splitk(H, Kin, B, 2, bm, bn);
      if (sk > 1) {
        const int64_t per = (int64_t)H * Kin, perc = H;
        for (int i = 0; i < 2; ++i) {
          float* slabW = W + ws.slab_a + (int64_t)i * SK_MAX * (per + 2 * perc + 8);
          float* slabC = slabW + (int64_t)SK_MAX * per;
          g.p[i].splitk = sk; g.p[i].C = slabW; g.p[i].colsum = slabC;
          add_reduce(red, Gv + h->v[i].w[0], slabW, per, per, sk);
          add_reduce(red, Gv + h->v[i].b[0], slabC, perc, perc, sk);
        }
      }
    }
    g_phase = l == 0 ? "V6.dW0:" : "V5.bwd:";
    PORL_TRY(launch_group(g, tile, s));
  }
  g_phase = "V7.combine:";
  if (defer) *defer = red;
  else PORL_TRY(launch_reduce(red, s));
  g_phase = "";
  return PORL_OK;
}

int porl_iql_value_backward(porl_iql* h, const porl_iql_hyper* hp, void* stream) {
  PORL_TRY(check_ready(h, true)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  const bool fold = (h->mode & PORL_IQL_MODE_FOLD_COMBINE) != 0;
  h->fin_v = ReduceArgs{};
  PORL_TRY(value_backward_impl(h, hp, (hipStream_t)stream, fold ? &h->fin_v : nullptr));
  h->fin_v_pending = fold;
  return PORL_OK;
}

// ---------------------------------------------------------------------------------------------------
// Adam(+EMA) over one flat group.  `fin` (optional): combine jobs that the backward pass left pending; those whose
// output lies inside the gradient buffer `g` are folded into this launch (short slab lists inside the float4 sweep,
// long ones as extra reduce+Adam blocks), the others (loss statistics) ride along as plain reduce blocks.
static int adam_launch(float* p, float* g, float* m, float* v, float* tgt, int64_t n, double lr, int step,
                       double b1, double b2, double eps, double ema_beta, hipStream_t s, const ReduceArgs* fin = nullptr) {
  if (n < 0 || n % 4) PORL_FAIL(PORL_ERR_INVALID, "adam range must be a non-negative multiple of 4 floats");
  if (step < 1) PORL_FAIL(PORL_ERR_INVALID, "adam step must be >= 1");
  if (n == 0 && !(fin && fin->njobs)) return PORL_OK;      // an empty range is an empty sweep (the grid maths below divide by it)
  AdamArgs a{};
  a.p = p; a.g = g; a.m = m; a.v = v; a.tgt = tgt;
  // torch._single_tensor_adam: python doubles, rounded to fp32 where they meet tensors
  const double bc1 = 1.0 - std::pow(b1, (double)step);
  const double bc2 = 1.0 - std::pow(b2, (double)step);
  a.s.step_size = (float)(lr / bc1);
  a.s.bc2_sqrt = (float)std::sqrt(bc2);
  a.s.omb1 = (float)(1.0 - b1); a.s.beta2 = (float)b2; a.s.omb2 = (float)(1.0 - b2);
  a.s.eps = (float)eps; a.s.ema_beta = (float)ema_beta; a.s.omeb = (float)(1.0 - ema_beta);
  a.n4 = n / 4;
  // one float4 per thread and a single trip through the block's loop whenever the grid allows it: a second, nearly
  // empty trip would double every block's memory round trips (measured: 12 -> 19 us on the 80 MB value group)
  const int sweep_blocks = (int)std::min<long>((a.n4 + 255) / 256, 1 << 20);
  a.span4 = sweep_blocks ? ((a.n4 + sweep_blocks - 1) / sweep_blocks + 255) / 256 * 256 : 256;
  int reduce_blocks = 0;
  double extra_bytes = 0.0;
  if (fin) {
    for (int q = 0; q < fin->njobs; ++q) {
      ReduceJob j = fin->job[q];
      const bool inside = j.out >= g && j.out + j.n <= g + n;
      const long off = inside ? (long)(j.out - g) : -1;
      extra_bytes += 4.0 * (double)j.n * j.nslab;
      const bool plain = j.op == 0 && j.scale == 1.f && !j.bias && j.act == 0;
      if (inside && plain && j.nslab <= ADAM_SWEEP_SLABS && off % 4 == 0 && j.n % 4 == 0 && j.stride % 4 == 0 &&
          aligned16(j.slab) && a.nregions < MAX_ADAM_REGIONS) {
        AdamRegion& R = a.region[a.nregions++];
        R.lo = off; R.hi = off + j.n; R.slab = j.slab; R.stride = j.stride; R.nslab = j.nslab;
        continue;
      }
      if (inside) {
        if (off % 4) PORL_FAIL(PORL_ERR_INVALID, "gradient tensors start on 16-byte boundaries");
        j.adam_off = off;
        a.skip_lo[a.nskip] = off; a.skip_hi[a.nskip] = ru4(off + j.n); ++a.nskip;
      } else {
        j.adam_off = -1;
      }
      a.job_block0[a.r.njobs] = reduce_blocks;
      a.r.job[a.r.njobs++] = j;
      reduce_blocks += (int)((j.n + j.width - 1) / j.width);
    }
    a.job_block0[a.r.njobs] = reduce_blocks;
  }
  ProfScope ps("adam_ema_kernel", s, 0.0, (double)n * (tgt ? 36.0 : 28.0) + extra_bytes);
  a.reduce_blocks = reduce_blocks;
  hipLaunchKernelGGL(adam_ema_kernel, dim3(sweep_blocks + reduce_blocks), dim3(256), 0, s, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iql_value_apply(porl_iql* h, const porl_iql_hyper* hp, void* stream) {
  PORL_TRY(check_ready(h, false)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  g_phase = "V8.";
  const int rc = adam_launch(h->buf.params_vf, h->buf.grads_vf, h->buf.adam_m_vf, h->buf.adam_v_vf, h->buf.params_tgt,
                             h->n_vf, hp->value_lr, hp->value_step, hp->adam_beta1, hp->adam_beta2, hp->adam_eps,
                             hp->ema_beta, (hipStream_t)stream, h->fin_v_pending ? &h->fin_v : nullptr);
  g_phase = "";
  h->fin_v_pending = false;
  return rc;
}

int porl_iql_policy_apply(porl_iql* h, const porl_iql_hyper* hp, void* stream) {
  PORL_TRY(check_ready(h, false)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  g_phase = "P9.";
  const int rc = adam_launch(h->buf.params_pol, h->buf.grads_pol, h->buf.adam_m_pol, h->buf.adam_v_pol, nullptr, h->n_pol,
                             hp->policy_lr, hp->policy_step, hp->adam_beta1, hp->adam_beta2, hp->adam_eps, 0.0,
                             (hipStream_t)stream, h->fin_p_pending ? &h->fin_p : nullptr);
  g_phase = "";
  h->fin_p_pending = false;
  return rc;
}

// ---------------------------------------------------------------------------------------------------
// policy mean (pre-bias, pre-activation) as split-K slabs: act_p[L-1] (B,H) x W_L^T (H,D)
static int policy_mean_slabs(porl_iql* h, int B, int* nslab, hipStream_t s) {
  const int H = h->cfg.hidden_dim, D = h->cfg.pol_out_dim, L = h->cfg.n_hidden;
  float* W = h->buf.workspace;
  if ((skinny_mask() & 2) && D <= SKN_T && H % 4 == 0 && skn_ok4(W + h->ws.act_p[L - 1], h->Hp) &&
      skn_ok4(h->buf.params_pol + h->pol.w[L], H) && h->Dp % 4 == 0) {
    MeanSkinnyArgs a{};
    a.A = W + h->ws.act_p[L - 1]; a.W = h->buf.params_pol + h->pol.w[L]; a.slab = W + h->ws.slab_mean;
    a.B = B; a.H = H; a.D = D; a.lda = h->Hp; a.ldw = H; a.ldo = h->Dp;
    a.tiles_m = cdiv(B, SKN_T);
    // few row tiles (small batches): more slabs keep the chip busy; at B >= 1024 eight 128-deep slabs halve what the
    // NLL kernel reads back
    skn_split(cdiv(H, SKN_T), a.tiles_m >= 16 ? SK_MAX / 2 : SK_MAX, a.nslab, a.ktiles);
    a.slab_stride = (long)B * h->Dp;
    *nslab = a.nslab;
    ProfScope ps("mean_skinny_kernel", s, 2.0 * B * (double)H * D, 4.0 * ((double)B * H + (double)D * H + (double)a.nslab * B * h->Dp));
    hipLaunchKernelGGL(mean_skinny_kernel, dim3(a.tiles_m * a.nslab), dim3(256), 0, s, a);
    PORL_HIP(hipGetLastError());
    return 0;
  }
  GemmGroup g{};
  g.nprob = 1;
  g.p[0] = make_prob(GEMM_NT, W + h->ws.act_p[L - 1], h->Hp, h->buf.params_pol + h->pol.w[L], H, W + h->ws.slab_mean,
                     h->Dp, B, D, H);
  const int tile = pick_tile(g);
  int bm, bn;
  tile_dims(tile, bm, bn);
  const int sk = pick_splitk(B, D, H, 1, bm, bn);
  // splitk == 1 still writes the raw product (bias and tanh are applied by the consumer)
  g.p[0].splitk = sk;
  *nslab = sk;
  if (sk == 1) { g.p[0].bias = nullptr; g.p[0].act = ACT_NONE; }
  return launch_group(g, tile, s);
}

// The part of the policy step that does not depend on the value networks: the policy MLP's forward on the loaded
// observations.  In data-parallel mode it runs while the value-gradient all-reduce is on the wire.
int porl_iql_policy_prefetch(porl_iql* h, void* stream) {
  PORL_TRY(check_ready(h, true)); DevGuard _dg(h->device);
  if (!h->have_pol_target) PORL_FAIL(PORL_ERR_INVALID, "policy step needs pol_target in porl_iql_load_batch");
  hipStream_t s = (hipStream_t)stream;
  const int B = h->batch, S = h->cfg.obs_dim, H = h->cfg.hidden_dim, L = h->cfg.n_hidden;
  float* W = h->buf.workspace;
  const Workspace& ws = h->ws;
  const float* Pp = h->buf.params_pol;
  int parts = 0;
  for (int l = 0; l < L; ++l) {
    FwdNet f{};
    if (l == 0) { f.in = W + ws.xs_slot[h->slot]; f.ldin = h->Sp; }
    else { f.in = W + ws.act_p[l - 1]; f.ldin = h->Hp; }
    f.W = Pp + h->pol.w[l]; f.b = Pp + h->pol.b[l];
    f.out = W + ws.act_p[l]; f.headw = nullptr; f.headout = nullptr;
    PORL_TRY(fwd_hidden_layer(h, &f, 1, B, l == 0 ? S : H, l == L - 1, &parts, s));
  }
  PORL_TRY(policy_mean_slabs(h, B, &h->pol_nslab, s));
  h->pol_prefetched = true;
  return PORL_OK;
}

// part: 1 = forward half only (second twin forward, policy forward, mean, weights + NLL + dL/dmean: everything of the
// policy phase that reads the VALUE parameters), 2 = gradient half only (needs the forward half of the same batch),
// 3 = both.
static int policy_backward_impl(porl_iql* h, const porl_iql_hyper* hp, hipStream_t s, ReduceArgs* defer, int part = 3) {
  if (!h->have_pol_target) PORL_FAIL(PORL_ERR_INVALID, "policy step needs pol_target in porl_iql_load_batch");
  const int B = h->batch, S = h->cfg.obs_dim, H = h->cfg.hidden_dim, L = h->cfg.n_hidden, D = h->cfg.pol_out_dim;
  const int Hp = h->Hp, Dp = h->Dp;
  float* W = h->buf.workspace;
  const Workspace& ws = h->ws;
  const float* Pv = h->buf.params_vf;
  const float* Pp = h->buf.params_pol;
  float* Gp = h->buf.grads_pol;
  int parts = 0;
  const bool LN = h->cfg.layer_norm != 0;

  const int nblk = cdiv(B, NLL_ROWS_PER_BLOCK);
  if (part == 2 && !h->pol_fwd_done) PORL_FAIL(PORL_ERR_INVALID, "porl_iql_policy_forward has not run for this batch");
  if ((part & 1) && !h->pol_fwd_done) {
  // -- forward: updated twins (head only) + policy hidden layers, 3 nets per launch; the policy net is left out
  //    when porl_iql_policy_prefetch already ran it for this batch ---------------------------------------------
  const bool pre = h->pol_prefetched;
  h->pol_prefetched = false;
  for (int l = 0; l < L; ++l) {
    FwdNet nets[3];
    const int K = l == 0 ? S : H;
    const bool last = l == L - 1;
    for (int i = 0; i < 2; ++i) {
      FwdNet& f = nets[i];
      if (l == 0) { f.in = W + ws.xs_slot[h->slot]; f.ldin = h->Sp; }
      else { f.in = W + ws.act_q[i][(l - 1) & 1]; f.ldin = Hp; }
      f.W = Pv + h->v[i].w[l]; f.b = Pv + h->v[i].b[l];
      f.out = (last && !LN) ? nullptr : W + ws.act_q[i][l & 1];
      f.headw = Pv + h->v[i].w[L]; f.headout = W + ws.hp_q[i];
      if (LN) { f.ln_g = Pv + h->v[i].lnw[l]; f.ln_b = Pv + h->v[i].lnb[l]; }
    }
    FwdNet& f = nets[2];
    if (l == 0) { f.in = W + ws.xs_slot[h->slot]; f.ldin = h->Sp; }
    else { f.in = W + ws.act_p[l - 1]; f.ldin = Hp; }
    f.W = Pp + h->pol.w[l]; f.b = Pp + h->pol.b[l];
    f.out = W + ws.act_p[l]; f.headw = nullptr; f.headout = nullptr;
    g_phase = l == 0 ? "P1.L0fwd:" : "P2.fwd:";
    PORL_TRY(fwd_hidden_layer(h, nets, pre ? 2 : 3, B, K, last, &parts, s));
  }
  int nslab = h->pol_nslab;
  g_phase = "P3.mean:";
  if (!pre) PORL_TRY(policy_mean_slabs(h, B, &nslab, s));
  g_phase = "P4.";

  // -- advantage weights, NLL, dL/dmean, dL/dlog_std --------------------------------------------------
  {
    PolicyNllArgs a{};
    for (int i = 0; i < 2; ++i) { a.hp_v[i] = W + ws.hp_q[i]; a.b_v[i] = Pv + h->v[i].b[L]; }
    a.parts = parts; a.target_v = W + ws.target_v_slot[h->slot];
    a.mean_slab = W + ws.slab_mean; a.nslab = nslab; a.slab_stride = (long)B * Dp;
    a.mean_bias = Pp + h->pol.b[L]; a.log_std = Pp + h->logstd_off;
    a.x = W + ws.xt_slot[h->slot]; a.ldx = Dp; a.dmean = W + ws.dmu; a.ldd = Dp;
    a.part_loss = W + ws.part_loss; a.part_min = W + ws.part_min; a.part_dls = W + ws.part_dls;
    a.B = B; a.D = D; a.ldm = Dp; a.tanh_mean = h->cfg.pol_tanh; a.weight_mode = h->cfg.weight_mode;
    a.alpha = hp->alpha; a.inv_batch = hp->inv_batch; a.rows_per_block = NLL_ROWS_PER_BLOCK;
    ProfScope ps("policy_nll_kernel", s, 0.0, 4.0 * B * (nslab + 2.0) * Dp);
    hipLaunchKernelGGL(policy_nll_kernel, dim3(nblk), dim3(256), 0, s, a);
    PORL_HIP(hipGetLastError());
  }
  g_phase = "";
  h->pol_fwd_done = true;
  }
  if (!(part & 2)) return PORL_OK;
  h->pol_fwd_done = false;

  // -- backward ---------------------------------------------------------------------------------------
  ReduceArgs red{};
  // the NLL kernel's per-block partials are combined by the step's final reduce launch
  add_reduce(red, Gp + h->logstd_off, W + ws.part_dls, D, D, nblk);            // d/dlog_std (already clamp-masked)
  add_reduce(red, h->buf.stats + 1, W + ws.part_loss, 1, 1, nblk);            // stats[1] = g_loss
  add_reduce(red, h->buf.stats + 2, W + ws.part_min, 1, 1, nblk, /*min*/ 1);  // stats[2] = min NLL
  if ((skinny_mask() & 4) && D <= SKN_T && H % 4 == 0 && Dp % 4 == 0 && Hp % 4 == 0 && skn_ok4(Pp + h->pol.w[L], H)) {
    // output layer on out_bwd_kernel (skinny.hpp): one pass over the last hidden activation gives dZ_{L-1} and the
    // dW_L / db_L slabs
    OutBwdArgs a{};
    a.dmu = W + ws.dmu; a.W = Pp + h->pol.w[L]; a.A = W + ws.act_p[L - 1];
    a.dZ = W + ws.dz_p[(L - 1) & 1];
    a.slabW = W + ws.slab_b; a.slabC = a.slabW + (int64_t)SK_MAX * D * H;
    a.B = B; a.H = H; a.D = D; a.lddmu = Dp; a.ldw = H; a.lda = Hp; a.lddz = Hp;
    a.tiles_n = cdiv(H, SKN_T);
    skn_split(cdiv(B, SKN_T), SK_MAX, a.nslab, a.rtiles);
    a.slabW_stride = (long)D * H; a.slabC_stride = D;
    add_reduce(red, Gp + h->pol.w[L], a.slabW, (long)D * H, (long)D * H, a.nslab);
    add_reduce(red, Gp + h->pol.b[L], a.slabC, D, D, a.nslab);
    g_phase = "P5.outbwd:";
    {
      ProfScope ps("out_bwd_kernel", s, 4.0 * B * (double)H * D, 4.0 * (2.0 * B * H + (double)B * D + (double)D * H + (double)a.nslab * D * H));
      hipLaunchKernelGGL(out_bwd_kernel, dim3(a.nslab * a.tiles_n), dim3(256), 0, s, a);
    }
    g_phase = "";
    PORL_HIP(hipGetLastError());
  } else {
    // output layer: dW_L = dmu^T H_{L-1} (D x H, skinny M), and dZ_{L-1} = (dmu W_L) . 1[H_{L-1} > 0]
    GemmGroup g{};
    g.nprob = 2;
    g.p[0] = make_prob(GEMM_TN, W + ws.dmu, Dp, W + ws.act_p[L - 1], Hp, Gp + h->pol.w[L], H, D, H, B);
    g.p[0].colsum = Gp + h->pol.b[L];
    g.p[1] = make_prob(GEMM_NN, W + ws.dmu, Dp, Pp + h->pol.w[L], H, W + ws.dz_p[(L - 1) & 1], Hp, B, H, D);
    g.p[1].mask = W + ws.act_p[L - 1]; g.p[1].ldmask = Hp;
    const int tile = D <= 64 ? TILE_64x128 : pick_tile(g);
    int bm, bn;
    tile_dims(tile, bm, bn);
    const int sk = pick_splitk(D, H, B, 1, bm, bn);
    if (sk > 1) {
      float* slabW = W + ws.slab_b;
      float* slabC = slabW + (int64_t)SK_MAX * D * H;
      g.p[0].splitk = sk; g.p[0].C = slabW; g.p[0].colsum = slabC;
      add_reduce(red, Gp + h->pol.w[L], slabW, (long)D * H, (long)D * H, sk);
      add_reduce(red, Gp + h->pol.b[L], slabC, D, D, sk);
    }
    g_phase = "P5.outbwd:";
    PORL_TRY(launch_group(g, tile, s));
  }
  for (int l = L - 1; l >= 0; --l) {
    const int Kin = l == 0 ? S : H;
    if (l == 0) {
      const int64_t per = (int64_t)H * S;
      float* slabW = W + ws.slab_pa;
      Dw0Net dn{W + ws.dz_p[0], Gp + h->pol.w[0], Gp + h->pol.b[0], slabW, slabW + (int64_t)SK_MAX * per};
      int rc = 0;
      if (dw0_skinny(&dn, 1, W + ws.xs_slot[h->slot], h->Sp, B, H, S, Hp, red, "P7.dW0:", s, &rc)) {
        if (rc) return rc;
        break;
      }
    }
    const float* dz = W + ws.dz_p[l & 1];
    const float* in = l == 0 ? W + ws.xs_slot[h->slot] : W + ws.act_p[l - 1];
    const int ldin = l == 0 ? h->Sp : Hp;
    GemmGroup g{};
    g.p[g.nprob] = make_prob(GEMM_TN, dz, Hp, in, ldin, Gp + h->pol.w[l], Kin, H, Kin, B);
    g.p[g.nprob++].colsum = Gp + h->pol.b[l];
    if (l > 0) {
      GemmProb q = make_prob(GEMM_NN, dz, Hp, Pp + h->pol.w[l], Kin, W + ws.dz_p[(l - 1) & 1], Hp, B, Kin, H);
      q.mask = W + ws.act_p[l - 1]; q.ldmask = Hp;
      g.p[g.nprob++] = q;
    }
    const int tile = pick_tile(g);
    if (l == 0) {
      int bm, bn;
      tile_dims(tile, bm, bn);
      const int sk = pick_splitk(H, Kin, B, 1, bm, bn);
      if (sk > 1) {
        const int64_t per = (int64_t)H * Kin;
        float* slabW = W + ws.slab_pa;
        float* slabC = slabW + (int64_t)SK_MAX * per;
        g.p[0].splitk = sk; g.p[0].C = slabW; g.p[0].colsum = slabC;
        add_reduce(red, Gp + h->pol.w[0], slabW, per, per, sk);
        add_reduce(red, Gp + h->pol.b[0], slabC, H, H, sk);
      }
    }
    g_phase = l == 0 ? "P7.dW0:" : "P6.bwd:";
    PORL_TRY(launch_group(g, tile, s));
  }
  g_phase = "P8.combine:";
  if (defer) *defer = red;
  else PORL_TRY(launch_reduce(red, s));
  g_phase = "";
  return PORL_OK;
}

int porl_iql_policy_forward(porl_iql* h, const porl_iql_hyper* hp, void* stream) {
  PORL_TRY(check_ready(h, true)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  return policy_backward_impl(h, hp, (hipStream_t)stream, nullptr, 1);
}

int porl_iql_policy_backward(porl_iql* h, const porl_iql_hyper* hp, void* stream) {
  PORL_TRY(check_ready(h, true)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  const bool fold = (h->mode & PORL_IQL_MODE_FOLD_COMBINE) != 0;
  h->fin_p = ReduceArgs{};
  PORL_TRY(policy_backward_impl(h, hp, (hipStream_t)stream, fold ? &h->fin_p : nullptr));
  h->fin_p_pending = fold;
  return PORL_OK;
}

// The whole update in one call.  Same arithmetic as the four phase calls; the two slab / partial-sum combines are
// folded into the Adam launches instead of running as launches of their own.
int porl_iql_step(porl_iql* h, const porl_iql_hyper* hp, void* stream) {
  PORL_TRY(check_ready(h, true)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  hipStream_t s = (hipStream_t)stream;
  ReduceArgs fin{};
  PORL_TRY(value_backward_impl(h, hp, s, g_iql_fold ? &fin : nullptr));
  g_phase = "V8.";
  PORL_TRY(adam_launch(h->buf.params_vf, h->buf.grads_vf, h->buf.adam_m_vf, h->buf.adam_v_vf, h->buf.params_tgt, h->n_vf,
                       hp->value_lr, hp->value_step, hp->adam_beta1, hp->adam_beta2, hp->adam_eps, hp->ema_beta, s, &fin));
  fin = ReduceArgs{};
  PORL_TRY(policy_backward_impl(h, hp, s, g_iql_fold ? &fin : nullptr));
  g_phase = "P9.";
  PORL_TRY(adam_launch(h->buf.params_pol, h->buf.grads_pol, h->buf.adam_m_pol, h->buf.adam_v_pol, nullptr, h->n_pol,
                       hp->policy_lr, hp->policy_step, hp->adam_beta1, hp->adam_beta2, hp->adam_eps, 0.0, s, &fin));
  g_phase = "";
  return PORL_OK;
}

// ---------------------------------------------------------------------------------------------------
static int pack_x(porl_iql* h, const float* x, int64_t x_rs, int batch, int64_t dst_off, hipStream_t s) {
  if (!x) PORL_FAIL(PORL_ERR_INVALID, "null input");
  if (batch < 1 || batch > h->cfg.max_batch) PORL_FAIL(PORL_ERR_INVALID, "batch %d outside [1,%d]", batch, h->cfg.max_batch);
  PackArgs a{};
  a.rows = batch; a.njobs = 1;
  a.job[0].src = x; a.job[0].dst = h->buf.workspace + dst_off; a.job[0].src_row_stride = x_rs;
  a.job[0].src_col_stride = 1; a.job[0].cols = h->cfg.obs_dim; a.job[0].ld = h->Sp;
  const long n = (long)batch * h->Sp;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024), 1), dim3(256), 0, s, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

// One Linear layer of up to 2 networks on a handful of rows (kernels.hpp: small_fwd_kernel).
static int small_fwd(int nnets, const float* const* Wt, const float* const* bias, const float* const* X, const long* ldx,
                     float* const* Y, const long* ldy, int N, int K, int B, int act, hipStream_t s) {
  SmallFwdArgs a{};
  for (int i = 0; i < nnets; ++i) { a.W[i] = Wt[i]; a.bias[i] = bias[i]; a.X[i] = X[i]; a.ldx[i] = ldx[i]; a.Y[i] = Y[i]; a.ldy[i] = ldy[i]; }
  a.N = N; a.K = K; a.B = B; a.act = act;
  const size_t lds = sizeof(float) * (size_t)B * K;
  if (lds > 64 * 1024) PORL_FAIL(PORL_ERR_UNSUPPORTED, "small-batch path: B*K too large");
  ProfScope ps("small_fwd_kernel", s, 2.0 * nnets * B * N * K, 4.0 * nnets * ((double)N * K + B * (N + K)));
  hipLaunchKernelGGL(small_fwd_kernel, dim3(std::min(cdiv(N, 4), 1024), nnets), dim3(256), lds, s, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iql_forward_value(porl_iql* h, int which, const float* x, int64_t x_rs, int32_t batch, float* v1_out,
                           float* v2_out, void* stream) {
  PORL_TRY(check_ready(h, false)); DevGuard _dg(h->device);
  if (!v1_out || !v2_out) PORL_FAIL(PORL_ERR_INVALID, "null output");
  hipStream_t s = (hipStream_t)stream;
  if (batch >= 1 && batch <= SMALL_FWD_MAX_B && !h->cfg.layer_norm && x) {
    // inference-sized batch: 3 GEMV launches straight from the caller's rows (no staging copy, no split-K)
    const int S = h->cfg.obs_dim, H = h->cfg.hidden_dim, L = h->cfg.n_hidden;
    float* W = h->buf.workspace;
    const float* P = which ? h->buf.params_tgt : h->buf.params_vf;
    const float* in[2] = {x, x};
    long ldin[2] = {(long)x_rs, (long)x_rs};
    for (int l = 0; l <= L; ++l) {
      const float* Wt[2] = {P + h->v[0].w[l], P + h->v[1].w[l]};
      const float* bs[2] = {P + h->v[0].b[l], P + h->v[1].b[l]};
      float* out[2];
      long ldo[2];
      for (int i = 0; i < 2; ++i) {
        out[i] = l == L ? (i ? v2_out : v1_out) : W + h->ws.act_t[i][l & 1];
        ldo[i] = l == L ? 1 : h->Hp;
      }
      PORL_TRY(small_fwd(2, Wt, bs, in, ldin, out, ldo, l == L ? 1 : H, l == 0 ? S : H, batch, l == L ? ACT_NONE : ACT_RELU, s));
      for (int i = 0; i < 2; ++i) { in[i] = out[i]; ldin[i] = ldo[i]; }
    }
    return PORL_OK;
  }
  // uses the s' staging buffer and the target scratch activations; invalidates a loaded minibatch
  PORL_TRY(pack_x(h, x, x_rs, batch, h->ws.xn, s));
  h->batch = 0;
  const int S = h->cfg.obs_dim, H = h->cfg.hidden_dim, L = h->cfg.n_hidden;
  float* W = h->buf.workspace;
  const float* P = which ? h->buf.params_tgt : h->buf.params_vf;
  int parts = 0;
  for (int l = 0; l < L; ++l) {
    FwdNet nets[2];
    for (int i = 0; i < 2; ++i) {
      FwdNet& f = nets[i];
      if (l == 0) { f.in = W + h->ws.xn; f.ldin = h->Sp; }
      else { f.in = W + h->ws.act_t[i][(l - 1) & 1]; f.ldin = h->Hp; }
      f.W = P + h->v[i].w[l]; f.b = P + h->v[i].b[l];
      const bool LN = h->cfg.layer_norm != 0;
      f.out = (l == L - 1 && !LN) ? nullptr : W + h->ws.act_t[i][l & 1];
      f.headw = P + h->v[i].w[L]; f.headout = W + h->ws.hp_t[i];
      if (LN) { f.ln_g = P + h->v[i].lnw[l]; f.ln_b = P + h->v[i].lnb[l]; }
    }
    PORL_TRY(fwd_hidden_layer(h, nets, 2, batch, l == 0 ? S : H, l == L - 1, &parts, s));
  }
  hipLaunchKernelGGL(head_finish_kernel, dim3(cdiv(batch, 256)), dim3(256), 0, s, W + h->ws.hp_t[0], W + h->ws.hp_t[1],
                     P + h->v[0].b[L], P + h->v[1].b[L], parts, batch, v1_out, v2_out);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iql_forward_policy(porl_iql* h, const float* x, int64_t x_rs, int32_t batch, float* mean_out,
                            int64_t mean_rs, void* stream) {
  PORL_TRY(check_ready(h, false)); DevGuard _dg(h->device);
  if (!mean_out) PORL_FAIL(PORL_ERR_INVALID, "null output");
  hipStream_t s = (hipStream_t)stream;
  if (batch >= 1 && batch <= SMALL_FWD_MAX_B && x) {
    const int S = h->cfg.obs_dim, H = h->cfg.hidden_dim, L = h->cfg.n_hidden, D = h->cfg.pol_out_dim;
    float* W = h->buf.workspace;
    const float* P = h->buf.params_pol;
    const float* in[1] = {x};
    long ldin[1] = {(long)x_rs};
    for (int l = 0; l <= L; ++l) {
      const float* Wt[1] = {P + h->pol.w[l]};
      const float* bs[1] = {P + h->pol.b[l]};
      // hidden activations go to the TARGET-net scratch: the policy-phase buffers may still be in use by a pipelined
      // update on another stream, the value-phase scratch is ordered on the caller's stream
      float* out[1] = {l == L ? mean_out : W + h->ws.act_t[0][l & 1]};
      long ldo[1] = {l == L ? (long)mean_rs : (long)h->Hp};
      const int act = l < L ? ACT_RELU : (h->cfg.pol_tanh ? ACT_TANH : ACT_NONE);
      PORL_TRY(small_fwd(1, Wt, bs, in, ldin, out, ldo, l == L ? D : H, l == 0 ? S : H, batch, act, s));
      in[0] = out[0]; ldin[0] = ldo[0];
    }
    return PORL_OK;
  }
  PORL_TRY(pack_x(h, x, x_rs, batch, h->ws.xn, s));
  h->batch = 0;
  const int S = h->cfg.obs_dim, H = h->cfg.hidden_dim, L = h->cfg.n_hidden, D = h->cfg.pol_out_dim;
  float* W = h->buf.workspace;
  const float* P = h->buf.params_pol;
  for (int l = 0; l < L; ++l) {
    FwdNet f{};
    if (l == 0) { f.in = W + h->ws.xn; f.ldin = h->Sp; }
    else { f.in = W + h->ws.act_p[l - 1]; f.ldin = h->Hp; }
    f.W = P + h->pol.w[l]; f.b = P + h->pol.b[l]; f.out = W + h->ws.act_p[l];
    PORL_TRY(fwd_hidden_layer(h, &f, 1, batch, l == 0 ? S : H, false, nullptr, s));
  }
  int nslab = 1;
  PORL_TRY(policy_mean_slabs(h, batch, &nslab, s));
  const long n = (long)batch * D;
  hipLaunchKernelGGL(mean_finish_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024)), dim3(256), 0, s,
                     W + h->ws.slab_mean, nslab, (long)batch * h->Dp, batch, D, h->Dp, P + h->pol.b[L],
                     h->cfg.pol_tanh, mean_out, (long)mean_rs);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

// ---------------------------------------------------------------------------------------------------
int porl_gemm_f32(int mode, int tile, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                  int32_t ldb, float* C, int32_t ldc, const float* bias, int act, const float* mask, int32_t ldmask,
                  int splitk, float* slab, void* stream) {
  g_short_blocks = false;
  if (mode < 0 || mode > 2) PORL_FAIL(PORL_ERR_INVALID, "mode must be 0..2");
  if (M < 1 || N < 1 || K < 0 || !A || !B || !C) PORL_FAIL(PORL_ERR_INVALID, "bad GEMM arguments");
  if (splitk > 1 && !slab) PORL_FAIL(PORL_ERR_INVALID, "splitk > 1 needs a slab buffer");
  if (splitk > 1 && ldc != N) PORL_FAIL(PORL_ERR_INVALID, "splitk > 1 needs a dense C (ldc == N)");
  DevGuard _dg(device_of(C));
  hipStream_t s = (hipStream_t)stream;
  GemmGroup g{};
  g.nprob = 1;
  g.p[0] = make_prob(mode, A, lda, B, ldb, C, ldc, M, N, K);
  if (tile < 0) tile = pick_tile(g);
  if (tile >= TILE_COUNT) PORL_FAIL(PORL_ERR_INVALID, "tile must be -1..%d", TILE_COUNT - 1);
  if (splitk > 1) {
    g.p[0].splitk = splitk; g.p[0].C = slab;
    PORL_TRY(launch_group(g, tile, s));
    // rows of the slab have stride ldc; combine all M*ldc floats (padding columns carry garbage that the
    // caller's ldc padding tolerates), then apply bias/act/mask per element
    if (mask) PORL_FAIL(PORL_ERR_UNSUPPORTED, "mask with splitk");
    const long n = (long)M * ldc;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024)), dim3(256), 0, s, C, slab,
                       splitk, n, n, bias, ldc, act);
    PORL_HIP(hipGetLastError());
    return PORL_OK;
  }
  g.p[0].bias = bias; g.p[0].act = act; g.p[0].mask = mask; g.p[0].ldmask = ldmask;
  if (g_dbg_a_scale && g_dbg_a_shift && mode == GEMM_NT) {
    g.p[0].apro = APRO_AFFINE_RELU; g.p[0].a_colscale = g_dbg_a_scale; g.p[0].a_colshift = g_dbg_a_shift;
  }
  if (g_dbg_resid) g.p[0].resid = g_dbg_resid;
  if (g_dbg_cstat) g.p[0].cstat = g_dbg_cstat;
  return launch_group(g, tile, s);
}

int porl_adam_ema(float* p, const float* g, float* m, float* v, float* target, int64_t n, double lr, int32_t step,
                  double beta1, double beta2, double eps, double ema_beta, void* stream) {
  if (!p || !g || !m || !v) PORL_FAIL(PORL_ERR_INVALID, "null buffer");
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || (target && !aligned16(target)))
    PORL_FAIL(PORL_ERR_INVALID, "buffers must be 16-byte aligned");
  DevGuard _dg(device_of(p));
  return adam_launch(p, const_cast<float*>(g), m, v, target, n, lr, step, beta1, beta2, eps, ema_beta, (hipStream_t)stream);
}

int porl_ema(float* target, const float* source, int64_t n, double ema_beta, void* stream) {
  if (!target || !source || n < 0 || n % 4) PORL_FAIL(PORL_ERR_INVALID, "need buffers and a multiple of 4 floats");
  if (!aligned16(target) || !aligned16(source)) PORL_FAIL(PORL_ERR_INVALID, "buffers must be 16-byte aligned");
  if (n == 0) return PORL_OK;
  DevGuard _dg(device_of(target));
  const long n4 = n / 4;
  ProfScope ps("ema_kernel", (hipStream_t)stream, 0.0, 12.0 * n);
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, target, source, n4,
                     (float)(1.0 - ema_beta), (float)ema_beta);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_reduce_mean(const float* x, int32_t n, float* out, void* stream) {
  if (!x || !out || n < 1) PORL_FAIL(PORL_ERR_INVALID, "bad arguments");
  DevGuard _dg(device_of(out));
  ReduceArgs r{};
  add_reduce(r, out, x, 1, 1, n, 0, 1.0f / n);          // fixed summation order (multi_reduce_kernel)
  return launch_reduce(r, (hipStream_t)stream);
}

int porl_softmax_mask(const float* logits, int64_t ld, int32_t batch, int32_t n_actions, float threshold,
                      int32_t write_probs, float* mask_out, void* stream) {
  if (!logits || !mask_out || batch < 1 || n_actions < 1 || n_actions > 4096 || ld < n_actions)
    PORL_FAIL(PORL_ERR_INVALID, "bad softmax-mask arguments");
  DevGuard _dg(device_of(mask_out));
  hipLaunchKernelGGL(softmax_mask_kernel, dim3(cdiv(batch, 256)), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, batch,
                     n_actions, threshold, write_probs, mask_out);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_gather_rows(const float* rows, int64_t row_stride, const int64_t* idx, int32_t n, int32_t width, float* out,
                     int64_t out_stride, void* stream) {
  if (!rows || !idx || !out || n < 0 || width < 1) PORL_FAIL(PORL_ERR_INVALID, "bad gather arguments");
  if (n == 0) return PORL_OK;
  DevGuard _dg(device_of(out));
  const int blocks = std::min(cdiv(n, 4), 2048);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, (long)row_stride, idx, n,
                     width, out, (long)out_stride);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_sample_indices(int64_t n_rows, int32_t batch, uint64_t seed, uint64_t step, int64_t base, int64_t* out,
                        void* stream) {
  if (n_rows < 1 || batch < 1 || batch > n_rows || !out) PORL_FAIL(PORL_ERR_INVALID, "need 1 <= batch <= n_rows");
  if (n_rows > (int64_t(1) << 40)) PORL_FAIL(PORL_ERR_INVALID, "n_rows too large");
  const int hb = feistel_half_bits(n_rows);
  DevGuard _dg(device_of(out));
  hipLaunchKernelGGL(sample_indices_kernel, dim3(cdiv(batch, 256)), dim3(256), 0, (hipStream_t)stream, n_rows, batch,
                     seed, step, hb, base, (int64_t)0, out);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_epoch_indices(int64_t n_rows, int64_t first, int32_t count, uint64_t seed, uint64_t epoch, int64_t base,
                       int64_t* out, void* stream) {
  if (n_rows < 1 || count < 1 || first < 0 || first + count > n_rows || !out)
    PORL_FAIL(PORL_ERR_INVALID, "need 0 <= first, 1 <= count, first + count <= n_rows");
  if (n_rows > (int64_t(1) << 40)) PORL_FAIL(PORL_ERR_INVALID, "n_rows too large");
  const int hb = feistel_half_bits(n_rows);
  DevGuard _dg(device_of(out));
  hipLaunchKernelGGL(sample_indices_kernel, dim3(cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream, n_rows, count,
                     seed, epoch, hb, base, first, out);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

// ---- prioritized replay: sum tree in HBM (per_tree.hpp) ------------------------------------------------------
int porl_per_update(double* tree, int64_t capacity, const int64_t* tree_idx, const double* td_error, int32_t n, double eps,
                    double alpha, int32_t* stamp, void* stream) {
  if (!tree || !tree_idx || !td_error || !stamp || capacity < 1 || n < 0) PORL_FAIL(PORL_ERR_INVALID, "bad arguments");
  if (n == 0) return PORL_OK;
  DevGuard _dg(device_of(tree));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(per_stamp_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, tree_idx, n, capacity, stamp);
  hipLaunchKernelGGL(per_set_leaves_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, tree, tree_idx, td_error, n, capacity, eps,
                     alpha, stamp);
  int levels = 0;
  for (int64_t v = 2 * capacity - 1; v > 1; v >>= 1) ++levels;          // depth of the deepest leaf
  hipLaunchKernelGGL(per_propagate_kernel, dim3(1), dim3(1024), 0, s, tree, tree_idx, n, levels);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_per_sample(const double* tree, int64_t capacity, const double* u, int32_t batch, int64_t n_entries, double beta,
                    int64_t* out_idx, double* out_prio, float* out_w, void* stream) {
  if (!tree || !u || !out_idx || !out_prio || !out_w || capacity < 1 || batch < 1 || n_entries < 1)
    PORL_FAIL(PORL_ERR_INVALID, "bad arguments");
  DevGuard _dg(device_of(tree));
  PerSampleArgs a{tree, capacity, u, batch, n_entries, beta, out_idx, out_prio, out_w};
  hipLaunchKernelGGL(per_sample_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_tune_set(const char* key, int value) {
  if (!key) PORL_FAIL(PORL_ERR_INVALID, "null key");
  if (!strcmp(key, "gemm_lds_pad")) { gemm_lds_pad() = std::max(0, value); return PORL_OK; }
  if (!strcmp(key, "gemm_lds_pad_min_blocks")) { gemm_lds_pad_min_blocks() = std::max(0, value); return PORL_OK; }
  if (!strcmp(key, "iql_pad_value")) { g_iql_pad_value = std::max(0, value); return PORL_OK; }
  if (!strcmp(key, "iql_pad_policy")) { g_iql_pad_policy = std::max(0, value); return PORL_OK; }
  if (!strcmp(key, "iql_pad_min_blocks")) { g_iql_pad_min_blocks = std::max(0, value); return PORL_OK; }
  if (!strcmp(key, "iql_pad_min_k")) { g_iql_pad_min_k = std::max(0, value); return PORL_OK; }
  if (!strcmp(key, "qnet_fused")) { g_qnet_fused = value != 0; return PORL_OK; }
  if (!strcmp(key, "qnet_two_groups")) { g_qnet_two_groups = value != 0; return PORL_OK; }
  if (!strcmp(key, "qnet_wgrad_share")) { g_qnet_wgrad_share = std::max(1, std::min(value, 16)); return PORL_OK; }
  if (!strcmp(key, "qnet_rows16")) { g_qnet_rows16 = value != 0; return PORL_OK; }
  if (!strcmp(key, "enc_dense_patch")) { g_enc_dense_patch = value != 0; return PORL_OK; }
  if (!strcmp(key, "enc_s2d")) { g_enc_s2d = value != 0; return PORL_OK; }
  if (!strcmp(key, "enc_patch_rows")) { g_enc_patch_rows = std::max(0, value); return PORL_OK; }
  if (!strcmp(key, "enc_gemm_sb")) { g_enc_gemm_sb = value; return PORL_OK; }
  if (!strcmp(key, "enc_tile_n96")) { g_enc_tile_n96 = value != 0; return PORL_OK; }
  if (!strcmp(key, "enc_bn_sweep")) { g_enc_bn_sweep = value != 0; return PORL_OK; }
  if (!strcmp(key, "enc_pconv_mfma")) { g_enc_pconv_mfma = value != 0; return PORL_OK; }
  if (!strcmp(key, "enc_bf16_operands_only")) { g_enc_bf16_operands_only = value != 0; return PORL_OK; }
  if (!strcmp(key, "iql_fold")) { g_iql_fold = value != 0; return PORL_OK; }
  if (!strcmp(key, "skinny")) { g_skinny = value == 1 ? 7 : value; return PORL_OK; }
  if (!strcmp(key, "skinny_pipelined")) { g_skinny_pipelined = value == 1 ? 7 : value; return PORL_OK; }     // -1: by width
  if (!strcmp(key, "skinny_pipelined_max_h")) { g_skinny_pipelined_max_h = value; return PORL_OK; }
  if (!strcmp(key, "dw0_slabs")) { g_dw0_slabs = std::max(1, std::min(value, SK_MAX)); return PORL_OK; }
  if (!strcmp(key, "l0_tile")) { g_l0_tile = value; return PORL_OK; }
  if (!strcmp(key, "l0_kernel")) { g_l0_kernel = value != 0; return PORL_OK; }
  if (!strcmp(key, "vbwd_tile_short")) { g_vbwd_tile_short = value; return PORL_OK; }
  if (!strncmp(key, "tile_map_short", 14) && key[14] >= '0' && key[14] <= '3' && !key[15] && value >= 0 && value < TILE_COUNT) {
    g_tile_map_short[key[14] - '0'] = value;
    return PORL_OK;
  }
  if (!strncmp(key, "tile_map", 8) && key[8] >= '0' && key[8] <= '3' && !key[9] && value >= 0 && value < TILE_COUNT) {
    g_tile_map[key[8] - '0'] = value;
    return PORL_OK;
  }
  PORL_FAIL(PORL_ERR_INVALID, "unknown tuning key '%s'", key);
}

// ---- stream signals: cross-stream ordering by 64-bit counters in signal memory ----------------------------------------
// (hipStreamWriteValue64 / hipStreamWaitValue64 on memory from hipExtMallocWithFlags(hipMallocSignalMemory): the command
// processor writes / polls a value in stream order.)  The pipelined update orders its two streams three times per
// update; as event record + stream-wait-event pairs each crossing showed up as 6-12 us of idle queue in the kernel
// trace even when the event had completed long before.
int porl_signal_create(void** out) {
  if (!out) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  int dev = 0, can = 0;
  PORL_HIP(hipGetDevice(&dev));
  PORL_HIP(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
  if (!can) PORL_FAIL(PORL_ERR_UNSUPPORTED, "stream wait-value operations are not supported on this device");
  void* p = nullptr;
  PORL_HIP(hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory));
  PORL_HIP(hipMemset(p, 0, 8));
  *out = p;
  return PORL_OK;
}
int porl_signal_destroy(void* sig) {
  if (sig) PORL_HIP(hipFree(sig));
  return PORL_OK;
}
int porl_signal_write(void* sig, uint64_t value, void* stream) {
  if (!sig) PORL_FAIL(PORL_ERR_INVALID, "null signal");
  PORL_HIP(hipStreamWriteValue64((hipStream_t)stream, sig, value, 0));
  return PORL_OK;
}
int porl_signal_wait_ge(void* sig, uint64_t value, void* stream) {
  if (!sig) PORL_FAIL(PORL_ERR_INVALID, "null signal");
  PORL_HIP(hipStreamWaitValue64((hipStream_t)stream, sig, value, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
  return PORL_OK;
}

// One pipelined update from ONE call: what porl_amd/agent/_iql.py:_full_update issues in signal mode on a single GPU —
// value phase on `main_stream`, policy phase on `side_stream`, ordered by the three counters — without a round trip
// through the caller's language per phase (the Python sequence costs ~110 us of host time per update, which is what
// bounds small configurations; this entry is ~20 launches back to back).  Same launches, same order per stream, same
// results as the phase calls.
int porl_iql_update_pipelined(porl_iql* h, const porl_iql_hyper* hp, int32_t batch, const float* rows, int64_t row_stride,
                              int64_t n_rows, int32_t act_dim, int32_t target_is_action, uint64_t seed, uint64_t step,
                              void* sig_value, void* sig_fwd, void* sig_policy, uint64_t seq, uint64_t wait_policy_seq,
                              uint64_t wait_fwd_seq, int32_t write_policy, void* main_stream, void* side_stream) {
  PORL_TRY(check_ready(h, false)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  if (!sig_value || !sig_fwd || !sig_policy || seq < 1) PORL_FAIL(PORL_ERR_INVALID, "three signal counters and seq >= 1 are required");
  if (main_stream == side_stream) PORL_FAIL(PORL_ERR_INVALID, "the two phases need two streams");
  if (!(h->mode & PORL_IQL_MODE_TWO_SLOTS) || !(h->mode & PORL_IQL_MODE_FOLD_COMBINE))
    PORL_FAIL(PORL_ERR_INVALID, "pipelined updates need PORL_IQL_MODE_TWO_SLOTS | PORL_IQL_MODE_FOLD_COMBINE");
  // reject bad minibatch arguments before anything is enqueued (porl_iql_load_batch_sampled checks them again)
  if (batch < 1 || batch > h->cfg.max_batch) PORL_FAIL(PORL_ERR_INVALID, "batch %d outside [1,%d]", batch, h->cfg.max_batch);
  if (!rows || n_rows < batch || n_rows > (int64_t(1) << 40) || act_dim < 0 ||
      row_stride < 2 * (int64_t)h->cfg.obs_dim + 2 + act_dim)
    PORL_FAIL(PORL_ERR_INVALID, "bad replay rows (need batch <= n_rows <= 2^40, row stride >= 2*S+2+A)");
  if (target_is_action ? h->cfg.pol_out_dim != act_dim : h->cfg.pol_out_dim != h->cfg.obs_dim)
    PORL_FAIL(PORL_ERR_INVALID, "policy target width mismatch");
  // the staging slot loaded next was last read by the policy phase PORL_IQL_SLOTS updates ago
  if (wait_policy_seq) PORL_TRY(porl_signal_wait_ge(sig_policy, wait_policy_seq, main_stream));
  PORL_TRY(porl_iql_load_batch_sampled(h, batch, rows, row_stride, n_rows, act_dim, target_is_action, seed, step, nullptr, main_stream));
  PORL_TRY(porl_iql_value_backward(h, hp, main_stream));
  // the previous update's policy phase has read the old value nets
  if (wait_fwd_seq) PORL_TRY(porl_signal_wait_ge(sig_fwd, wait_fwd_seq, main_stream));
  PORL_TRY(porl_iql_value_apply(h, hp, main_stream));
  PORL_TRY(porl_signal_write(sig_value, seq, main_stream));
  PORL_TRY(porl_signal_wait_ge(sig_value, seq, side_stream));
  PORL_TRY(porl_iql_policy_forward(h, hp, side_stream));
  PORL_TRY(porl_signal_write(sig_fwd, seq, side_stream));
  PORL_TRY(porl_iql_policy_backward(h, hp, side_stream));
  PORL_TRY(porl_iql_policy_apply(h, hp, side_stream));
  if (write_policy) PORL_TRY(porl_signal_write(sig_policy, seq, side_stream));
  return PORL_OK;
}

int porl_tune_set_ptr(const char* key, void* ptr) {
  if (!key) PORL_FAIL(PORL_ERR_INVALID, "null key");
  if (!strcmp(key, "qnet_stamps")) { g_qnet_stamps = (unsigned long long*)ptr; return PORL_OK; }
  if (!strcmp(key, "gemm_a_scale")) { g_dbg_a_scale = (const float*)ptr; return PORL_OK; }
  if (!strcmp(key, "gemm_a_shift")) { g_dbg_a_shift = (const float*)ptr; return PORL_OK; }
  if (!strcmp(key, "gemm_resid")) { g_dbg_resid = (const float*)ptr; return PORL_OK; }
  if (!strcmp(key, "gemm_cstat")) { g_dbg_cstat = (float*)ptr; return PORL_OK; }
  PORL_FAIL(PORL_ERR_INVALID, "unknown tuning key '%s'", key);
}

int porl_state2costmap(float* state, int64_t state_rs, int32_t batch, int32_t n_ang, int32_t n_dist, float* out,
                        void* stream) {
  if (!state || !out || batch < 1 || n_ang < 4 || n_dist < 4) PORL_FAIL(PORL_ERR_INVALID, "bad costmap arguments");
  if (batch > 65535) PORL_FAIL(PORL_ERR_INVALID, "batch > 65535");
  DevGuard _dg(device_of(out));
  hipStream_t s = (hipStream_t)stream;
  // constants exactly as util/costmap.py:19-20,34,45 forms them (python doubles rounded to fp32 at the tensor op)
  const double pi = 3.14159265358979323846;
  const float dist_inc = (float)((4.0 + 1e-4) / n_dist);
  const float ang_inc = (float)((2.0 * pi + 1e-4) / n_ang);
  const float deg_min = (float)(-pi + (2.0 * pi + 2e-4) / n_ang), deg_max = (float)(pi - (2.0 * pi + 2e-4) / n_ang);
  const float dist_max = (float)(4.0 - 4.0 / n_dist);
  hipLaunchKernelGGL(costmap_kernel, dim3(n_ang, batch), dim3(std::min(256, n_dist)), 0, s, state, (long)state_rs, n_ang,
                     n_dist, dist_inc, ang_inc, deg_min, deg_max, dist_max, out);
  PORL_HIP(hipGetLastError());
  const long n = (long)batch * (n_ang + 2);
  hipLaunchKernelGGL(clamp_gt8_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024)), dim3(256), 0, s, state,
                     (long)state_rs, n_ang + 2, batch);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_prof_enable(int on) {
  g_prof.on = on != 0;
  g_prof.recs.clear();
  g_prof.pool_used = 0;
  return PORL_OK;
}

int porl_prof_read(porl_prof_entry* out, int max_entries) {
  if (!out || max_entries < 1) PORL_FAIL(PORL_ERR_INVALID, "bad profile buffer");
  PORL_HIP(hipDeviceSynchronize());
  const int nl = std::min<int>((int)g_prof.labels.size(), max_entries);
  for (int i = 0; i < nl; ++i) {
    memset(&out[i], 0, sizeof(out[i]));
    strncpy(out[i].name, g_prof.labels[i].c_str(), sizeof(out[i].name) - 1);
  }
  for (const ProfRec& r : g_prof.recs) {
    if (r.label >= nl) continue;
    float ms = 0.f;
    PORL_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    out[r.label].launches += 1;
    out[r.label].total_ms += ms;
    out[r.label].flops += r.flops;
    out[r.label].bytes += r.bytes;
  }
  return nl;
}

}  // extern "C"

// =====================================================================================================
// Discrete-action CQL engine (QNetwork S -> hidden... -> A; src/porl/net/q_network.py:8-30)
// =====================================================================================================
struct porl_qnet {
  porl_qnet_cfg cfg;
  MlpLayout net;
  int64_t n_params = 0;
  std::vector<TensorInfo> tensors;
  porl_qnet_buffers buf{};
  bool bound = false;
  int device = -1;
  int batch = 0;
  int Sp = 0, Ap = 0, ld[PORL_MAX_HIDDEN + 2] = {0};     // padded leading dims per layer output
  int wld[PORL_MAX_HIDDEN + 1] = {0};                    // row stride of layer l's weight image (floats)
  struct {
    int64_t xs, xn, rew, done, actions;                  // actions: int64 stored in 2 floats each
    int64_t act[PORL_MAX_HIDDEN + 1], tmp[2], dz[2], slab, part_td, part_pen, fslab, total;
  } ws;
  // one-launch path (qnet_fused.hpp): every width <= 128, at most QF_MAX_LIN Linear layers, LDS plan fits
  bool fused_ok = false;
  QnetFusedArgs fargs{};
  int fused_lds_bytes = 0;
  int fused2_lds_w2 = 0, fused2_lds_bytes = 0;       // two-group kernel: offset of the second weight image; 0 = does not fit
  QnetFusedArgs fargs16{};                           // the same plan for 16 rows per block (qnet_fused2_kernel<16>)
  int fused16_lds_w2 = 0, fused16_lds_bytes = 0;
  int64_t fslab_stride = 0;
  bool fslab_clean = false;          // alignment gaps of the flat layout are never written: zeroed once
  bool slab_clean = false;           // same for the split-K slabs of the multi-launch path
};


extern "C" {

int porl_qnet_create(const porl_qnet_cfg* c, porl_qnet** out) {
  if (!c || !out) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  if (c->state_dim < 1 || c->n_actions < 1 || c->max_batch < 1) PORL_FAIL(PORL_ERR_INVALID, "dimensions must be positive");
  if (c->n_hidden < 1 || c->n_hidden > PORL_MAX_HIDDEN) PORL_FAIL(PORL_ERR_INVALID, "n_hidden must be in [1,%d]", PORL_MAX_HIDDEN);
  if (c->n_actions > 4096) PORL_FAIL(PORL_ERR_UNSUPPORTED, "more than 4096 outputs");   // (A x N outputs of the distributional nets)
  porl_qnet* h = new porl_qnet();
  h->cfg = *c;
  const int L = c->n_hidden, B = c->max_batch;
  MlpLayout& m = h->net;
  m.n_lin = L + 1;
  m.dims[0] = c->state_dim;
  for (int l = 0; l < L; ++l) {
    if (c->hidden[l] < 1) { delete h; PORL_FAIL(PORL_ERR_INVALID, "hidden size must be positive"); }
    m.dims[l + 1] = c->hidden[l];
  }
  m.dims[L + 1] = c->n_actions;
  // Parameter layout = the image the one-launch step kernel parks in LDS (csrc/qnet_fused.hpp): layer l is
  // round32(out) rows of (round16(in) + 4) floats — the (out, in) weights in the top-left corner, zeros elsewhere —
  // directly followed by round32(out) bias floats.  A block stages a layer with one linear, fully coalesced copy:
  // no per-element address arithmetic, no bounds selects.  Zeros stay zeros under Adam (their gradient is always 0).
  int64_t cur = 0;
  for (int l = 0; l <= L; ++l) {
    const int rows = (m.dims[l + 1] + 31) & ~31;
    h->wld[l] = ((m.dims[l] + 15) & ~15) + 4;
    m.w[l] = cur;
    h->tensors.push_back({cur, m.dims[l + 1], m.dims[l]});
    cur += (int64_t)rows * h->wld[l];
    m.b[l] = cur;
    h->tensors.push_back({cur, 0, m.dims[l + 1]});
    cur += rows;
  }
  h->n_params = cur;
  h->Sp = (int)ru4(c->state_dim);
  h->Ap = (int)ru4(c->n_actions);
  int maxld = h->Ap;
  for (int l = 0; l <= L; ++l) { h->ld[l] = (int)ru4(m.dims[l + 1]); maxld = std::max(maxld, h->ld[l]); }
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += ru4(n); return r; };
  h->ws.xs = take((int64_t)B * h->Sp); h->ws.xn = take((int64_t)B * h->Sp);
  h->ws.rew = take(B); h->ws.done = take(B); h->ws.actions = take(2 * (int64_t)B);
  for (int l = 0; l <= L; ++l) h->ws.act[l] = take((int64_t)B * h->ld[l]);
  h->ws.tmp[0] = take((int64_t)B * maxld); h->ws.tmp[1] = take((int64_t)B * maxld);
  h->ws.dz[0] = take((int64_t)B * maxld); h->ws.dz[1] = take((int64_t)B * maxld);
  h->ws.slab = take((int64_t)SK_MAX * (cur + 64));
  const int nblk = cdiv(B, 16);                          // (16-row blocks: the finest partition any step kernel uses)
  h->ws.part_td = take(nblk); h->ws.part_pen = take(nblk);
  {
    // LDS plan of the fused kernel
    bool ok = (L + 1 <= QF_MAX_LIN);
    int maxw = 0, wmax = 0;
    for (int l = 0; l <= L + 1 && ok; ++l) { ok = m.dims[l] <= QF_MAX_W; maxw = std::max(maxw, (m.dims[l] + 31) & ~31); }
    QnetFusedArgs& fa = h->fargs;
    int off = 0;
    if (ok) {
      fa.n_lin = L + 1;
      for (int l = 0; l <= L + 1; ++l) {
        fa.dims[l] = m.dims[l];
        fa.lds_act[l] = off;
        off += QF_ROWS * (((m.dims[l] + 31) & ~31) + 4);
      }
      for (int l = 0; l <= L; ++l) {
        fa.w_off[l] = m.w[l]; fa.b_off[l] = m.b[l];
        wmax = std::max(wmax, ((m.dims[l + 1] + 31) & ~31) * (((m.dims[l] + 15) & ~15) + 4 + 1));   // image + bias row
      }
      fa.lds_tmp[0] = off; off += QF_ROWS * (maxw + 4);
      fa.lds_tmp[1] = off; off += QF_ROWS * (maxw + 4);
      fa.lds_w = off; off += wmax;
      ok = off * (int)sizeof(float) <= QF_MAX_LDS_BYTES;
      if (ok && (off + wmax) * (int)sizeof(float) <= QF_MAX_LDS_BYTES) {
        h->fused2_lds_w2 = off;
        h->fused2_lds_bytes = (off + wmax) * (int)sizeof(float);
      }
    }
    if (ok && h->fused2_lds_w2 > 0) {
      // 16-row plan: same buffers, half the rows
      QnetFusedArgs& f16 = h->fargs16;
      f16 = fa;
      int o16 = 0;
      for (int l = 0; l <= L + 1; ++l) { f16.lds_act[l] = o16; o16 += 16 * (((m.dims[l] + 31) & ~31) + 4); }
      f16.lds_tmp[0] = o16; o16 += 16 * (maxw + 4);
      f16.lds_tmp[1] = o16; o16 += 16 * (maxw + 4);
      f16.lds_w = o16; o16 += wmax;
      h->fused16_lds_w2 = o16;
      h->fused16_lds_bytes = (o16 + wmax) * (int)sizeof(float);
    }
    h->fused_ok = ok;
    h->fused_lds_bytes = off * (int)sizeof(float);
    h->fslab_stride = ru4(cur);
    h->ws.fslab = ok ? take((int64_t)nblk * h->fslab_stride) : 0;
  }
  h->ws.total = o;
  *out = h;
  return PORL_OK;
}

void porl_qnet_destroy(porl_qnet* h) { delete h; }
int64_t porl_qnet_param_floats(const porl_qnet* h) { return h ? h->n_params : 0; }
int32_t porl_qnet_tensors(const porl_qnet* h) { return h ? (int32_t)h->tensors.size() : 0; }
int porl_qnet_tensor_info(const porl_qnet* h, int index, int64_t* offset, int32_t* rows, int32_t* cols, int32_t* row_stride) {
  if (!h || index < 0 || index >= (int)h->tensors.size()) PORL_FAIL(PORL_ERR_INVALID, "tensor index out of range");
  if (offset) *offset = h->tensors[index].off;
  if (rows) *rows = h->tensors[index].rows;
  if (cols) *cols = h->tensors[index].cols;
  if (row_stride) *row_stride = h->tensors[index].rows ? h->wld[index / 2] : 1;
  return PORL_OK;
}
int64_t porl_qnet_workspace_floats(const porl_qnet* h) { return h ? h->ws.total : 0; }
int32_t porl_qnet_one_launch(const porl_qnet* h) { return h && h->fused_ok && g_qnet_fused ? 1 : 0; }

int porl_qnet_bind(porl_qnet* h, const porl_qnet_buffers* b) {
  if (!h || !b) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  const void* ptrs[] = {b->params, b->params_tgt, b->grads, b->adam_m, b->adam_v, b->workspace, b->stats};
  for (const void* p : ptrs) {
    if (!p) PORL_FAIL(PORL_ERR_INVALID, "null buffer");
    if (!aligned16(p)) PORL_FAIL(PORL_ERR_INVALID, "buffers must be 16-byte aligned");
  }
  h->buf = *b;
  h->bound = true;
  h->device = device_of(b->workspace);
  h->batch = 0;
  h->fslab_clean = false;
  h->slab_clean = false;
  return PORL_OK;
}

static int qnet_ready(const porl_qnet* h, bool need_batch) {
  g_short_blocks = false;
  if (!h) PORL_FAIL(PORL_ERR_INVALID, "null engine");
  if (!h->bound) PORL_FAIL(PORL_ERR_UNBOUND, "porl_qnet_bind() has not been called");
  if (need_batch && h->batch <= 0) PORL_FAIL(PORL_ERR_INVALID, "no minibatch loaded (porl_qnet_load_batch)");
  return 0;
}

int porl_qnet_load_batch(porl_qnet* h, int32_t batch, const float* states, int64_t s_rs, const int64_t* actions,
                         int64_t a_rs, const float* rewards, int64_t r_rs, const float* next_states, int64_t n_rs,
                         const float* dones, int64_t d_rs, void* stream) {
  PORL_TRY(qnet_ready(h, false)); DevGuard _dg(h->device);
  if (batch < 1 || batch > h->cfg.max_batch) PORL_FAIL(PORL_ERR_INVALID, "batch %d outside [1,%d]", batch, h->cfg.max_batch);
  if (!states) PORL_FAIL(PORL_ERR_INVALID, "null states");
  float* W = h->buf.workspace;
  hipStream_t s = (hipStream_t)stream;
  PackArgs a{};
  a.rows = batch;
  auto job = [&](const float* src, int64_t rs, float* dst, int cols, int ld) {
    PackJob& j = a.job[a.njobs++];
    j.src = src; j.dst = dst; j.src_row_stride = rs; j.src_col_stride = 1; j.cols = cols; j.ld = ld;
  };
  job(states, s_rs, W + h->ws.xs, h->cfg.state_dim, h->Sp);
  if (next_states) job(next_states, n_rs, W + h->ws.xn, h->cfg.state_dim, h->Sp);
  if (rewards) job(rewards, r_rs, W + h->ws.rew, 1, 1);
  if (dones) job(dones, d_rs, W + h->ws.done, 1, 1);
  const long n = (long)batch * h->Sp;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024), a.njobs), dim3(256), 0, s, a);
  PORL_HIP(hipGetLastError());
  if (actions) {
    hipLaunchKernelGGL(pack_i64_kernel, dim3(cdiv(batch, 256)), dim3(256), 0, s, actions, (long)a_rs, batch,
                       reinterpret_cast<int64_t*>(W + h->ws.actions));
    PORL_HIP(hipGetLastError());
  }
  h->batch = batch;
  return PORL_OK;
}

// forward of `nnets` parameter sets; set k reads input in_k and writes hidden activations to dst_k[l]
static int qnet_forward(porl_qnet* h, int nnets, const float* const* params, const float* const* inputs,
                        float* const (*dst)[PORL_MAX_HIDDEN + 1], int B, hipStream_t s) {
  const int L = h->cfg.n_hidden;
  for (int l = 0; l <= L; ++l) {
    GemmGroup g{};
    g.nprob = nnets;
    const int K = h->net.dims[l], Nn = h->net.dims[l + 1];
    for (int k = 0; k < nnets; ++k) {
      const float* in = l == 0 ? inputs[k] : dst[k][l - 1];
      const int ldin = l == 0 ? h->Sp : h->ld[l - 1];
      GemmProb p = make_prob(GEMM_NT, in, ldin, params[k] + h->net.w[l], h->wld[l], dst[k][l], h->ld[l], B, Nn, K);
      p.bias = params[k] + h->net.b[l];
      p.act = l < L ? ACT_RELU : ACT_NONE;
      g.p[k] = p;
    }
    PORL_TRY(launch_group(g, pick_tile(g), s));
  }
  return PORL_OK;
}

// gradient of one minibatch in two launches: the fused step kernel (32 rows per block), then the block-order
// sum of the partial gradients (+ loss statistics)
struct QnetSampling { int64_t n_rows = 0; uint64_t seed = 0, step = 0; };

static int qnet_fused_backward(porl_qnet* h, const porl_qnet_hyper* hp, int B, const float* states, int64_t s_rs,
                               const float* next_states, int64_t n_rs, const int64_t* actions, const float* rew,
                               const float* done, const int64_t* idx, hipStream_t s, bool with_adam = false,
                               const porl_qnet_variant* var = nullptr, const QnetSampling* samp = nullptr) {
  float* W = h->buf.workspace;
  QnetFusedArgs a = h->fargs;
  a.params = h->buf.params; a.params_tgt = h->buf.params_tgt;
  a.states = states; a.s_rs = s_rs; a.next_states = next_states; a.n_rs = n_rs;
  a.actions = actions; a.rew = rew; a.done = done; a.idx = idx;
  a.slab = W + h->ws.fslab; a.slab_stride = h->fslab_stride;
  a.part_td = W + h->ws.part_td; a.part_pen = W + h->ws.part_pen;
  a.B = B;
  a.gamma = hp->gamma; a.alpha = hp->alpha; a.inv_batch = hp->inv_batch;
  a.log_A = (float)std::log((double)h->cfg.n_actions);
  a.stamps = g_qnet_stamps;
  if (var) {
    a.double_dqn = var->double_dqn; a.is_w = var->is_weights; a.w_uniform = var->uniform_weight; a.td_abs = var->td_abs;
    a.next_mask = var->next_mask; a.td_off = var->td_off;
  }
  static bool attr_set = false;
  if (!attr_set) {
    PORL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 QF_MAX_LDS_BYTES));
    PORL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_fused2_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 QF_MAX_LDS_BYTES));
    PORL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_fused2_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 QF_MAX_LDS_BYTES));
    attr_set = true;
  }
  // 16 rows per block while 32-row blocks would leave CUs idle (config 3 at B = 4096: 128 blocks on 256 CUs)
  const bool two = g_qnet_two_groups && h->fused2_lds_w2 > 0;
  if (samp && samp->n_rows > 0) {
    if (!two) PORL_FAIL(PORL_ERR_UNSUPPORTED, "in-kernel sampling needs the two-group step kernel");
    a.samp_n = samp->n_rows; a.samp_seed = samp->seed; a.samp_step = samp->step; a.samp_hb = feistel_half_bits(samp->n_rows);
    a.idx = nullptr;
  }
  const bool rows16 = two && g_qnet_rows16 && h->fused16_lds_w2 > 0 && cdiv(B, QF_ROWS) < NUM_CU;
  a.wgrad_share = g_qnet_wgrad_share;
  const int nblk = cdiv(B, rows16 ? 16 : QF_ROWS);
  if (rows16) {
    const QnetFusedArgs& f16 = h->fargs16;
    for (int l = 0; l <= QF_MAX_LIN; ++l) a.lds_act[l] = f16.lds_act[l];
    a.lds_tmp[0] = f16.lds_tmp[0]; a.lds_tmp[1] = f16.lds_tmp[1]; a.lds_w = f16.lds_w;
  }
  if (!h->fslab_clean) {
    PORL_HIP(hipMemsetAsync(W + h->ws.fslab, 0, sizeof(float) * cdiv(h->cfg.max_batch, 16) * h->fslab_stride, s));
    h->fslab_clean = true;
  }
  {
    double macs = 0;
    for (int l = 0; l < a.n_lin; ++l) macs += (double)a.dims[l] * a.dims[l + 1];
    ProfScope ps("qnet_fused_kernel", s, 2.0 * B * macs * 4.0, 8.0 * B * a.dims[0]);
    if (rows16)
      hipLaunchKernelGGL(qnet_fused2_kernel<16>, dim3(nblk), dim3(512), (size_t)h->fused16_lds_bytes, s, a, h->fused16_lds_w2);
    else if (two)
      hipLaunchKernelGGL(qnet_fused2_kernel<32>, dim3(nblk), dim3(512), (size_t)h->fused2_lds_bytes, s, a, h->fused2_lds_w2);
    else
      hipLaunchKernelGGL(qnet_fused_kernel, dim3(nblk), dim3(256), (size_t)h->fused_lds_bytes, s, a);
    PORL_HIP(hipGetLastError());
  }
  QnetAdam ad{};
  if (with_adam) {
    if (hp->step < 1) PORL_FAIL(PORL_ERR_INVALID, "adam step must be >= 1");
    // the scalars of adam_launch: python doubles, rounded to fp32 where they meet tensors
    ad.p = h->buf.params; ad.m = h->buf.adam_m; ad.v = h->buf.adam_v;
    ad.omb1 = (float)(1.0 - hp->adam_beta1); ad.beta2 = (float)hp->adam_beta2; ad.omb2 = (float)(1.0 - hp->adam_beta2);
    ad.eps = (float)hp->adam_eps;
    ad.step_size = (float)(hp->lr / (1.0 - std::pow(hp->adam_beta1, (double)hp->step)));
    ad.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(hp->adam_beta2, (double)hp->step));
  }
  ProfScope ps(with_adam ? "qnet_reduce_kernel+adam" : "qnet_reduce_kernel", s, 0.0, 4.0 * nblk * h->n_params);
  hipLaunchKernelGGL(qnet_reduce_kernel, dim3(cdiv((int)h->n_params, 32)), dim3(256), 0, s, W + h->ws.fslab, (long)h->fslab_stride,
                     nblk, (long)h->n_params, h->buf.grads, W + h->ws.part_td, W + h->ws.part_pen, hp->inv_batch, hp->alpha,
                     h->buf.stats, ad);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

// backward of the multi-launch path from dL/d(output) `dz` (B, ld[L]): dW_l = dZ_l^T In_l (split over the batch),
// dZ_{l-1} = (dZ_l W_l) . 1[In_l > 0], top down; needs the online forward's activations in ws.act[] and the batch in ws.xs
static int qnet_backward_chain(porl_qnet* h, float* dz, int B, hipStream_t s) {
  const int L = h->cfg.n_hidden;
  float* W = h->buf.workspace;
  float* G = h->buf.grads;
  ReduceArgs red{};
  float* slab = W + h->ws.slab;
  for (int l = L; l >= 0; --l) {
    const int out_d = h->net.dims[l + 1], in_d = h->net.dims[l];
    const float* in = l == 0 ? W + h->ws.xs : W + h->ws.act[l - 1];
    const int ldin = l == 0 ? h->Sp : h->ld[l - 1];
    GemmGroup g{};
    g.p[g.nprob] = make_prob(GEMM_TN, dz, h->ld[l], in, ldin, G + h->net.w[l], h->wld[l], out_d, in_d, B);
    g.p[g.nprob].colsum = G + h->net.b[l];
    GemmProb& wg = g.p[g.nprob++];
    float* dz_next = nullptr;
    if (l > 0) {
      dz_next = W + h->ws.dz[(l - 1) & 1];
      GemmProb q = make_prob(GEMM_NN, dz, h->ld[l], h->buf.params + h->net.w[l], h->wld[l], dz_next, h->ld[l - 1], B, in_d, out_d);
      q.mask = in; q.ldmask = ldin;
      g.p[g.nprob++] = q;
    }
    const int tile = TILE_64x64;
    const int sk = pick_splitk(out_d, in_d, B, 1, 64, 64);
    if (sk > 1) {
      const int64_t per = (int64_t)out_d * h->wld[l];          // slabs have the padded row stride of the gradient image
      if (!h->slab_clean) {                                    // their padding columns are never written: zero them once
        PORL_HIP(hipMemsetAsync(W + h->ws.slab, 0, sizeof(float) * (size_t)SK_MAX * (h->n_params + 64), s));
        h->slab_clean = true;
      }
      if (red.njobs + 2 > 8) { PORL_TRY(launch_reduce(red, s)); red = ReduceArgs{}; }
      float* slabW = slab; slab += (int64_t)sk * per;
      float* slabC = slab; slab += (int64_t)sk * out_d;
      wg.splitk = sk; wg.C = slabW; wg.colsum = slabC;
      add_reduce(red, G + h->net.w[l], slabW, per, per, sk);
      add_reduce(red, G + h->net.b[l], slabC, out_d, out_d, sk);
    }
    PORL_TRY(launch_group(g, tile, s));
    if (red.njobs == 8 || l == 0) { PORL_TRY(launch_reduce(red, s)); red = ReduceArgs{}; }
    dz = dz_next;
  }
  return PORL_OK;
}

// Multi-launch gradient of the loaded minibatch (any layer widths): forwards through the grouped GEMM, loss head with
// the optional DQN variants, backward chain.  `var` may be null (plain CQL / DQN).
static int qnet_general_backward(porl_qnet* h, const porl_qnet_hyper* hp, const porl_qnet_variant* var, hipStream_t s);

int porl_qnet_cql_backward(porl_qnet* h, const porl_qnet_hyper* hp, void* stream) {
  PORL_TRY(qnet_ready(h, true)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  hipStream_t s = (hipStream_t)stream;
  const int B = h->batch;
  float* W = h->buf.workspace;
  if (h->fused_ok && g_qnet_fused)
    return qnet_fused_backward(h, hp, B, W + h->ws.xs, h->Sp, W + h->ws.xn, h->Sp,
                               reinterpret_cast<const int64_t*>(W + h->ws.actions), W + h->ws.rew, W + h->ws.done, nullptr, s);
  return qnet_general_backward(h, hp, nullptr, s);
}

static int qnet_general_backward(porl_qnet* h, const porl_qnet_hyper* hp, const porl_qnet_variant* var, hipStream_t s) {
  const int B = h->batch, L = h->cfg.n_hidden, A = h->cfg.n_actions;
  float* W = h->buf.workspace;
  // forward: target net on s' (activations ping-pong in tmp), online net on s (activations kept)
  float* dst[2][PORL_MAX_HIDDEN + 1];
  for (int l = 0; l <= L; ++l) { dst[0][l] = W + h->ws.tmp[l & 1]; dst[1][l] = W + h->ws.act[l]; }
  const float* params[2] = {h->buf.params_tgt, h->buf.params};
  const float* inputs[2] = {W + h->ws.xn, W + h->ws.xs};
  PORL_TRY(qnet_forward(h, 2, params, inputs, dst, B, s));
  float* Qn = dst[0][L];
  float* Q = dst[1][L];
  float* dz = W + h->ws.dz[L & 1];
  const float* Qon = nullptr;
  if (var && var->double_dqn) {
    // Double DQN: the online network on s' as well; its activations ping-pong through the two dZ buffers (free until the
    // loss head writes dL/dQ), so Q_online(s') ends in the buffer dL/dQ goes to — a row is read before it is written
    float* dst2[1][PORL_MAX_HIDDEN + 1];
    for (int l = 0; l <= L; ++l) dst2[0][l] = W + h->ws.dz[l & 1];
    const float* p2[1] = {h->buf.params};
    const float* in2[1] = {W + h->ws.xn};
    PORL_TRY(qnet_forward(h, 1, p2, in2, dst2, B, s));
    Qon = dst2[0][L];
  }
  const int nblk = cdiv(B, 256);
  {
    CqlLossArgs a{};
    a.Qon = Qon;
    if (var) { a.is_w = var->is_weights; a.w_uniform = var->uniform_weight; a.td_abs = var->td_abs; a.next_mask = var->next_mask; a.td_off = var->td_off; }
    a.Q = Q; a.Qn = Qn; a.ldq = h->ld[L];
    a.actions = reinterpret_cast<const int64_t*>(W + h->ws.actions); a.rew = W + h->ws.rew; a.done = W + h->ws.done;
    a.dQ = dz; a.part_td = W + h->ws.part_td; a.part_pen = W + h->ws.part_pen;
    a.B = B; a.A = A; a.gamma = hp->gamma; a.alpha = hp->alpha; a.inv_batch = hp->inv_batch;
    a.log_A = (float)std::log((double)A);
    hipLaunchKernelGGL(cql_loss_kernel, dim3(nblk), dim3(256), 0, s, a);
    PORL_HIP(hipGetLastError());
    hipLaunchKernelGGL(cql_finalize_kernel, dim3(1), dim3(64), 0, s, W + h->ws.part_td, W + h->ws.part_pen, nblk,
                       hp->inv_batch, hp->alpha, h->buf.stats);
    PORL_HIP(hipGetLastError());
  }
  return qnet_backward_chain(h, dz, B, s);
}

// ---- general forward / backward pieces for losses computed outside the engine (QR-DQN, C51: dist_losses.hpp) ----------
// Forward of the online (which_params = 0) or target (1) network on the LOADED batch's states (which_input = 0) or
// next states (1); out (batch, n_outputs) with row stride out_rs.  keep != 0 stores the hidden activations for
// porl_qnet_backward (online network on the states only).
int porl_qnet_forward_loaded(porl_qnet* h, int which_params, int which_input, int keep, float* out, int64_t out_rs,
                             void* stream) {
  PORL_TRY(qnet_ready(h, true)); DevGuard _dg(h->device);
  if (!out || out_rs < h->cfg.n_actions) PORL_FAIL(PORL_ERR_INVALID, "bad output");
  if (keep && (which_params != 0 || which_input != 0)) PORL_FAIL(PORL_ERR_INVALID, "keep: online network on the states only");
  hipStream_t s = (hipStream_t)stream;
  const int L = h->cfg.n_hidden, B = h->batch;
  float* W = h->buf.workspace;
  float* dst[1][PORL_MAX_HIDDEN + 1];
  for (int l = 0; l <= L; ++l) dst[0][l] = keep ? W + h->ws.act[l] : W + h->ws.tmp[l & 1];
  const float* params[1] = {which_params ? h->buf.params_tgt : h->buf.params};
  const float* inputs[1] = {W + (which_input ? h->ws.xn : h->ws.xs)};
  PORL_TRY(qnet_forward(h, 1, params, inputs, dst, B, s));
  PackArgs a{};
  a.rows = B; a.njobs = 1;
  a.job[0].src = dst[0][L]; a.job[0].dst = out; a.job[0].src_row_stride = h->ld[L]; a.job[0].src_col_stride = 1;
  a.job[0].cols = h->cfg.n_actions; a.job[0].ld = (int)out_rs;
  const long n = (long)B * out_rs;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024), 1), dim3(256), 0, s, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

// Backward from dL/d(output) `dout` (batch, n_outputs; row stride dout_rs) through the online network whose forward was
// kept by porl_qnet_forward_loaded: leaves the gradient in grads (complete: slabs combined); porl_qnet_apply follows.
int porl_qnet_backward(porl_qnet* h, const float* dout, int64_t dout_rs, void* stream) {
  PORL_TRY(qnet_ready(h, true)); DevGuard _dg(h->device);
  if (!dout || dout_rs < h->cfg.n_actions) PORL_FAIL(PORL_ERR_INVALID, "bad gradient input");
  hipStream_t s = (hipStream_t)stream;
  const int L = h->cfg.n_hidden, B = h->batch;
  float* W = h->buf.workspace;
  float* dz = W + h->ws.dz[L & 1];
  PackArgs a{};
  a.rows = B; a.njobs = 1;
  a.job[0].src = dout; a.job[0].dst = dz; a.job[0].src_row_stride = dout_rs; a.job[0].src_col_stride = 1;
  a.job[0].cols = h->cfg.n_actions; a.job[0].ld = h->ld[L];
  const long n = (long)B * h->ld[L];
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024), 1), dim3(256), 0, s, a);
  PORL_HIP(hipGetLastError());
  return qnet_backward_chain(h, dz, B, s);
}

static int dist_args_ok(int B, int A, int N, int64_t ld) {
  if (B < 1 || A < 1 || N < 1 || N > DIST_MAX_N || ld < (int64_t)A * N) PORL_FAIL(PORL_ERR_INVALID, "bad distributional-loss shapes (N <= %d)", DIST_MAX_N);
  return 0;
}

int porl_qr_loss(const float* z_cur, const float* z_next_online, const float* z_next_target, int64_t ld, const int64_t* actions,
                 const float* rewards, const float* dones, int32_t batch, int32_t n_actions, int32_t n_quantiles, float gamma,
                 float kappa, float* dz_out, float* row_loss, void* stream) {
  if (!z_cur || !z_next_online || !z_next_target || !actions || !rewards || !dones || !dz_out || !row_loss)
    PORL_FAIL(PORL_ERR_INVALID, "null argument");
  PORL_TRY(dist_args_ok(batch, n_actions, n_quantiles, ld));
  DevGuard _dg(device_of(dz_out));
  QrLossArgs a{z_cur, z_next_online, z_next_target, (long)ld, actions, rewards, dones, dz_out, row_loss, batch, n_actions,
               n_quantiles, gamma, kappa, 1.0f / batch};
  hipLaunchKernelGGL(qr_loss_kernel, dim3(cdiv(batch, 4)), dim3(256), 0, (hipStream_t)stream, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iqn_quantile_huber(const float* current, const float* target, const float* taus, int32_t batch, int32_t n_current,
                            int32_t n_target, float kappa, float* dcurrent_out, float* row_loss, void* stream) {
  if (!current || !target || !taus || !dcurrent_out || !row_loss || batch < 1 || n_current < 1 || n_target < 1)
    PORL_FAIL(PORL_ERR_INVALID, "bad arguments");
  DevGuard _dg(device_of(dcurrent_out));
  IqnLossArgs a{current, target, taus, dcurrent_out, row_loss, batch, n_current, n_target, kappa, 1.0f / batch};
  hipLaunchKernelGGL(iqn_loss_kernel, dim3(cdiv(batch, 4)), dim3(256), 0, (hipStream_t)stream, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

namespace {
inline unsigned iqn_blocks(long items) { return (unsigned)std::min<long>(std::max<long>((items + 255) / 256, 1), 256L * 16); }
inline int iqn_dims_ok(int32_t batch, int32_t n_tau, int32_t third) {
  if (batch < 1 || n_tau < 1 || third < 1) PORL_FAIL(PORL_ERR_INVALID, "batch %d, n_tau %d, width/actions %d must be >= 1", batch, n_tau, third);
  if ((int64_t)batch * n_tau * third > (int64_t)1 << 40) PORL_FAIL(PORL_ERR_INVALID, "tensor too large");
  return PORL_OK;
}
}  // namespace

int porl_iqn_cos_embed(const float* taus, int64_t n, int32_t embedding_dim, float* out, void* stream) {
  if (!taus || !out) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  if (n < 1 || embedding_dim < 1 || n > ((int64_t)1 << 40) / embedding_dim) PORL_FAIL(PORL_ERR_INVALID, "n %lld, embedding_dim %d", (long long)n, embedding_dim);
  DevGuard _dg(device_of(out));
  hipLaunchKernelGGL(iqn_cos_embed_kernel, dim3(iqn_blocks(n * embedding_dim)), dim3(256), 0, (hipStream_t)stream, taus, (long)n,
                     embedding_dim, out);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iqn_hadamard(const float* feat, int64_t ldf, const float* emb, int32_t batch, int32_t n_tau, int32_t width,
                      float* out, void* stream) {
  if (!feat || !emb || !out) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  PORL_TRY(iqn_dims_ok(batch, n_tau, width));
  if (ldf < width) PORL_FAIL(PORL_ERR_INVALID, "feature row stride %lld < width %d", (long long)ldf, width);
  const bool al = !((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(emb) | reinterpret_cast<uintptr_t>(out)) & 15u);
  DevGuard _dg(device_of(out));
  hipLaunchKernelGGL(iqn_hadamard_kernel, dim3(iqn_blocks((long)batch * n_tau * width / 4)), dim3(256), 0, (hipStream_t)stream,
                     feat, (long)ldf, emb, batch, n_tau, width, out, al ? 1 : 0);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iqn_hadamard_backward(const float* dout, const float* feat, int64_t ldf, const float* emb, int32_t batch,
                               int32_t n_tau, int32_t width, float* dfeat, float* demb, void* stream) {
  if (!dout || !feat || !emb || (!dfeat && !demb)) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  PORL_TRY(iqn_dims_ok(batch, n_tau, width));
  if (ldf < width) PORL_FAIL(PORL_ERR_INVALID, "feature row stride %lld < width %d", (long long)ldf, width);
  DevGuard _dg(device_of(dout));
  hipLaunchKernelGGL(iqn_hadamard_bwd_kernel, dim3(iqn_blocks((long)batch * width)), dim3(256), 0, (hipStream_t)stream, dout,
                     feat, (long)ldf, emb, batch, n_tau, width, dfeat, demb);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iqn_select(const float* z, const int64_t* actions, int32_t batch, int32_t n_tau, int32_t n_actions, float* out,
                    void* stream) {
  if (!z || !actions || !out) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  PORL_TRY(iqn_dims_ok(batch, n_tau, n_actions));
  DevGuard _dg(device_of(out));
  hipLaunchKernelGGL(iqn_select_kernel, dim3(iqn_blocks((long)batch * n_tau)), dim3(256), 0, (hipStream_t)stream, z, actions,
                     batch, n_tau, n_actions, out);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iqn_scatter(const float* dsel, const int64_t* actions, int32_t batch, int32_t n_tau, int32_t n_actions, float* dz,
                     void* stream) {
  if (!dsel || !actions || !dz) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  PORL_TRY(iqn_dims_ok(batch, n_tau, n_actions));
  DevGuard _dg(device_of(dz));
  hipLaunchKernelGGL(iqn_scatter_kernel, dim3(iqn_blocks((long)batch * n_tau * n_actions)), dim3(256), 0, (hipStream_t)stream,
                     dsel, actions, batch, n_tau, n_actions, dz);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_iqn_target(const float* z_online_next, const float* z_target_next, const float* rewards, const float* dones,
                    float gamma, int32_t batch, int32_t n_tau, int32_t n_actions, float* td, int64_t* next_actions,
                    void* stream) {
  if (!z_online_next || !z_target_next || !rewards || !dones || !td) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  PORL_TRY(iqn_dims_ok(batch, n_tau, n_actions));
  DevGuard _dg(device_of(td));
  hipLaunchKernelGGL(iqn_target_kernel, dim3(iqn_blocks(batch)), dim3(256), 0, (hipStream_t)stream, z_online_next,
                     z_target_next, rewards, dones, gamma, batch, n_tau, n_actions, td, next_actions);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_grad_clip(float* grads, int64_t n, float max_norm, float* norm_coef, double* workspace, void* stream) {
  if (!grads || !norm_coef || !workspace) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  if (n < 0 || n > (int64_t)1 << 40) PORL_FAIL(PORL_ERR_INVALID, "n = %lld", (long long)n);
  if (!(max_norm > 0.f)) PORL_FAIL(PORL_ERR_INVALID, "max_norm must be positive");
  DevGuard _dg(device_of(grads));
  hipStream_t s = (hipStream_t)stream;
  const int nb = (int)std::min<long>(CLIP_BLOCKS, std::max<long>(1, (n + 4095) / 4096));
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, grads, (long)n, workspace);
  PORL_HIP(hipGetLastError());
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, s, workspace, nb, max_norm, norm_coef);
  PORL_HIP(hipGetLastError());
  if (n > 0) {
    hipLaunchKernelGGL(scale_by_kernel, dim3(iqn_blocks(n)), dim3(256), 0, s, grads, (long)n, norm_coef);
    PORL_HIP(hipGetLastError());
  }
  return PORL_OK;
}

int porl_c51_loss(const float* logits_cur, const float* logits_next_target, int64_t ld, const int64_t* actions,
                  const float* rewards, const float* dones, const float* support, int32_t batch, int32_t n_actions,
                  int32_t n_atoms, float gamma, float v_min, float v_max, float* dlogits_out, float* row_loss, void* stream) {
  if (!logits_cur || !logits_next_target || !actions || !rewards || !dones || !support || !dlogits_out || !row_loss)
    PORL_FAIL(PORL_ERR_INVALID, "null argument");
  PORL_TRY(dist_args_ok(batch, n_actions, n_atoms, ld));
  if (n_atoms < 2 || !(v_max > v_min)) PORL_FAIL(PORL_ERR_INVALID, "need n_atoms >= 2 and v_max > v_min");
  DevGuard _dg(device_of(dlogits_out));
  // delta_z as the reference forms it (python double (v_max - v_min) / (atom_size - 1), meeting fp32 tensors as fp32)
  const float delta = (float)(((double)v_max - (double)v_min) / (double)(n_atoms - 1));
  C51LossArgs a{logits_cur, logits_next_target, (long)ld, actions, rewards, dones, support, dlogits_out, row_loss, batch, n_actions,
                n_atoms, gamma, v_min, v_max, delta, 1.0f / batch};
  hipLaunchKernelGGL(c51_loss_kernel, dim3(cdiv(batch, 4)), dim3(256), 0, (hipStream_t)stream, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

int porl_qnet_apply(porl_qnet* h, const porl_qnet_hyper* hp, void* stream) {
  PORL_TRY(qnet_ready(h, false)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  return adam_launch(h->buf.params, h->buf.grads, h->buf.adam_m, h->buf.adam_v, nullptr, h->n_params, hp->lr, hp->step,
                     hp->adam_beta1, hp->adam_beta2, hp->adam_eps, 0.0, (hipStream_t)stream);
}

int porl_qnet_learn(porl_qnet* h, const porl_qnet_hyper* hp, void* stream) {
  PORL_TRY(qnet_ready(h, true)); DevGuard _dg(h->device);
  if (!hp) PORL_FAIL(PORL_ERR_INVALID, "null hyper-parameters");
  if (h->fused_ok && g_qnet_fused) {
    float* W = h->buf.workspace;
    return qnet_fused_backward(h, hp, h->batch, W + h->ws.xs, h->Sp, W + h->ws.xn, h->Sp,
                               reinterpret_cast<const int64_t*>(W + h->ws.actions), W + h->ws.rew, W + h->ws.done, nullptr,
                               (hipStream_t)stream, true);
  }
  PORL_TRY(porl_qnet_cql_backward(h, hp, stream));
  return porl_qnet_apply(h, hp, stream);
}

// learn() on rows idx[b] of the replay arrays for networks the one-launch kernel does not cover: gather into the staging
// buffers (one launch), multi-launch gradient with the variant's loss head, Adam.  Same arithmetic per element as the
// one-launch kernel's loss stage; sums run in the grouped GEMM's order instead of per 32-row block.
static int qnet_general_learn(porl_qnet* h, const porl_qnet_hyper* hp, const porl_qnet_variant* var, int batch,
                              const float* states, int64_t s_rs, const float* next_states, int64_t n_rs,
                              const int64_t* actions, const float* rewards, const float* dones, const int64_t* idx,
                              hipStream_t s) {
  float* W = h->buf.workspace;
  QnetGatherArgs a{};
  a.states = states; a.next_states = next_states; a.s_rs = (long)s_rs; a.n_rs = (long)n_rs;
  a.actions = actions; a.rew = rewards; a.done = dones; a.idx = idx;
  a.xs = W + h->ws.xs; a.xn = W + h->ws.xn; a.act_out = reinterpret_cast<int64_t*>(W + h->ws.actions);
  a.rew_out = W + h->ws.rew; a.done_out = W + h->ws.done;
  a.B = batch; a.S = h->cfg.state_dim; a.ld = h->Sp;
  hipLaunchKernelGGL(qnet_gather_kernel, dim3((unsigned)(((long)batch * h->Sp + 255) / 256)), dim3(256), 0, s, a);
  PORL_HIP(hipGetLastError());
  h->batch = batch;
  PORL_TRY(qnet_general_backward(h, hp, var, s));
  return porl_qnet_apply(h, hp, (void*)s);
}

int porl_qnet_learn_indexed(porl_qnet* h, const float* states, int64_t s_rs, const int64_t* actions, const float* rewards,
                            const float* next_states, int64_t n_rs, const float* dones, const int64_t* idx, int32_t batch,
                            const porl_qnet_hyper* hp, void* stream) {
  PORL_TRY(qnet_ready(h, false)); DevGuard _dg(h->device);
  if (!hp || !states || !actions || !rewards || !next_states || !dones) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  if (batch < 1 || batch > h->cfg.max_batch) PORL_FAIL(PORL_ERR_INVALID, "batch %d outside [1,%d]", batch, h->cfg.max_batch);
  if (!h->fused_ok || !g_qnet_fused)
    return qnet_general_learn(h, hp, nullptr, batch, states, s_rs, next_states, n_rs, actions, rewards, dones, idx, (hipStream_t)stream);
  return qnet_fused_backward(h, hp, batch, states, s_rs, next_states, n_rs, actions, rewards, dones, idx,
                             (hipStream_t)stream, true);
}

int32_t porl_qnet_can_sample(const porl_qnet* h) {
  return h && h->fused_ok && g_qnet_fused && g_qnet_two_groups && h->fused2_lds_w2 > 0 ? 1 : 0;
}

int porl_qnet_learn_sampled(porl_qnet* h, const float* states, int64_t s_rs, const int64_t* actions, const float* rewards,
                            const float* next_states, int64_t n_rs, const float* dones, int64_t n_rows, uint64_t seed,
                            uint64_t draw, int32_t batch, const porl_qnet_hyper* hp, void* stream) {
  PORL_TRY(qnet_ready(h, false)); DevGuard _dg(h->device);
  if (!hp || !states || !actions || !rewards || !next_states || !dones) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  if (batch < 1 || batch > h->cfg.max_batch) PORL_FAIL(PORL_ERR_INVALID, "batch %d outside [1,%d]", batch, h->cfg.max_batch);
  if (n_rows < batch || n_rows > (int64_t(1) << 40)) PORL_FAIL(PORL_ERR_INVALID, "need batch <= n_rows <= 2^40");
  if (!porl_qnet_can_sample(h)) PORL_FAIL(PORL_ERR_UNSUPPORTED, "in-kernel sampling needs the two-group one-launch step kernel");
  QnetSampling sp;
  sp.n_rows = n_rows; sp.seed = seed; sp.step = draw;
  return qnet_fused_backward(h, hp, batch, states, s_rs, next_states, n_rs, actions, rewards, dones, nullptr,
                             (hipStream_t)stream, true, nullptr, &sp);
}

int porl_qnet_learn_variant(porl_qnet* h, const float* states, int64_t s_rs, const int64_t* actions, const float* rewards,
                            const float* next_states, int64_t n_rs, const float* dones, const int64_t* idx, int32_t batch,
                            const porl_qnet_hyper* hp, const porl_qnet_variant* variant, void* stream) {
  PORL_TRY(qnet_ready(h, false)); DevGuard _dg(h->device);
  if (!hp || !states || !actions || !rewards || !next_states || !dones || !variant) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  if (batch < 1 || batch > h->cfg.max_batch) PORL_FAIL(PORL_ERR_INVALID, "batch %d outside [1,%d]", batch, h->cfg.max_batch);
  if (!h->fused_ok || !g_qnet_fused)      // wide networks (a layer > 128 wide, > 5 Linear layers): the multi-launch path
    return qnet_general_learn(h, hp, variant, batch, states, s_rs, next_states, n_rs, actions, rewards, dones, idx, (hipStream_t)stream);
  return qnet_fused_backward(h, hp, batch, states, s_rs, next_states, n_rs, actions, rewards, dones, idx,
                             (hipStream_t)stream, true, variant);
}

int porl_qnet_sync_target(porl_qnet* h, void* stream) {
  PORL_TRY(qnet_ready(h, false)); DevGuard _dg(h->device);
  PORL_HIP(hipMemcpyAsync(h->buf.params_tgt, h->buf.params, sizeof(float) * h->n_params, hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
  return PORL_OK;
}

// Q(s, .) for a loaded batch of states (which = 0 online, 1 target) -> q_out (batch, n_actions), row stride q_rs
int porl_qnet_forward(porl_qnet* h, int which, const float* states, int64_t s_rs, int32_t batch, float* q_out,
                      int64_t q_rs, void* stream) {
  PORL_TRY(qnet_ready(h, false)); DevGuard _dg(h->device);
  if (!q_out) PORL_FAIL(PORL_ERR_INVALID, "null output");
  hipStream_t s = (hipStream_t)stream;
  PORL_TRY(porl_qnet_load_batch(h, batch, states, s_rs, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, stream));
  h->batch = 0;
  const int L = h->cfg.n_hidden;
  float* W = h->buf.workspace;
  float* dst[1][PORL_MAX_HIDDEN + 1];
  for (int l = 0; l <= L; ++l) dst[0][l] = W + h->ws.tmp[l & 1];
  const float* params[1] = {which ? h->buf.params_tgt : h->buf.params};
  const float* inputs[1] = {W + h->ws.xs};
  PORL_TRY(qnet_forward(h, 1, params, inputs, dst, batch, s));
  PackArgs a{};
  a.rows = batch; a.njobs = 1;
  a.job[0].src = dst[0][L]; a.job[0].dst = q_out; a.job[0].src_row_stride = h->ld[L]; a.job[0].src_col_stride = 1;
  a.job[0].cols = h->cfg.n_actions; a.job[0].ld = (int)q_rs;
  if (q_rs < h->cfg.n_actions) PORL_FAIL(PORL_ERR_INVALID, "q_rs smaller than n_actions");
  const long n = (long)batch * q_rs;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024), 1), dim3(256), 0, s, a);
  PORL_HIP(hipGetLastError());
  return PORL_OK;
}

// mean_b( logsumexp_a Q(s_b, a) - ln A - Q(s_b, a_b) ) -> out[0]   (compute_cql_penalty)
int porl_qnet_penalty(porl_qnet* h, const float* states, int64_t s_rs, const int64_t* actions, int64_t a_rs,
                      int32_t batch, float* out, void* stream) {
  PORL_TRY(qnet_ready(h, false)); DevGuard _dg(h->device);
  if (!out || !actions) PORL_FAIL(PORL_ERR_INVALID, "null argument");
  hipStream_t s = (hipStream_t)stream;
  PORL_TRY(porl_qnet_load_batch(h, batch, states, s_rs, actions, a_rs, nullptr, 0, nullptr, 0, nullptr, 0, stream));
  h->batch = 0;
  const int L = h->cfg.n_hidden;
  float* W = h->buf.workspace;
  float* dst[1][PORL_MAX_HIDDEN + 1];
  for (int l = 0; l <= L; ++l) dst[0][l] = W + h->ws.tmp[l & 1];
  const float* params[1] = {h->buf.params};
  const float* inputs[1] = {W + h->ws.xs};
  PORL_TRY(qnet_forward(h, 1, params, inputs, dst, batch, s));
  const int nblk = cdiv(batch, 256);
  hipLaunchKernelGGL(cql_penalty_kernel, dim3(nblk), dim3(256), 0, s, dst[0][L], h->ld[L],
                     reinterpret_cast<const int64_t*>(W + h->ws.actions), batch, h->cfg.n_actions,
                     (float)std::log((double)h->cfg.n_actions), W + h->ws.part_pen);
  PORL_HIP(hipGetLastError());
  ReduceArgs r{};
  // sum of the per-block partials, scaled by 1/B: reuse the finalize kernel (td part = 0)
  hipLaunchKernelGGL(cql_finalize_kernel, dim3(1), dim3(64), 0, s, W + h->ws.part_pen, W + h->ws.part_pen, nblk,
                     1.0f / batch, 0.0f, W + h->ws.part_td);
  PORL_HIP(hipGetLastError());
  PORL_HIP(hipMemcpyAsync(out, W + h->ws.part_td + 2, sizeof(float), hipMemcpyDeviceToDevice, s));
  (void)r;
  return PORL_OK;
}

}  // extern "C"

#include "encoder_api.inc"
