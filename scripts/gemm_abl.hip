// Timing-only ablation harness for the GEMM main loop (results are wrong for PORL_ABL != 0).
#include "../porl_amd/csrc/gemm_f32.hpp"
using namespace porl;
extern "C" int abl_gemm(int mode, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        void* stream) {
  GemmGroup g{};
  g.nprob = 1;
  g.p[0] = make_prob(mode, A, lda, B, ldb, C, ldc, M, N, K);
  plan_group(g, ABL_TILE);
  const TileCfg c = tile_cfg(ABL_TILE);
  dim3 grid(g.total_blocks), block(64 * c.wm * c.wn);
#if ABL_TILE == 0
  hipLaunchKernelGGL((gemm_f32_kernel<128, 128, GEMM_BK, 2, 2, true, false>), grid, block, 0, (hipStream_t)stream, g);
#else
  hipLaunchKernelGGL((gemm_f32_kernel<64, 64, GEMM_BK, 2, 2, true, false>), grid, block, 0, (hipStream_t)stream, g);
#endif
  return (int)hipGetLastError();
}

#ifdef PORL_STAMP
extern "C" int abl_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif
