// Issue-rate probe for the f32-input MFMA forms (cycles per instruction as one wave sees them), 1 or 2 waves per SIMD,
// one or two accumulator chains, operands from registers or re-read from LDS each step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int SHAPE, int CHAINS, bool LDS>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[64 * 68];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 64 * 68; i += blockDim.x) sm[i] = 0.001f * (i % 97);
  __syncthreads();
  float a = 1.0f + lane * 0.01f, b = 0.5f;
  f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  f16v d0, d1;
  for (int r = 0; r < 16; ++r) { d0[r] = 0; d1[r] = 0; }
  const float* ap = sm + (lane & 15) * 68 + 4 * (lane >> 4);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float4 av = make_float4(a, a, a, a), bv = make_float4(b, b, b, b);
    if (LDS) { av = *reinterpret_cast<const float4*>(ap + 16 * (it & 3)); bv = *reinterpret_cast<const float4*>(ap + 16 * ((it + 1) & 3)); }
    if (SHAPE == 16) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, c0, 0, 0, 0);
      if (CHAINS == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, c0, 0, 0, 0);
      if (CHAINS == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, c0, 0, 0, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, d0, 0, 0, 0);
      if (CHAINS == 2) d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, d1, 0, 0, 0); else d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, d0, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, d0, 0, 0, 0);
      if (CHAINS == 2) d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, d1, 0, 0, 0); else d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, d0, 0, 0, 0);
    }
  }
  float s = c0[0] + c1[0] + d0[0] + d1[0];
  asm volatile("s_nop 0" :: "v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + t / 64] = t1 - t0;
  out[blockIdx.x * blockDim.x + t] = s;
}
template <int SHAPE, int CHAINS, bool LDS>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 1024 * 512); hipMalloc(&cyc, 8 * 8192);
  const int iters = 256;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<SHAPE, CHAINS, LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s %d waves/block: %6.1f cycles per MFMA (wave 0), %6.1f (wave %d)\n", name, threads / 64, (double)h[0] / (4.0 * iters),
         (double)h[threads / 64 - 1] / (4.0 * iters), threads / 64 - 1);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int threads : {256, 512}) {
    if (threads == 256) {
      run<16, 2, false>("16x16x4, 2 chains, registers", 256); run<16, 1, false>("16x16x4, 1 chain, registers", 256);
      run<16, 2, true>("16x16x4, 2 chains, LDS operands", 256);
      run<32, 2, false>("32x32x2, 2 chains, registers", 256); run<32, 1, false>("32x32x2, 1 chain, registers", 256);
      run<32, 2, true>("32x32x2, 2 chains, LDS operands", 256);
    } else {
      run<16, 2, false>("16x16x4, 2 chains, registers", 512); run<16, 1, false>("16x16x4, 1 chain, registers", 512);
      run<16, 2, true>("16x16x4, 2 chains, LDS operands", 512);
      run<32, 2, false>("32x32x2, 2 chains, registers", 512); run<32, 1, false>("32x32x2, 1 chain, registers", 512);
      run<32, 2, true>("32x32x2, 2 chains, LDS operands", 512);
    }
  }
  return 0;
}
