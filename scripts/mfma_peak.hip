// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 under different companions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, unsigned long long* stamps) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int t = threadIdx.x;
  for (int i = t; i < 16384; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float a0 = in[t], b0 = in[t + 256];
  float4 v = make_float4(a0, b0, a0, b0), w = v;
  const float* gp = in + (size_t)blockIdx.x * 4096 + t * 4;
  float4 gl = make_float4(0, 0, 0, 0), wn = v;
  float4 glv[8];
  for (int q = 0; q < 8; ++q) glv[q] = v;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x + (u & 1 ? w.y : w.x), v.y, acc[u % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 1 && (u & 3) == 0) { w = *reinterpret_cast<const float4*>(lds + ((t * 4 + u * 64 + it * 16) & 16380)); }
      if (MODE == 2 && (u & 3) == 0) { *reinterpret_cast<float4*>(lds + ((t * 4 + u * 1024) & 16380)) = v; }
      if (MODE == 3 && u == 0) { gl = *reinterpret_cast<const float4*>(gp + ((it * 1024) & 0xFFFFF)); }
      if (MODE == 3 && u == 15) { v.x += gl.x * 1e-30f; }
      // kernel-like densities: one K-tile = 64 MFMAs (4 iterations of this 16-MFMA body)
      if (MODE == 4 && (it & 3) == 0 && u < 8) { *reinterpret_cast<float4*>(lds + ((t * 4 + u * 1024) & 16380)) = v; }
      if (MODE == 5 && (it & 3) == 1 && u < 8) { glv[u] = *reinterpret_cast<const float4*>(gp + (((it >> 2) * 32 + u * 128 * 1024) & 0x1FFFFF)); }
      if (MODE == 5 && (it & 3) == 0 && u < 8) { *reinterpret_cast<float4*>(lds + ((t * 4 + u * 1024) & 16380)) = glv[u]; }
      if (MODE == 6 && (u & 3) == 3) { wn = *reinterpret_cast<const float4*>(lds + ((t * 4 + u * 64 + it * 16) & 16380)); }
      if (MODE == 6 && (u & 3) == 2) { w = wn; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int q = 0; q < 8; ++q) s += glv[q].x;
  out[blockIdx.x * 256 + t] = s + w.x + gl.y + wn.z;
  if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC, int MODE>
void run(const char* name, int blocks, float* out, float* in, unsigned long long* st) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, in, iters, st);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, in, iters, st);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, tick = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; tick += h[2 * b + 1]; }
  cyc /= blocks; tick /= blocks;
  const double nm = 16.0 * iters;
  const double flops = (double)blocks * 4 * nm * 4096;
  printf("%-44s blocks=%4d: %6.1f cyc/MFMA/wave  clock %.3f GHz  kernel %.1f us  %.1f TF\n", name, blocks, cyc / nm, cyc / tick * 0.1, ms * 1e3, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *out, *in; unsigned long long* st;
  hipMalloc(&out, 1 << 22); hipMalloc(&in, 64 << 20); hipMalloc(&st, 1 << 16);
  std::vector<float> h(16 << 20, 1e-3f);
  hipMemcpy(in, h.data(), 64 << 20, hipMemcpyHostToDevice);
  run<4, 0>("4 acc, MFMA only", 256, out, in, st);
  run<2, 0>("2 acc, MFMA only", 256, out, in, st);
  run<1, 0>("1 acc, MFMA only", 256, out, in, st);
  run<4, 0>("4 acc, MFMA only, 2 blocks/CU", 512, out, in, st);
  run<4, 1>("4 acc + ds_read_b128 per 4 MFMA", 256, out, in, st);
  run<4, 2>("4 acc + ds_write_b128 per 4 MFMA", 256, out, in, st);
  run<4, 3>("4 acc + global_load_dwordx4 per 16 MFMA", 256, out, in, st);
  run<4, 4>("8 ds_write_b128 on 8 consecutive MFMAs per 64", 256, out, in, st);
  run<4, 5>("8 loads + 8 ds_write per 64 MFMA (GEMM-like)", 256, out, in, st);
  run<4, 6>("ds_read_b128 per 4 MFMA, prefetched 3 ahead", 256, out, in, st);
  return 0;
}
