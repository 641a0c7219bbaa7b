// Micro-benchmark: what does one memory instruction cost when issued between f32 MFMAs (one wave / SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { NONE, RD128, RD64, RD32, WR128, WR32, GLD };

// KIND every N MFMAs; results of reads are parked in a ring and folded in 2 rounds later (never a stall)
template <int KIND, int N>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, unsigned long long* stamps) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int t = threadIdx.x;
  for (int i = t; i < 16384; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float a0 = in[t], b0 = in[t + 256];
  float4 ring[4];
  for (int q = 0; q < 4; ++q) ring[q] = make_float4(a0, b0, a0, b0);
  float sink = 0.f;
  const float* gp = in + (size_t)blockIdx.x * 8192 + t * 4;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[u & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u % N == 0) {
        const int slot = (u / N) & 3;
        if (u / N >= 2 || true) sink += ring[(slot + 2) & 3].x;          // consume what was requested 2 rounds ago
        const int off = (t * 4 + (u / N) * 1024 + it * 64) & 16380;
        if (KIND == RD128) ring[slot] = *reinterpret_cast<const float4*>(lds + off);
        if (KIND == RD64) { const float2 q = *reinterpret_cast<const float2*>(lds + ((t * 2 + u * 512) & 16382)); ring[slot].x = q.x; ring[slot].y = q.y; }
        if (KIND == RD32) ring[slot].x = lds[(t + u * 256 + it) & 16383];
        if (KIND == WR128) *reinterpret_cast<float4*>(lds + off) = ring[slot];
        if (KIND == WR32) lds[(t + u * 256) & 16383] = ring[slot].x;
        if (KIND == GLD) ring[slot] = *reinterpret_cast<const float4*>(gp + ((it * 2048 + (u / N) * 1024 * 64) & 0x3FFFFF));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = sink;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int q = 0; q < 4; ++q) s += ring[q].y;
  out[blockIdx.x * 256 + t] = s;
  if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND, int N>
void run(const char* name, float* out, float* in, unsigned long long* st) {
  const int iters = 1000, blocks = 256;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(256), 0, 0, out, in, iters, st);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, tick = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; tick += h[2 * b + 1]; }
  const double nm = 32.0 * iters * blocks;
  const double per = cyc / nm;
  printf("%-34s every %2d MFMA: %6.1f cyc/MFMA  (+%5.1f cyc per memory instruction)  clock %.2f GHz\n", name, N, per,
         (per - 64.0) * N, cyc / tick * 0.1);
}

int main() {
  float *out, *in; unsigned long long* st;
  hipMalloc(&out, 1 << 22); hipMalloc(&in, 64 << 20); hipMalloc(&st, 1 << 16);
  std::vector<float> h(16 << 20, 1e-3f);
  hipMemcpy(in, h.data(), 64 << 20, hipMemcpyHostToDevice);
  run<NONE, 1>("no filler", out, in, st);
#define ALLN(K, NAME) run<K, 1>(NAME, out, in, st); run<K, 2>(NAME, out, in, st); run<K, 4>(NAME, out, in, st); run<K, 8>(NAME, out, in, st);
  ALLN(RD128, "ds_read_b128") ALLN(RD64, "ds_read_b64") ALLN(RD32, "ds_read_b32")
  ALLN(WR128, "ds_write_b128") ALLN(WR32, "ds_write_b32") ALLN(GLD, "global_load_dwordx4")
  return 0;
}
