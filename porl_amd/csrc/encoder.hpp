// Kernels of the costmap encoder forward (reference agent/fasternet.py:428-438, config 5): everything around
// the 1x1-convolution GEMMs.  Activations are NHWC fp32 ("rows" = (sample, y, x) positions, channels
// contiguous), so every 1x1 convolution is a plain (rows, Cin) x (Cout, Cin)^T product for gemm_f32.hpp
// and all kernels here are HBM-bound row sweeps with 16-byte accesses.  64-bit row indices throughout:
// batch 512 is 2.9 M positions of up to 384 channels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace porl {

// ---------------------------------------------------------------------------------------------------
// PatchEmbed.proj (fasternet.py:238, Conv2d(3, E, 4, stride 4, bias=False)) on the NCHW costmap image.
// One block per (patch row, sample): the 3*4 image rows of the strip and the transposed weights sit in
// LDS; lane = patch column, wave = channel quarter; k runs (c, ky, kx) ascending.
//   img (B, 3, Hi, Wi), w (E, 48) -> out (B, Hi/4, Wi/4, E)
// ---------------------------------------------------------------------------------------------------
constexpr int PE_K = 48;
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                          float* __restrict__ out, int Hi, int Wi, int E) {
  extern __shared__ float pe_lds[];
  float* strip = pe_lds;                 // [12][Wi]
  float* wl = pe_lds + 12 * Wi;          // [48][E]
  const int py = blockIdx.x, b = blockIdx.y, Hp = Hi >> 2, Wp = Wi >> 2;
  for (int i = threadIdx.x; i < 12 * (Wi >> 2); i += 256) {
    const int row = i / (Wi >> 2), x4 = i - row * (Wi >> 2);
    const int c = row >> 2, ky = row & 3;
    const float4 v = *reinterpret_cast<const float4*>(img + (((long)b * 3 + c) * Hi + 4 * py + ky) * Wi + 4 * x4);
    *reinterpret_cast<float4*>(strip + row * Wi + 4 * x4) = v;
  }
  for (int i = threadIdx.x; i < PE_K * E; i += 256) {
    const int e = i / PE_K, k = i - e * PE_K;
    wl[k * E + e] = w[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, cq = threadIdx.x >> 6, eq = E >> 2;
  for (int px = lane; px < Wp; px += 64) {
    float* orow = out + (((long)b * Hp + py) * Wp + px) * E;
    for (int c0 = cq * eq; c0 < (cq + 1) * eq; c0 += 8) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int row = 0; row < 12; ++row) {
        const float4 a = *reinterpret_cast<const float4*>(strip + row * Wi + 4 * px);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const float* wk = wl + (row * 4 + kx) * E + c0;
          const float4 w0 = *reinterpret_cast<const float4*>(wk), w1 = *reinterpret_cast<const float4*>(wk + 4);
          acc[0] = fmaf(av[kx], w0.x, acc[0]); acc[1] = fmaf(av[kx], w0.y, acc[1]);
          acc[2] = fmaf(av[kx], w0.z, acc[2]); acc[3] = fmaf(av[kx], w0.w, acc[3]);
          acc[4] = fmaf(av[kx], w1.x, acc[4]); acc[5] = fmaf(av[kx], w1.y, acc[5]);
          acc[6] = fmaf(av[kx], w1.z, acc[6]); acc[7] = fmaf(av[kx], w1.w, acc[7]);
        }
      }
      *reinterpret_cast<float4*>(orow + c0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(orow + c0 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// PatchEmbed straight from the lidar state: the costmap image (util/costmap.py:7-64) has at most one beam pixel per
// row in channel 0 plus a 5-pixel goal cross in all three channels, so 94 % of the 4x4 patches are empty and
// their convolution is exactly 0.  Instead of rasterising 1.1 MB per sample and convolving it densely, a block
// marks the set pixels of one patch row as 48-bit masks (bit k = (c, ky, kx), the weight's own order), and a
// patch's output is the sum of the weight columns of its set bits in ascending k — bit-identical to the dense
// fmaf chain over {0, 1} pixels.
//   patch_stats_kernel : BatchNorm batch statistics from the non-empty patches only (fp64, one partial per block)
//   patch_bn_kernel    : x1 = fma(conv, alpha, beta) for every position (empty patches get beta), NHWC
// Pixel semantics are costmap_kernel's (kernels.hpp): values > 8 read as 0, beams rolled by n_ang/2, distance bin
// 0 cleared, python-style wrap of the cross at bin -1.
// ---------------------------------------------------------------------------------------------------
struct CostmapGeom { int n_ang, n_dist; float dist_inc, ang_inc, deg_min, deg_max, dist_max; };

// marks the set pixels of patch row py of sample `st` in mask[Wp] (LDS, zeroed by the caller); threads 0..3 take
// the four beams of the strip, thread 4 the goal cross
__device__ __forceinline__ void patch_row_masks(const float* __restrict__ st, const CostmapGeom& g, int py,
                                                unsigned long long* mask, int t) {
  auto rd = [&](int i) { const float v = st[i]; return v > 8.f ? 0.f : v; };
  auto wrap = [](long i, int n) { return i < 0 ? i + n : i; };
  auto set = [&](int c, int ky, long col) {
    atomicOr(&mask[col >> 2], 1ull << (c * 16 + ky * 4 + (int)(col & 3)));
  };
  if (t < 4) {
    const int r = 4 * py + t;
    const int src = (r - g.n_ang / 2 + g.n_ang) % g.n_ang;
    const long bin = (long)(rd(src) / g.dist_inc);
    if (bin > 0 && bin < g.n_dist) set(0, t, bin);
  } else if (t == 4) {
    const float gx = rd(g.n_ang), gy = rd(g.n_ang + 1);
    float deg = atan2f(gy, gx);
    deg = fminf(fmaxf(deg, g.deg_min), g.deg_max);
    const long deg_bin = (long)((deg + 3.14159265358979323846f) / g.ang_inc);
    const float cd = fminf(sqrtf(gx * gx + gy * gy), g.dist_max);
    const long dist_bin = (long)(cd / g.dist_inc);
    for (int ky = 0; ky < 4; ++ky) {
      const long r = 4 * py + ky;
      const bool is_deg = r == wrap(deg_bin, g.n_ang);
      const bool near = is_deg || r == wrap(deg_bin - 1, g.n_ang) || r == wrap(deg_bin + 1, g.n_ang);
      for (int c = 0; c < 3; ++c) {
        if (near) set(c, ky, wrap(dist_bin, g.n_dist));
        if (is_deg) { set(c, ky, wrap(dist_bin - 1, g.n_dist)); set(c, ky, wrap(dist_bin + 1, g.n_dist)); }
      }
    }
  }
}

// sum of the weight columns of the set bits, ascending k; wl = [48][E] in LDS
__device__ __forceinline__ float patch_value(unsigned long long m, const float* wl, int E, int e) {
  float acc = 0.f;
  while (m) {
    const int k = __builtin_ctzll(m);
    acc = acc + wl[k * E + e];
    m &= m - 1;
  }
  return acc;
}

constexpr int PE_MAX_E = 128;

// One block per sample (grid-stride): all Hp*Wp masks of the sample at once, the non-empty patches compacted into
// a list, then thread (e, half) adds the values of channel e over its half of the list in list order (fp64).
__global__ __launch_bounds__(256) void patch_stats_kernel(const float* __restrict__ state, long state_rs, int batch,
                                                          CostmapGeom g, const float* __restrict__ w, int E,
                                                          double* __restrict__ partial) {
  extern __shared__ float ps_lds[];
  const int Wp = g.n_dist >> 2, Hp = g.n_ang >> 2, P = Hp * Wp;
  float* wl = ps_lds;                                                                     // [48][E]
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(ps_lds + PE_K * E);   // [P]
  int* list = reinterpret_cast<int*>(mask + P);                                          // [P] (worst case)
  __shared__ int n_list;
  __shared__ int counts[256];
  __shared__ double red[2][2 * PE_MAX_E];
  const int t = threadIdx.x;
  for (int i = t; i < PE_K * E; i += 256) {
    const int e = i / PE_K, k = i - e * PE_K;
    wl[k * E + e] = w[i];
  }
  const int e = t % E, half = t / E;            // threads >= 2E idle in the accumulation (E >= 128: one half only)
  const int halves = 256 / E >= 2 ? 2 : 1;
  double s = 0, q = 0;
  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const float* st = state + (long)b * state_rs;
    __syncthreads();                                             // list/masks of the previous sample are consumed
    for (int i = t; i < P; i += 256) mask[i] = 0ull;
    __syncthreads();
    for (int py = t >> 3; py < Hp; py += 32) patch_row_masks(st, g, py, mask + py * Wp, t & 7);
    __syncthreads();
    // ordered compaction (thread t owns a contiguous chunk of patches): the list, and with it the order of the
    // fp64 sums, is the same on every run
    {
      const int chunk = (P + 255) / 256, i0 = t * chunk, i1 = min(P, i0 + chunk);
      int cnt = 0;
      for (int i = i0; i < i1; ++i) cnt += mask[i] != 0ull;
      counts[t] = cnt;
      __syncthreads();
      int off = 0;
      for (int j = 0; j < t; ++j) off += counts[j];
      for (int i = i0; i < i1; ++i)
        if (mask[i]) list[off++] = i;
      if (t == 255) n_list = off;
    }
    __syncthreads();
    if (half < halves) {
      const int n = n_list;
      for (int i = half; i < n; i += halves) {
        const float v = patch_value(mask[list[i]], wl, E, e);
        s += v;
        q += (double)v * v;
      }
    }
  }
  if (half < halves) { red[half][e] = s; red[half][E + e] = q; }
  __syncthreads();
  if (t < 2 * E) {
    double r = red[0][t];
    if (halves == 2) r += red[1][t];
    partial[(long)blockIdx.x * 2 * E + t] = r;
  }
}

// thread i of a block = (patch column i / (E/4), channel group i % (E/4)): consecutive lanes write consecutive
// 16-byte pieces of the output rows
// OUT_BF16: the normalised patch embedding leaves as bf16 (the encoder's bf16-activation mode, encoder_bf16.hpp); `out`
// then points at a (rows, E) bf16 tensor.  The arithmetic up to the store is the fp32 path's.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void patch_bn_kernel(const float* __restrict__ state, long state_rs, CostmapGeom g,
                                                       const float* __restrict__ w, int E, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, float* __restrict__ out, int rows_per_block) {
  extern __shared__ float ps_lds[];
  float* wl = ps_lds;
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(ps_lds + PE_K * E);
  const int t = threadIdx.x, e4n = E >> 2, Wp = g.n_dist >> 2, Hp = g.n_ang >> 2;
  const long b = blockIdx.y;
  for (int i = t; i < PE_K * E; i += 256) {
    const int e = i / PE_K, k = i - e * PE_K;
    wl[k * E + e] = w[i];
  }
  // a block walks `rows_per_block` patch rows with the transposed weight parked once (small images: at 84x84 a patch
  // row is 21 x 96 outputs, less work than parking the 18 KB weight)
  for (int py = blockIdx.x * rows_per_block; py < (int)(blockIdx.x + 1) * rows_per_block && py < Hp; ++py) {
  __syncthreads();                        // the previous row's masks are no longer read (first trip: the weight is parked)
  for (int i = t; i < Wp; i += 256) mask[i] = 0ull;
  __syncthreads();
  patch_row_masks(state + b * state_rs, g, py, mask, t);
  __syncthreads();
  float* orow = out + (b * Hp + py) * (long)Wp * (OUT_BF16 ? E / 2 : E);      // bf16 rows are half as long
  for (int i = t; i < Wp * e4n; i += 256) {
    const int px = i / e4n, c = (i - px * e4n) << 2;
    const unsigned long long m = mask[px];
    const float4 a4 = *reinterpret_cast<const float4*>(alpha + c);
    float4 o = *reinterpret_cast<const float4*>(beta + c);      // fma(0, alpha, beta)
    if (m) {
      o.x = fmaf(patch_value(m, wl, E, c), a4.x, o.x);
      o.y = fmaf(patch_value(m, wl, E, c + 1), a4.y, o.y);
      o.z = fmaf(patch_value(m, wl, E, c + 2), a4.z, o.z);
      o.w = fmaf(patch_value(m, wl, E, c + 3), a4.w, o.w);
    }
    if constexpr (OUT_BF16) {
      typedef __bf16 pb_bf16x4 __attribute__((ext_vector_type(4)));
      pb_bf16x4 ob;
      ob.x = (__bf16)o.x; ob.y = (__bf16)o.y; ob.z = (__bf16)o.z; ob.w = (__bf16)o.w;
      reinterpret_cast<pb_bf16x4*>(orow)[i] = ob;
    } else {
      reinterpret_cast<float4*>(orow)[i] = o;
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm2d batch statistics (fasternet.py:164,240,255 in train mode): per-channel sum and sum of squares
// of an (M, C) row-major activation, accumulated in fp64 like the CPU reference's accumulate type.
// Stage 1: each block owns a contiguous row range and writes one partial per channel; stage 2 (one
// thread per channel) adds the partials in block order, forms alpha = invstd*gamma, beta = bias - mean*alpha
// and updates the running statistics (momentum, unbiased variance).
// ---------------------------------------------------------------------------------------------------
constexpr int CS_MAX_BLOCKS = 2048;
__global__ __launch_bounds__(256) void colstats_partial_kernel(const float* __restrict__ x, long M, int C,
                                                               double* __restrict__ partial) {
  extern __shared__ double cs_lds[];      // [rows_par][2][C]
  const int C4 = C >> 2, rows_par = 256 / C4;
  const int c4 = threadIdx.x % C4, r = threadIdx.x / C4;
  const long per = (M + gridDim.x - 1) / gridDim.x;
  const long m0 = (long)blockIdx.x * per, m1 = m0 + per < M ? m0 + per : M;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (r < rows_par) {
    for (long m = m0 + r; m < m1; m += rows_par) {
      const float4 v = *reinterpret_cast<const float4*>(x + m * C + 4 * c4);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cs_lds[(r * 2 + 0) * C + 4 * c4 + j] = s[j];
      cs_lds[(r * 2 + 1) * C + 4 * c4 + j] = q[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    double t = 0;
    for (int rr = 0; rr < rows_par; ++rr) t += cs_lds[rr * 2 * C + i];
    partial[(long)blockIdx.x * 2 * C + i] = t;
  }
}

// Same stage-1 output, but from the per-32-row column sums a GEMM epilogue left behind (GemmProb::cstat,
// ((M+31)/32, 2, C) floats) instead of a second pass over the activation: 1/16 of the traffic.
__global__ __launch_bounds__(256) void colstats_from_blocks_kernel(const float* __restrict__ cstat, long nrb, int C,
                                                                   double* __restrict__ partial) {
  const long per = (nrb + gridDim.x - 1) / gridDim.x;
  const long b0 = (long)blockIdx.x * per, b1 = b0 + per < nrb ? b0 + per : nrb;
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    // four entries in flight per thread (a single chain of dependent loads ran this sweep at 0.9 TB/s); the four partial
    // sums are added in a fixed order
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    long rb = b0;
    for (; rb + 3 < b1; rb += 4) {
      const float a0 = cstat[rb * 2 * C + i], a1 = cstat[(rb + 1) * 2 * C + i];
      const float a2 = cstat[(rb + 2) * 2 * C + i], a3 = cstat[(rb + 3) * 2 * C + i];
      t0 += (double)a0; t1 += (double)a1; t2 += (double)a2; t3 += (double)a3;
    }
    for (; rb < b1; ++rb) t0 += (double)cstat[rb * 2 * C + i];
    partial[(long)blockIdx.x * 2 * C + i] = (t0 + t1) + (t2 + t3);
  }
}

// one wave per channel: lanes add the block partials in a fixed strided order, then a shuffle tree
__global__ __launch_bounds__(64) void bn_finish_kernel(const double* __restrict__ partial, int nblocks, int C, long M,
                                                       const float* __restrict__ gamma, const float* __restrict__ bias,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       int training, double eps, double momentum,
                                                       float* __restrict__ alpha, float* __restrict__ beta) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float mean_f, invstd_f;
  if (training) {
    double s = 0, q = 0;
    for (int b = lane; b < nblocks; b += 64) {
      s += partial[(long)b * 2 * C + c];
      q += partial[(long)b * 2 * C + C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      q += __shfl_xor(q, o);
    }
    const double mean = s / (double)M;
    double var = q / (double)M - mean * mean;
    if (var < 0) var = 0;
    mean_f = (float)mean;
    invstd_f = (float)(1.0 / sqrt(var + eps));
    if (lane == 0) {
      const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
      run_mean[c] = (float)(momentum * mean + (1.0 - momentum) * (double)run_mean[c]);
      run_var[c] = (float)(momentum * unbiased + (1.0 - momentum) * (double)run_var[c]);
    }
  } else {
    mean_f = run_mean[c];
    invstd_f = (float)(1.0 / sqrt((double)run_var[c] + eps));
  }
  if (lane == 0) {
    const float a = __fmul_rn(invstd_f, gamma[c]);
    alpha[c] = a;
    beta[c] = __fsub_rn(bias[c], __fmul_rn(mean_f, a));
  }
}

// y = x*alpha[c] + beta[c] (+ReLU), in place or not; n4 = M*C/4 float4 items
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, int C4,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       int relu) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 a = reinterpret_cast<const float4*>(alpha)[c4], b = reinterpret_cast<const float4*>(beta)[c4];
    float4 o = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------------
// Partial_conv3 (fasternet.py:110-138): 3x3, pad 1, on the first Cp channels.  This kernel lays the
// (rows, 9*Cp) patch matrix out for the GEMM, k = (ky, kx, ci), zero outside the image, and copies the
// untouched channels Cp..C-1 of x into y (the `cat` of forward_split_cat, :132-136).
//   x (B, H, W, C) -> col (B*H*W, 9*Cp), y[:, Cp:] = x[:, Cp:]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ col,
                                                        float* __restrict__ y, long rows, int H, int W, int C, int Cp) {
  const int cp4 = Cp >> 2, rest4 = (C - Cp) >> 2, per_row = 9 * cp4 + rest4;
  const long total = rows * per_row;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / per_row;
    const int j = (int)(i - m * per_row);
    if (j < 9 * cp4) {
      const int t = j / cp4, ci4 = j - t * cp4;
      const int ky = t / 3, kx = t - ky * 3;
      const int xx = (int)(m % W), yy = (int)((m / W) % H);
      const int sy = yy + ky - 1, sx = xx + kx - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sy >= 0 && sy < H && sx >= 0 && sx < W)
        v = *reinterpret_cast<const float4*>(x + (m + (long)(ky - 1) * W + (kx - 1)) * C + 4 * ci4);
      *reinterpret_cast<float4*>(col + m * (9L * Cp) + (long)t * Cp + 4 * ci4) = v;
    } else {
      const int c = Cp + 4 * (j - 9 * cp4);
      *reinterpret_cast<float4*>(y + m * C + c) = *reinterpret_cast<const float4*>(x + m * C + c);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Partial_conv3 as a direct convolution (the patch matrix above costs more HBM traffic than the conv costs
// FLOPs).  One block = 64 lanes x (OCB/8) waves on a tile of TR rows x W columns of one sample:
// lane = (row, group of 4 consecutive x), wave = group of 8 output channels.  The input tile with its halo
// sits in LDS channel-planar ([row][ci][x], x shifted by 4 so the 4 centre values are one aligned 16-byte
// read).  The weights of a wave are uniform, so they arrive through the scalar cache (s_load_dwordx8 from the
// [ky][ci][kx][oc] copy made by pack_pconv_kernel) and feed v_pk_fma_f32 as SGPR pairs: no LDS traffic and no
// VGPRs for them, 48 packed FMAs per 3 LDS reads.  blockIdx.z selects the OCB-wide slice of output channels.
// Also copies the untouched channels Cp..C-1 (slice 0 only).
//   x (B, H, W, C) -> y[:, :Cp] = conv3x3(x[:, :Cp]), y[:, Cp:] = x[:, Cp:]
// ---------------------------------------------------------------------------------------------------
template <int CP, int OCB, int W>
__global__ __launch_bounds__(64 * (OCB / 8)) void pconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                  float* __restrict__ y, int H, int C) {
  constexpr int XG = W / 4, TR = 64 / XG, RS = W + 8, NT = 64 * (OCB / 8);
  static_assert(W % 4 == 0 && 64 % XG == 0 && OCB % 8 == 0 && CP % OCB == 0 && CP % 4 == 0, "tile shape");
  extern __shared__ float pc_lds[];
  float* hl = pc_lds;                              // [TR + 2][CP][RS]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r0 = blockIdx.x * TR, oc0 = blockIdx.z * OCB;
  const long b = blockIdx.y;
  const float* xs = x + b * H * W * C;
  float* ys = y + b * H * W * C;
  // input rows r0-1 .. r0+TR, columns -1 .. W, CP channels; zeros outside the image
  constexpr int CP4 = CP / 4;
  for (int i = t; i < (TR + 2) * (W + 2) * CP4; i += NT) {
    const int c4 = i % CP4, xx = (i / CP4) % (W + 2), rr = i / (CP4 * (W + 2));
    const int gy = r0 + rr - 1, gx = xx - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const float4*>(xs + ((long)gy * W + gx) * C + 4 * c4);
    float* d = hl + (rr * CP + 4 * c4) * RS + gx + 4;
    d[0] = v.x; d[RS] = v.y; d[2 * RS] = v.z; d[3 * RS] = v.w;
  }
  __syncthreads();
  const int row = lane / XG, xg = lane % XG;
  float acc[4][8];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[p][o] = 0.f;
  const float* wbase = wp + oc0 + wave * 8;
#pragma unroll 1
  for (int ky = 0; ky < 3; ++ky) {
    const float* hrow = hl + (row + ky) * CP * RS + 4 * xg + 4;
#pragma unroll 2
    for (int ci = 0; ci < CP; ++ci) {
      const float4 mid = *reinterpret_cast<const float4*>(hrow + ci * RS);
      const float in[6] = {hrow[ci * RS - 1], mid.x, mid.y, mid.z, mid.w, hrow[ci * RS + 4]};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* wk = wbase + ((ky * CP + ci) * 3 + kx) * CP;
        float wv[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) wv[o] = wk[o];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int o = 0; o < 8; ++o) acc[p][o] = fmaf(in[p + kx], wv[o], acc[p][o]);
      }
    }
  }
  const int gy = r0 + row;
  if (gy < H) {
    float* o = ys + ((long)gy * W + 4 * xg) * C + oc0 + wave * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<float4*>(o + (long)p * C) = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
      *reinterpret_cast<float4*>(o + (long)p * C + 4) = make_float4(acc[p][4], acc[p][5], acc[p][6], acc[p][7]);
    }
  }
  if (blockIdx.z == 0) {
    const int rest4 = (C - CP) / 4, rows_here = min(TR, H - r0);
    const long base = (long)r0 * W * C;
    for (int i = t; i < rows_here * W * rest4; i += NT) {
      const int c4 = i % rest4, pos = i / rest4;
      const long off = base + (long)pos * C + CP + 4 * c4;
      *reinterpret_cast<float4*>(ys + off) = *reinterpret_cast<const float4*>(xs + off);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Partial_conv3 on the fp32 MATRIX pipe (round 3): implicit GEMM, out (POS = 128 consecutive positions of one sample) x
// (32 output channels per pass), K = 9 taps x CP input channels, v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains; the
// summation order is (tap, ci) instead of the direct kernel's (ky, ci, kx): results agree to rounding).  The direct
// kernel above runs 184 GFLOP per update on the vector ALU at 48 TFLOP/s (3.7 ms); the same work is ~25 % of one
// matrix-pipe millisecond.  The touched input rows + halo (zeros outside the image) sit in LDS as [row][x][ci]
// (ci-contiguous: a lane reads 4 consecutive ci with one 16-byte load, lane half kh takes ci = 8 g + 4 kh + j), the
// weight of the current 32-channel pass as [tap][oc][ci] from the zero-padded image pack_pconv_mfma_kernel made.  Any
// geometry.  Also copies the untouched channels CP..C-1 of its positions.
//   x (B, H, W, C) -> y[:, :CP] = conv3x3(x[:, :CP]), y[:, CP:] = x[:, CP:]
// ---------------------------------------------------------------------------------------------------
struct PconvMfmaArgs {
  const float* x; float* y; const float* w;     // w: [pass][tap][32][CP + 4]
  int C, Hh, Ww, tiles_per_sample, tiles_total, tiles_per_block;
};

__global__ void pack_pconv_mfma_kernel(const float* __restrict__ src, float* __restrict__ dst, int CP, int npass) {
  const int XS = CP + 4, n = npass * 9 * 32 * XS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int ci = i % XS, oc = (i / XS) % 32, tap = (i / (XS * 32)) % 9, ps = i / (XS * 32 * 9);
    const int o = ps * 32 + oc;
    dst[i] = (o < CP && ci < CP) ? src[((long)o * CP + ci) * 9 + tap] : 0.f;
  }
}

template <int CP>
__global__ __launch_bounds__(256) void pconv_f32_mfma_kernel(const PconvMfmaArgs a) {
  constexpr int XS = CP + 4, POS = 128, NPASS = (CP + 31) / 32, U = CP / 4;
  static_assert(CP % 8 == 0, "the k loop walks groups of 8 input channels");
  extern __shared__ __attribute__((aligned(16))) float pm_lds[];
  float* Wl = pm_lds;                               // [9][32][XS]
  float* X = Wl + 9 * 32 * XS;                      // [nr][Ww + 2][XS]
  typedef float pm_f32x16 __attribute__((ext_vector_type(16)));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, kh = lane >> 5;
  const int P = a.Hh * a.Ww, WW2 = a.Ww + 2;
  for (int k = 0; k < a.tiles_per_block; ++k) {
    const int gt = blockIdx.x * a.tiles_per_block + k;
    if (gt >= a.tiles_total) break;                                // block-uniform
    const int b = gt / a.tiles_per_sample, tile = gt - b * a.tiles_per_sample;
    const int p0 = tile * POS, p1 = min(P, p0 + POS) - 1;
    const int y_lo = p0 / a.Ww - 1, nr = p1 / a.Ww + 1 - y_lo + 1;
    const float* xs = a.x + (long)b * P * a.C;
    float* ys = a.y + (long)b * P * a.C;
    if (k) __syncthreads();                                        // the previous tile's fragments have been read
    for (int i = t; i < nr * WW2 * U; i += 256) {
      const int pix = i / U, u = i - pix * U, ry = pix / WW2, rx = pix - ry * WW2;
      const int yy = y_lo + ry, xx = rx - 1;
      const bool ok = yy >= 0 && yy < a.Hh && xx >= 0 && xx < a.Ww;
      const float4 v = *reinterpret_cast<const float4*>(xs + (ok ? ((long)yy * a.Ww + xx) * a.C + u * 4 : 0));
      *reinterpret_cast<float4*>(X + pix * XS + u * 4) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int p = p0 + 32 * wave + li;
    const int pc = p < P ? p : p0;                                 // lanes past the sample compute a valid pixel, unused
    const int yy = pc / a.Ww, xx = pc - yy * a.Ww;
    const float* xa = X + ((yy - y_lo) * WW2 + xx + 1) * XS + kh * 4;
#pragma unroll 1
    for (int ps = 0; ps < NPASS; ++ps) {
      if (ps || k == 0 || NPASS > 1) {
        if (ps) __syncthreads();                                   // every wave is done with the previous pass's weights
        for (int i = t; i < 9 * 32 * XS / 4; i += 256)
          reinterpret_cast<float4*>(Wl)[i] = reinterpret_cast<const float4*>(a.w + (long)ps * 9 * 32 * XS)[i];
      }
      __syncthreads();
      pm_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wb = Wl + li * XS + kh * 4;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int off = ((tap / 3 - 1) * WW2 + (tap % 3 - 1)) * XS;
#pragma unroll
        for (int g = 0; g < CP / 8; ++g) {
          const float4 fa = *reinterpret_cast<const float4*>(xa + off + g * 8);
          const float4 fb = *reinterpret_cast<const float4*>(wb + tap * 32 * XS + g * 8);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
        }
      }
      const int oc = ps * 32 + li;
      if (oc < CP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pos = p0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (pos < P) ys[(long)pos * a.C + oc] = acc[r];
        }
      }
    }
    // the untouched channels of this tile's positions
    const int rest4 = (a.C - CP) / 4, npos = p1 - p0 + 1;
    for (int i = t; i < npos * rest4; i += 256) {
      const int pos = p0 + i / rest4, c = CP + 4 * (i % rest4);
      *reinterpret_cast<float4*>(ys + (long)pos * a.C + c) = *reinterpret_cast<const float4*>(xs + (long)pos * a.C + c);
    }
  }
}

// conv weight (Cp, Cp, 3, 3) -> [ky][ci][kx][oc], the order pconv3x3_kernel walks
__global__ void pack_pconv_kernel(const float* __restrict__ src, float* __restrict__ dst, int CP) {
  const int n = CP * CP * 9;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int oc = i % CP, kx = (i / CP) % 3, ci = (i / (3 * CP)) % CP, ky = i / (3 * CP * CP);
    dst[i] = src[((long)oc * CP + ci) * 9 + ky * 3 + kx];
  }
}


// PatchMerging.reduction input (fasternet.py:253, Conv2d(C, 2C, 2, stride 2)): (B, H, W, C) -> (B*H/2*W/2, 4C)
// with k = (ky, kx, c)
__global__ __launch_bounds__(256) void space_to_depth_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             long out_rows, int H, int W, int C) {
  const int c4n = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long total = out_rows * 4 * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / (4 * c4n);
    const int j = (int)(i - m * 4 * c4n);
    const int t = j / c4n, c4 = j - t * c4n, ky = t >> 1, kx = t & 1;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho);
    const long b = m / ((long)Wo * Ho);
    const long src = ((b * H + 2 * oy + ky) * W + 2 * ox + kx) * C + 4 * c4;
    reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<const float4*>(x + src);
  }
}

// AdaptiveAvgPool2d(1) (fasternet.py:368): mean over the P positions of each sample; 64 channels per block as 16
// float4 columns x 16 row lanes (16-byte loads, two rows in flight per lane), fp64 partial sums combined in lane order.
// x (B, P, C) -> out (B, C); C a multiple of 4
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ x, float* __restrict__ out, int P, int C) {
  __shared__ double part[16][64];
  const int q = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * q;
  const long b = blockIdx.y;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (c < C) {
    const float* base = x + b * (long)P * C + c;
    int p = r;
    for (; p + 16 < P; p += 32) {
      const float4 u = *reinterpret_cast<const float4*>(base + (long)p * C);
      const float4 v = *reinterpret_cast<const float4*>(base + (long)(p + 16) * C);
      s0 += u.x; s1 += u.y; s2 += u.z; s3 += u.w;
      s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    if (p < P) {
      const float4 u = *reinterpret_cast<const float4*>(base + (long)p * C);
      s0 += u.x; s1 += u.y; s2 += u.z; s3 += u.w;
    }
  }
  part[r][4 * q] = s0; part[r][4 * q + 1] = s1; part[r][4 * q + 2] = s2; part[r][4 * q + 3] = s3;
  __syncthreads();
  const int cc = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x < 64 && cc < C) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
    out[b * C + cc] = (float)(t / (double)P);
  }
}

// conv weight (O, I, KH, KW) -> (O, KH, KW, I): the k order the patch matrices above use
__global__ void permute_oihw_ohwi_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int I, int KK) {
  const int n = O * I * KK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int o = i / (I * KK), rem = i - o * I * KK, t = rem / I, ci = rem - t * I;
    dst[i] = src[((long)o * I + ci) * KK + t];
  }
}

}  // namespace porl
