// Prioritized replay on the device (reference src/porl/buffer/sum_tree.py:4-77 and
// prioritized_replay_buffer.py:36-108): the sum tree lives in HBM as fp64 in the reference's own heap layout
// (root 0, children 2i+1 / 2i+2, leaf of data slot d at d + capacity - 1), so tree indices mean the same thing.
//   * priority write-back: the batch's leaves are set (a leaf written twice keeps the LAST value, like the
//     reference's sequential loop), then every ancestor is recomputed from its two children level by level
//     inside one block (deterministic; the reference adds rounded differences up the tree instead, which drifts
//     by ulps — sums here are the exact pairwise sums)
//   * stratified sampling: segment i draws s = a + (b - a) * u_i exactly like random.uniform(a, b) with the host's
//     u_i = random.random(), then walks down (`s <= left ? left : right, s -= left`), one lane per sample
//   * importance weights (n_entries * p / total)^-beta normalised by their maximum, same block
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace porl {

// stamp[leaf data slot] = 1 + the largest batch position that writes it
__global__ void per_stamp_kernel(const int64_t* __restrict__ tree_idx, int n, int64_t capacity, int* __restrict__ stamp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMax(&stamp[tree_idx[i] - (capacity - 1)], i + 1);
}

// winners write their priority and clear the stamp; priority = (|td| + eps)^alpha in fp64 (prioritized_replay_buffer.py:21)
__global__ void per_set_leaves_kernel(double* __restrict__ tree, const int64_t* __restrict__ tree_idx,
                                      const double* __restrict__ td_error, int n, int64_t capacity, double eps,
                                      double alpha, int* __restrict__ stamp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t slot = tree_idx[i] - (capacity - 1);
  if (stamp[slot] == i + 1) {
    tree[tree_idx[i]] = pow(fabs(td_error[i]) + eps, alpha);
    stamp[slot] = 0;
  }
}

// one block: every touched leaf walks to the root, recomputing each ancestor from its children; a barrier per level
__global__ __launch_bounds__(1024) void per_propagate_kernel(double* __restrict__ tree, const int64_t* __restrict__ tree_idx,
                                                             int n, int levels) {
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    int64_t node = i < n ? tree_idx[i] : 0;
    for (int l = 0; l < levels; ++l) {
      if (node != 0) {
        node = (node - 1) / 2;
        tree[node] = tree[2 * node + 1] + tree[2 * node + 2];
      }
      __syncthreads();
    }
  }
}

struct PerSampleArgs {
  const double* tree; int64_t capacity;
  const double* u;             // (batch,) uniforms in [0, 1) from the host generator
  int batch; int64_t n_entries; double beta;
  int64_t* out_idx;            // tree indices
  double* out_prio;            // sampled priorities
  float* out_w;                // importance weights, normalised by their maximum
};

__global__ __launch_bounds__(256) void per_sample_kernel(const PerSampleArgs a) {
  __shared__ double red[256];
  const int64_t size = 2 * a.capacity - 1;
  const double total = a.tree[0];
  const double segment = total / (double)a.batch;
  double wmax = 0.0;
  for (int i = threadIdx.x; i < a.batch; i += 256) {
    const double lo = segment * (double)i, hi = segment * (double)(i + 1);
    double s = lo + (hi - lo) * a.u[i];                       // random.uniform(a, b) = a + (b - a) * random()
    int64_t idx = 0;
    for (;;) {
      const int64_t left = 2 * idx + 1;
      if (left >= size) break;
      const double lv = a.tree[left];
      if (s <= lv) idx = left;
      else { s -= lv; idx = left + 1; }
    }
    const double p = a.tree[idx];
    a.out_idx[i] = idx;
    a.out_prio[i] = p;
    const double w = pow((double)a.n_entries * (p / total), -a.beta);
    a.out_prio[a.batch + i] = w;                                // raw weight, normalised below
    wmax = fmax(wmax, w);
  }
  red[threadIdx.x] = wmax;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  wmax = red[0];
  for (int i = threadIdx.x; i < a.batch; i += 256) a.out_w[i] = (float)(a.out_prio[a.batch + i] / wmax);
}

}  // namespace porl
