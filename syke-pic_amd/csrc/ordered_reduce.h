// Ordered (bitwise reproducible) per-channel reductions of the training step's partial rows - the sums the conv /
// depthwise / BatchNorm-backward kernels leave per block as [row][2][C] floats - followed by a per-channel finalize
// functor.  Shared by train_kernels.hip (ResNets) and train_effnet.hip (channel-padded EfficientNets).  Stands in for the
// reductions inside torch's BatchNorm2d forward / backward (sykepic/train/train.py:240-242).
#pragma once
#include "spk_common.h"

namespace spk_reduce {

// Ordered parallel sum of `count` partial rows for 64 channels at a time:
// block = 64 channels x 16 row groups; thread (c, r) adds rows r, r+16, ...
// (coalesced 256-B rows), then the 16 group sums are added in index order.
// Deterministic for a given `count`.  which: 0 / 1 selects [row][which][C].
__device__ __forceinline__ double colsum64(const float* __restrict__ partials, int count, int C,
                                           int which, int c, int r, bool valid, double* sm) {
  double acc = 0.0;
  if (valid)
    for (int t = r; t < count; t += 16) acc += (double)partials[((size_t)t * 2 + which) * C + c];
  sm[r * 64 + (threadIdx.x & 63)] = acc;
  __syncthreads();
  double tot = 0.0;
  if (r == 0)
    for (int k = 0; k < 16; ++k) tot += sm[k * 64 + (threadIdx.x & 63)];
  __syncthreads();
  return tot;
}

// few partial rows: one stage, one block per 64 channels
template <class Fin>
__global__ __launch_bounds__(1024) void finalize_kernel(const float* __restrict__ partials, int count, int C, Fin fin) {
  __shared__ double sm[16 * 64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const bool valid = c < C;
  const double s1 = colsum64(partials, count, C, 0, c, r, valid, sm);
  const double s2 = colsum64(partials, count, C, 1, c, r, valid, sm);
  if (r == 0 && valid) fin(c, s1, s2);
}

// Stage 1 when there are many partial rows (one per M tile of the conv kernels: 6272 on ResNet-50's first stage at batch
// 256): blockIdx.y takes a contiguous slice of the `count` rows and writes one row of out[slices][2][C]; finalize_kernel
// over those rows follows.  (Both stages in ONE launch - the block that arrives last at a per-channel-group counter
// finalizes, slice rows handed over by write-through stores and agent-scope loads - was built and measured in round 5:
// 12.5-13.8 us against 6 + 6 for the two launches, bit-identical, no gain: the time of these kernels is their dependent
// load chain, not the launch.  With an agent-scope release in front of the counter instead: 15-20 us, the release writes
// back the conv output sitting dirty in the XCD's L2.)
template <int HEADER_ONLY = 0>   // (a template so that both translation units may carry it)
__global__ __launch_bounds__(1024) void partial_rows_kernel(const float* __restrict__ partials, int count, int C,
                                                            int rows_per_slice, float* __restrict__ out) {
  __shared__ double sm[16 * 64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const bool valid = c < C;
  const int t0 = blockIdx.y * rows_per_slice;
  const int cnt = max(0, min(count - t0, rows_per_slice));
  const float* base = partials + (size_t)t0 * 2 * C;
  const double s1 = colsum64(base, cnt, C, 0, c, r, valid, sm);
  const double s2 = colsum64(base, cnt, C, 1, c, r, valid, sm);
  if (r == 0 && valid) {
    out[((size_t)blockIdx.y * 2 + 0) * C + c] = (float)s1;
    out[((size_t)blockIdx.y * 2 + 1) * C + c] = (float)s2;
  }
}

}  // namespace spk_reduce

namespace spk_reduce {

// the ordered per-channel reduction of `count` partial rows followed by `fin`: one stage for few rows, else two (64 slices)
template <class Fin>
inline int reduce_finalize(const float* partials, int count, int C, float* tmp, const Fin& fin, hipStream_t s) {
  if (count > 128 && tmp) {
    const int slices = 64;
    const int rps = (count + slices - 1) / slices;
    hipLaunchKernelGGL((spk_reduce::partial_rows_kernel<0>), dim3((C + 63) / 64, slices), dim3(1024), 0, s, partials, count, C,
                       rps, tmp);
    partials = tmp;
    count = slices;
  }
  hipLaunchKernelGGL((spk_reduce::finalize_kernel<Fin>), dim3((C + 63) / 64), dim3(1024), 0, s, partials, count, C, fin);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace spk_reduce
