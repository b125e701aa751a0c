// Zero-sum rounding of convolution weights to fp16, and the per-channel activation means it is weighted with.
//
// The eval path stores weights in fp16.  Round-to-nearest loses dw_k = fp16(w_k) - w_k per weight, and an output loses
// sum_k dw_k x_k.  Over the data that error has a MEAN, sum_k dw_k E[x_k] - the inputs of a conv are post-ReLU (or raw
// pixel) values whose means are far from zero - and this systematic part, the same for every pixel of an output
// channel, is what survives the network's averaging (global pool, head): it is most of the logit error of a plain fp16
// forward.  The hi + lo split weights remove ALL of dw at twice the MFMA work.  Zero-sum rounding
// removes the mean at NO run-time cost: each row is rounded to nearest, then the weights that sit closest to a rounding
// midpoint are rounded the other way - one at a time, cheapest first - until sum_k mu_k dw_k has gone to (within the
// smallest available step of) zero, mu_k = E[x_k] from a calibration batch.  A flipped weight was ~0.5 ulp off either
// way, so the row's squared rounding error hardly moves (measured +0.02..1 %), while its mu-weighted sum drops by 2-4
// orders of magnitude.  Measured on the synthetic ResNet-50s (tests/archive/diagnostics/zero_sum_round.py): plain fp16 1.64e-3
// worst |dp|, zero-sum with calibrated means 3.9e-4 - below the 37-conv hi + lo default (8.8e-4) with no second product.
// Groups: a k x k conv is balanced per filter TAP (each cin slice on its own), so a border pixel, which sees a subset
// of the taps, keeps the cancellation.
//
// Stands behind the same call sites as every other eval kernel: `net(x)` in sykepic/compute/probability.py:189.
#include "spk_common.h"

namespace {

// neighbour of the fp16 value `h` in the direction of `up`
__device__ __forceinline__ float f16_neighbour(unsigned short h, bool up, bool* ok) {
  const bool neg = (h & 0x8000u) != 0;
  const unsigned short mag = h & 0x7fffu;
  unsigned short r;
  if (mag == 0) r = up ? 0x0001u : 0x8001u;                 // +-0 -> smallest subnormal of the wanted sign
  else if (neg == up) r = (unsigned short)(h - 1);           // towards zero
  else r = (unsigned short)(h + 1);                          // away from zero
  *ok = (r & 0x7fffu) < 0x7c00u;                             // never step onto inf
  return (float)__builtin_bit_cast(_Float16, r);
}

// One wave per (row, group).  w: [rows][row_len] fp32, mu: [mu_period] (index k % mu_period) or null (all ones).
// out: same shape, every value exactly representable in fp16 and within one fp16 ulp of w.
__global__ __launch_bounds__(64) void zero_sum_round_kernel(const float* __restrict__ w, const float* __restrict__ mu,
                                                            float* __restrict__ out, int row_len, int mu_period,
                                                            int max_iter) {
#pragma clang fp contract(off)   // no FMA contraction: oracle/zero_sum.py restates this arithmetic operation by operation
  extern __shared__ float sm[];          // step[row_len], cost[row_len]
  float* step = sm;
  float* cost = sm + row_len;
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * row_len;
  // nearest fp16 value q of v and its other neighbour (the fp16 value on the far side of v); ok = false: none
  auto neighbours = [](float v, float* q, float* alt) {
    const _Float16 hq = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
    *q = (float)hq;
    bool ok = false;
    *alt = *q;
    if (*q != v) *alt = f16_neighbour(__builtin_bit_cast(unsigned short, hq), *q < v, &ok);
    return ok;
  };
  float S = 0.f;
  for (int k = lane; k < row_len; k += 64) {
    const float v = w[base + k];
    float q, alt;
    const bool ok = neighbours(v, &q, &alt);
    const float d = q - v, da = alt - v;
    const float m = mu ? mu[k % mu_period] : 1.f;
    step[k] = ok ? m * (alt - q) : 0.f;
    cost[k] = da * da - d * d;
    out[base + k] = q;
    S += m * d;
  }
  // wave sum in a fixed order (butterfly): the same S on every lane
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) S += __shfl_xor(S, o, 64);
  __syncthreads();   // (one wave per block: orders this wave's LDS writes before its reads)
  for (int it = 0; it < max_iter; ++it) {
    // the flip that removes the most |S| per unit of added squared error, among those that move S towards zero
    // without overshooting past -S
    float best = 3.0e38f;
    int best_k = 0x7fffffff;
    const float aS = fabsf(S);
    for (int k = lane; k < row_len; k += 64) {
      const float st = step[k];
      if (st * S < 0.f && fabsf(st) < 2.f * aS) {
        const float gain = aS - fabsf(S + st);
        const float sc = cost[k] / fmaxf(gain, 1e-37f);
        if (sc < best) { best = sc; best_k = k; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int ok_ = __shfl_xor(best_k, o, 64);
      if (ob < best || (ob == best && ok_ < best_k)) { best = ob; best_k = ok_; }
    }
    if (best_k == 0x7fffffff) break;   // (uniform: every lane holds the same winner)
    const float st = step[best_k];
    S += st;
    if (lane == 0) {
      float q, alt;
      (void)neighbours(w[base + best_k], &q, &alt);
      out[base + best_k] = alt;
      step[best_k] = 0.f;
    }
    __syncthreads();
  }
}

// per-channel sums of a 16-bit NHWC tensor [rows][C] (C % 8 == 0): slice s adds rows s, s + S, ... -> part[s][C]
template <int DT>
__global__ void chan_sum_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, size_t rows, int C) {
  const int cx = blockIdx.x * blockDim.x + threadIdx.x;   // 16-byte chunk of the row
  if (cx * 8 >= C) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const u32x4_t v = *(const u32x4_t*)(x + r * C + cx * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[2 * j] += lo_f32<DT>(v[j]);
      acc[2 * j + 1] += hi_f32<DT>(v[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[(size_t)blockIdx.y * C + cx * 8 + j] = acc[j];
}

__global__ void chan_mean_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean, int slices, int C,
                                          float inv_rows) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int i = 0; i < slices; ++i) s += part[(size_t)i * C + c];   // fixed order
  mean[c] = s * inv_rows;
}

}  // namespace

int spk_launch_zero_sum_round(const float* w, const float* mu, float* out, size_t rows, int row_len, int mu_period,
                              hipStream_t s) {
  if (rows == 0 || row_len <= 0 || mu_period <= 0 || rows > 0x7fffffffull) return -2;
  const size_t lds = (size_t)row_len * 8;
  if (lds > 64 * 1024) return -2;
  // a row settles in about 2-5 % of its length in flips; the cap only bounds a pathological row
  const int max_iter = row_len < 64 ? row_len : (row_len / 4 > 64 ? row_len / 4 : 64);
  hipLaunchKernelGGL(zero_sum_round_kernel, dim3((unsigned)rows), dim3(64), lds, s, w, mu, out, row_len, mu_period,
                     max_iter);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_chan_mean_slices(size_t rows) { return (int)(rows < 256 ? (rows ? rows : 1) : 256); }

// mean[c] = (1 / rows) sum_r x[r][c]; part: scratch of spk_chan_mean_slices(rows) * C floats
int spk_launch_chan_mean(const bf16_t* x, float* part, float* mean, size_t rows, int C, int dt, hipStream_t s) {
  if (C % 8 || rows == 0) return -2;
  const int slices = spk_chan_mean_slices(rows);
  const dim3 grid((unsigned)((C / 8 + 63) / 64), (unsigned)slices);
  if (dt == DT_F16) hipLaunchKernelGGL(chan_sum_kernel<DT_F16>, grid, dim3(64), 0, s, x, part, rows, C);
  else hipLaunchKernelGGL(chan_sum_kernel<DT_BF16>, grid, dim3(64), 0, s, x, part, rows, C);
  hipLaunchKernelGGL(chan_mean_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, part, mean, slices, C,
                     (float)(1.0 / (double)rows));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
