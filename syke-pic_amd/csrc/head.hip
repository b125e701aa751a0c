// Classifier head on gfx950: the reference's head is a stack of Linear
// layers with NO activation between them (sykepic/train/network.py:58-63,
// quirk Q1), then either the base-1.3 softmax of net_pass
// (sykepic/compute/probability.py:191-194) or CrossEntropyLoss
// (sykepic/train/train.py:127,241).  The head is 0.01 % of the FLOPs, so it
// stays in exact fp32 (LDS-tiled FMA GEMM) — this keeps the logits' error
// budget for the bf16 backbone.
#include "spk_common.h"
#include <cstdlib>

namespace {

// C[i][j] = sum_k A(i,k) * B(j,k) (+ bias[j]);  A(i,k) = A[i*sai + k*sak], B likewise.
// 64x64 tile, 256 threads, 4x4 outputs per thread, K step 16.
// accumulate != 0: C += result (used for gradient accumulation).
__global__ __launch_bounds__(256) void sgemm_strided_kernel(
    const float* __restrict__ A, long sai, long sak, const float* __restrict__ B, long sbj, long sbk,
    const float* __restrict__ bias, float* __restrict__ C, long sci, long scj, int M, int N, int K,
    float alpha, int accumulate) {
  __shared__ float sa[16][64 + 4];
  __shared__ float sb[16][64 + 4];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int ty = tid >> 4, tx = tid & 15;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int e = tid; e < 64 * 16; e += 256) {
      // choose the element order so the fastest-varying global index is contiguous
      int r, kk;
      if (sak == 1) { kk = e & 15; r = e >> 4; } else { r = e & 63; kk = e >> 6; }
      const int gi = i0 + r, gk = k0 + kk;
      sa[kk][r] = (gi < M && gk < K) ? A[gi * sai + gk * sak] : 0.f;
    }
    for (int e = tid; e < 64 * 16; e += 256) {
      int r, kk;
      if (sbk == 1) { kk = e & 15; r = e >> 4; } else { r = e & 63; kk = e >> 6; }
      const int gj = j0 + r, gk = k0 + kk;
      sb[kk][r] = (gj < N && gk < K) ? B[gj * sbj + gk * sbk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = sa[kk][ty * 4 + u]; bv[u] = sb[kk][tx * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(av[u], bv[v], acc[u][v]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int gi = i0 + ty * 4 + u;
    if (gi >= M) continue;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int gj = j0 + tx * 4 + v;
      if (gj >= N) continue;
      float r = acc[u][v] * alpha + (bias ? bias[gj] : 0.f);
      float* c = C + gi * sci + gj * scj;
      *c = accumulate ? *c + r : r;
    }
  }
}

// The same product on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: an fmaf chain per output, k ascending) for ANY strides:
// a wave owns a 16 x 16 output tile, a block 2 x 2 of them; per 4-deep K step a lane supplies ONE element of each operand
// (row lane & 15, k = k0 + (lane >> 4)) - with a unit stride along the rows (the weight-gradient form: A = dY^T, B = X^T,
// K = the batch) the 16 lanes of a k read 64 contiguous bytes.  Eight K steps of loads in flight.  The head's backward
// GEMMs (2048 x 256 x 256 and smaller) took 37 us each on the LDS-tiled FMA kernel above, 0.22 ms of a ResNet-50 step.
__global__ __launch_bounds__(256) void sgemm_mfma_strided_kernel(
    const float* __restrict__ A, long sai, long sak, const float* __restrict__ B, long sbj, long sbk,
    const float* __restrict__ bias, float* __restrict__ C, long sci, long scj, int M, int N, int K,
    float alpha, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int i0 = blockIdx.y * 32 + (wave >> 1) * 16, j0 = blockIdx.x * 32 + (wave & 1) * 16;
  if (i0 >= M || j0 >= N) return;
  const float* pa = A + (long)min(i0 + r, M - 1) * sai;   // (rows past the edge: clamped, their outputs are not stored)
  const float* pb = B + (long)min(j0 + r, N - 1) * sbj;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;
  for (int k0 = 0; k0 < K; k0 += 4 * U) {
    float va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 4 * u + g;
      const bool ok = k < K;
      const int kc = ok ? k : K - 1;
      va[u] = pa[(long)kc * sak];
      vb[u] = pb[(long)kc * sbk];
      if (!ok) va[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[u], vb[u], acc, 0, 0, 0);
  }
  // C layout: col = lane & 15, row = (lane >> 4) * 4 + reg
  const int gj = j0 + r;
  if (gj >= N) return;
  const float bj = bias ? bias[gj] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int gi = i0 + g * 4 + e;
    if (gi >= M) continue;
    float* c = C + gi * sci + gj * scj;
    const float v = acc[e] * alpha + bj;
    *c = accumulate ? *c + v : v;
  }
}

// y[n][out] = x[n][in] . w[out][in]^T + b with the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain).  One 16x16 output tile per
// block, K split over the 4 waves and summed through LDS in fixed order.
// Operand trick: each lane loads a float4 (k = k0 + 4*(lane>>4) + 0..3) of its
// row; MFMA step s consumes element s of every lane, i.e. the k-set
// {k0+s, k0+4+s, k0+8+s, k0+12+s} — the same set for A and B, any order sums.
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ y, int n, int in,
                                                          int out) {
  __shared__ float red[4][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const int r = lane & 15, g = lane >> 4;
  const int ai = min(i0 + r, n - 1), bj = min(j0 + r, out - 1);  // clamp: extra rows discarded
  const float* pa = x + (size_t)ai * in;
  const float* pb = w + (size_t)bj * in;
  const int kper = ((in + 63) / 64) * 16;  // per-wave K range, multiple of 16
  const int kbeg = wave * kper, kend = min(in, kbeg + kper);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    const int k = k0 + 4 * g;
    f32x4_t va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
    if (k + 3 < kend) {
      va = *(const f32x4_t*)(pa + k);
      vb = *(const f32x4_t*)(pb + k);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < kend) { va[e] = pa[k + e]; vb[e] = pb[k + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[e], vb[e], acc, 0, 0, 0);
  }
  // C layout: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][g * 4 + e][r] = acc[e];
  __syncthreads();
  const int t = threadIdx.x;
  const int ri = t >> 4, cj = t & 15;
  const int gi = i0 + ri, gj = j0 + cj;
  if (gi < n && gj < out)
    y[(size_t)gi * out + gj] = ((red[0][ri][cj] + red[1][ri][cj]) + (red[2][ri][cj] + red[3][ri][cj])) +
                               (b ? b[gj] : 0.f);
}

__device__ __forceinline__ float wave_max(float v) {
  for (int d = 32; d; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// one wave per row: p = softmax(z * scale)
__global__ void softmax_kernel(const float* __restrict__ z, float* __restrict__ p, int n, int c,
                               float scale) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* zr = z + (size_t)row * c;
  float mx = -INFINITY;
  for (int j = lane; j < c; j += 64) mx = fmaxf(mx, zr[j] * scale);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < c; j += 64) s += expf(zr[j] * scale - mx);
  s = wave_sum(s);
  const float inv = 1.0f / s;
  for (int j = lane; j < c; j += 64) p[(size_t)row * c + j] = expf(zr[j] * scale - mx) * inv;
}

// Mean cross-entropy + arg-max accuracy + dlogits in ONE block so that the
// loss sum has a fixed summation order (bitwise reproducible).
__global__ __launch_bounds__(1024) void ce_kernel(const float* __restrict__ z,
                                                 const int64_t* __restrict__ y, int n, int c,
                                                 float* __restrict__ stats,
                                                 float* __restrict__ dz) {
  constexpr int WAVES = 16;   // a row per wave at a time: 4 waves walked 64 rows each, one latency chain per row (92 us)
  __shared__ float s_loss[WAVES];
  __shared__ float s_corr[WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float loss = 0.f, corr = 0.f;
  const float invn = 1.0f / (float)n;
  for (int row = wave; row < n; row += WAVES) {
    const float* zr = z + (size_t)row * c;
    const int label = (int)y[row];
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    for (int j = lane; j < c; j += 64) {
      const float v = zr[j];
      if (v > mx) { mx = v; arg = j; }
    }
    // arg-max with torch's tie rule: lowest index among the maxima
    for (int d = 32; d; d >>= 1) {
      const float om = __shfl_xor(mx, d);
      const int oa = __shfl_xor(arg, d);
      if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    float s = 0.f;
    for (int j = lane; j < c; j += 64) s += expf(zr[j] - mx);
    s = wave_sum(s);
    const float lse = mx + logf(s);
    if (lane == 0) {
      loss += lse - zr[label];
      corr += (arg == label) ? 1.f : 0.f;
    }
    if (dz) {
      const float inv = 1.0f / s;
      for (int j = lane; j < c; j += 64)
        dz[(size_t)row * c + j] = (expf(zr[j] - mx) * inv - (j == label ? 1.f : 0.f)) * invn;
    }
  }
  if (lane == 0) { s_loss[wave] = loss; s_corr[wave] = corr; }
  __syncthreads();
  if (threadIdx.x == 0) {   // fixed order
    float tl = 0.f, tc = 0.f;
    for (int q = 0; q < WAVES; ++q) { tl += s_loss[q]; tc += s_corr[q]; }
    stats[0] += tl;
    stats[1] += tc;
  }
}

// SURVEY.md §8f rank 2 — per-ROI prediction from probabilities and per-class
// thresholds (reference sykepic/compute/prediction.py:49-71): the
// highest-probability class that clears ITS OWN threshold wins (lowest index on
// ties); if none does, arg-max with classified = 0.  thr == nullptr: scalar
// rule, classified = p[argmax] > scalar_thr.  One wave per row.
__global__ void predict_kernel(const float* __restrict__ p, int n, int c, const float* __restrict__ thr,
                               float scalar_thr, int* __restrict__ pred, unsigned char* __restrict__ ok) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* pr = p + (size_t)row * c;
  float best = -1.f, bestq = -1.f;
  int arg = 0x7fffffff, argq = 0x7fffffff;
  for (int j = lane; j < c; j += 64) {
    const float v = pr[j];
    if (v > best) { best = v; arg = j; }
    if (thr && v >= thr[j] && v > bestq) { bestq = v; argq = j; }
  }
  for (int d = 32; d; d >>= 1) {
    const float ob = __shfl_xor(best, d), oq = __shfl_xor(bestq, d);
    const int oa = __shfl_xor(arg, d), oaq = __shfl_xor(argq, d);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    if (oq > bestq || (oq == bestq && oaq < argq)) { bestq = oq; argq = oaq; }
  }
  if (lane == 0) {
    if (thr) {
      const bool any = argq != 0x7fffffff;
      pred[row] = any ? argq : arg;
      ok[row] = any ? 1 : 0;
    } else {
      pred[row] = arg;
      ok[row] = best > scalar_thr ? 1 : 0;
    }
  }
}

}  // namespace

int spk_launch_predict(const float* p, int n, int c, const float* thr, float scalar_thr, int* pred,
                       unsigned char* ok, hipStream_t s) {
  hipLaunchKernelGGL(predict_kernel, dim3((n + 3) / 4), dim3(256), 0, s, p, n, c, thr, scalar_thr, pred, ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_sgemm(const float* A, long sai, long sak, const float* B, long sbj, long sbk,
                     const float* bias, float* C, long sci, long scj, int M, int N, int K,
                     float alpha, int accumulate, hipStream_t s) {
  static const bool fma = getenv("SPK_SGEMM_FMA") && atoi(getenv("SPK_SGEMM_FMA")) != 0;   // the LDS-tiled FMA kernel (A/B runs)
  if (!fma) {
    hipLaunchKernelGGL(sgemm_mfma_strided_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, s, A, sai, sak, B, sbj, sbk,
                       bias, C, sci, scj, M, N, K, alpha, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  hipLaunchKernelGGL(sgemm_strided_kernel, grid, dim3(256), 0, s, A, sai, sak, B, sbj, sbk, bias, C,
                     sci, scj, M, N, K, alpha, accumulate);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_linear_fwd(const float* x, const float* w, const float* b, float* y, int n, int in,
                          int out, hipStream_t s) {
  // y[n][out] = x[n][in] . w[out][in]^T + b
  if (in % 4) return spk_launch_sgemm(x, in, 1, w, in, 1, b, y, out, 1, n, out, in, 1.f, 0, s);
  hipLaunchKernelGGL(linear_mfma_kernel, dim3((out + 15) / 16, (n + 15) / 16), dim3(256), 0, s, x, w, b,
                     y, n, in, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_softmax(const float* z, float* p, int n, int c, float scale, hipStream_t s) {
  hipLaunchKernelGGL(softmax_kernel, dim3((n + 3) / 4), dim3(256), 0, s, z, p, n, c, scale);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_ce(const float* z, const int64_t* y, int n, int c, float* stats, float* dlogits,
                  hipStream_t s) {
  hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(1024), 0, s, z, y, n, c, stats, dlogits);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
