// HBM-bound kernels of the training step (gfx950), bf16 activations, fp32
// statistics and parameters.  They stand in for the torch autograd pieces the
// reference reaches through `net(x)` in train mode, `loss.backward()` and
// `optimizer.step()` (sykepic/train/train.py:240-243): BatchNorm2d (batch
// statistics, running-stat update, backward), ReLU / residual add, MaxPool2d
// and AdaptiveAvgPool2d backward, bias gradients, Adam / SGD.
//
// Layout: activations NHWC, so a tensor is [M pixels][C] with C contiguous;
// every kernel moves 16 B (8 channels) per lane.  Per-channel reductions are
// two-stage and ordered (block partials -> fixed-order finalize), never
// atomics, so a step is bitwise reproducible.
#include "spk_common.h"
#include "ordered_reduce.h"

namespace {

constexpr int DT = DT_BF16;

__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = lo_f32<DT>(v[j]); f[2 * j + 1] = hi_f32<DT>(v[j]); }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  u32x4_t v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = pack2<DT>(f[2 * j], f[2 * j + 1]);
  return v;
}

// ---- BatchNorm forward (train): what the last stage of the reduction does with a channel's two sums ----
// partials: [m_tiles][2][C] (sum, sum of squares) written by the conv epilogue
struct BnFwdFin {
  double M;
  const float *gamma, *beta;
  float *rmean, *rvar, *mean_out, *invstd_out, *scale, *shift;
  float eps, momentum;
  __device__ __forceinline__ void operator()(int c, double s1, double s2) const {
    const double mean = s1 / M;
    double var = s2 / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    // running stats: unbiased variance, momentum 0.1 (torch BatchNorm2d defaults)
    const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
};

// ---- BatchNorm backward, stage 2: dgamma, dbeta (into the flat grad buffer when wanted) + the three per-channel
// coefficients of the apply pass ----
struct BnBwdFin {
  int C;
  double M;
  const float *gamma, *invstd;
  float *dgamma, *dbeta, *coef;
  __device__ __forceinline__ void operator()(int c, double s1, double s2) const {
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    coef[c] = (float)(s1 / M);            // mean(dz)
    coef[C + c] = (float)(s2 / M);        // mean(dz * xhat)
    coef[2 * C + c] = gamma[c] * invstd[c];
  }
};

// a = relu?(y*scale + shift (+ res))
__global__ void bn_apply_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                const float* __restrict__ shift, const bf16_t* __restrict__ res,
                                bf16_t* __restrict__ a, unsigned char* __restrict__ mask, size_t n8, int C,
                                int relu) {
  const int c8 = C >> 3;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8) * 8;
    float v[8], r[8];
    unpack8(*(const u32x4_t*)(y + i * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * scale[cc + j] + shift[cc + j];
    if (res) {
      unpack8(*(const u32x4_t*)(res + i * 8), r);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    if (relu) {
      // one mask bit per element (1/16 of re-reading `a` in the two backward passes)
      unsigned bits = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bits |= (v[j] > 0.f ? 1u : 0u) << j;
        v[j] = fmaxf(v[j], 0.f);
      }
      if (mask) mask[i] = (unsigned char)bits;
    }
    *(u32x4_t*)(a + i * 8) = pack8(v);
  }
}

// ---- BatchNorm backward ----
// stage 1: per-block partial sums over rows of dz = g*(a>0) and dz*xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const bf16_t* __restrict__ g, const unsigned char* __restrict__ mask, const bf16_t* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ partials,
    int M, int C, int relu, int rows_per_block) {
  // A block owns rows x one channel tile (blockIdx.y of gridDim.y, <= 256 channels each): on the 7x7 / 14x14 layers
  // (2048 channels x 12544 rows) a thread used to walk ALL rows of its block for one 8-channel group, one row in flight,
  // on 196 blocks - latency-bound at a third of the rate of the apply pass.
  extern __shared__ float sm[];  // [rows_in_flight][2][TW]
  const int c8 = C >> 3;
  const int tile = (c8 + (int)gridDim.y - 1) / (int)gridDim.y, TW = tile * 8;
  const int cb = blockIdx.y * tile, ce = min(c8, cb + tile);
  const int tpr = tile < 256 ? tile : 256;   // threads per row
  const int rif = 256 / tpr;                 // rows in flight
  const int lane_c = threadIdx.x % tpr, lane_r = threadIdx.x / tpr;
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(M, row0 + rows_per_block);
  for (int cc = cb + lane_c; cc < ce && lane_r < rif; cc += tpr) {
    float s1[8], s2[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = s2[j] = 0.f; mu[j] = mean[cc * 8 + j]; is[j] = invstd[cc * 8 + j]; }
    for (int r = row0 + lane_r; r < row1; r += rif) {
      const size_t o = (size_t)r * C + cc * 8;
      float gv[8], yv[8];
      unpack8(*(const u32x4_t*)(g + o), gv);
      unpack8(*(const u32x4_t*)(y + o), yv);
      if (relu) {
        const unsigned bits = mask[o >> 3];
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = (bits >> j) & 1u ? gv[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += gv[j]; s2[j] += gv[j] * (yv[j] - mu[j]) * is[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sm[(lane_r * 2 + 0) * TW + (cc - cb) * 8 + j] = s1[j];
      sm[(lane_r * 2 + 1) * TW + (cc - cb) * 8 + j] = s2[j];
    }
  }
  __syncthreads();
  const int nch = (ce - cb) * 8;
  for (int i = threadIdx.x; i < 2 * nch; i += 256) {
    const int which = i >= nch, col = i - which * nch;
    float t = 0.f;
    for (int r = 0; r < rif; ++r) t += sm[(r * 2 + which) * TW + col];
    partials[((size_t)blockIdx.x * 2 + which) * C + cb * 8 + col] = t;
  }
}

// stage 3: dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); the
// shortcut branch receives dz itself (g_res = dz or += dz).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const bf16_t* __restrict__ g, const unsigned char* __restrict__ mask, const bf16_t* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ coef,
    bf16_t* __restrict__ dy, bf16_t* __restrict__ g_res, int res_accumulate, int M, int C, int relu, int rows_per_block) {
  // A thread owns ONE 8-channel chunk (the five per-channel constants live in registers) and walks rows: the flat-index
  // version paid a 64-bit `i % c8` and 40 per-channel loads for every 16 bytes of dy (bn_bwd 7.5 -> 6.8 ms per ResNet-50
  // step).  The same rewrite of bn_apply_kernel (two constants per channel) measured 3 % slower than its flat form, and
  // unrolling the row loop by four made both slower: neither is kept.
  const int c8 = C >> 3;
  const int tpr = c8 < 256 ? c8 : 256;
  const int rif = 256 / tpr;
  const int lane_c = threadIdx.x % tpr, lane_r = threadIdx.x / tpr;
  const int row0 = blockIdx.x * rows_per_block, row1 = min(M, row0 + rows_per_block);
  if (lane_r >= rif) return;
  for (int cc8 = lane_c; cc8 < c8; cc8 += tpr) {
    const int cc = cc8 * 8;
    float mu[8], is[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mean[cc + j]; is[j] = invstd[cc + j];
      k0[j] = coef[cc + j]; k1[j] = coef[C + cc + j]; k2[j] = coef[2 * C + cc + j];
    }
    for (int r = row0 + lane_r; r < row1; r += rif) {
      const size_t i = (size_t)r * c8 + cc8;
      float gv[8], yv[8], o[8];
      unpack8(*(const u32x4_t*)(g + i * 8), gv);
      unpack8(*(const u32x4_t*)(y + i * 8), yv);
      if (relu) {
        const unsigned bits = mask[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = (bits >> j) & 1u ? gv[j] : 0.f;
      }
      if (g_res) {
        if (res_accumulate) {
          float rv[8];
          unpack8(*(const u32x4_t*)(g_res + i * 8), rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) rv[j] += gv[j];
          *(u32x4_t*)(g_res + i * 8) = pack8(rv);
        } else {
          *(u32x4_t*)(g_res + i * 8) = pack8(gv);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (yv[j] - mu[j]) * is[j];
        o[j] = k2[j] * (gv[j] - k0[j] - xh * k1[j]);
      }
      *(u32x4_t*)(dy + i * 8) = pack8(o);
    }
  }
}

// ---- MaxPool2d(3,2,1) with saved arg-max, and its backward ----
__global__ void maxpool_idx_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                   unsigned char* __restrict__ idx, int n, int h, int w, int c, int k,
                                   int stride, int pad, int ho, int wo) {
  const int c8 = c >> 3;
  const size_t total = (size_t)n * ho * wo * c8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    size_t p = i / c8;
    const int ox = (int)(p % wo);
    p /= wo;
    const int oy = (int)(p % ho);
    const int img = (int)(p / ho);
    float best[8];
    unsigned char bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    for (int r = 0; r < k; ++r) {
      const int iy = oy * stride - pad + r;
      if ((unsigned)iy >= (unsigned)h) continue;
      for (int s = 0; s < k; ++s) {
        const int ix = ox * stride - pad + s;
        if ((unsigned)ix >= (unsigned)w) continue;
        float v[8];
        unpack8(*(const u32x4_t*)(x + (((size_t)img * h + iy) * w + ix) * c + cc * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (v[j] > best[j]) { best[j] = v[j]; bi[j] = (unsigned char)(r * k + s); }  // first max wins
      }
    }
    *(u32x4_t*)(y + i * 8) = pack8(best);
    u32x2_t pk;
    pk[0] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
    pk[1] = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
    *(u32x2_t*)(idx + i * 8) = pk;
  }
}

// gather form: every input pixel sums the gradients of the windows whose
// saved arg-max points at it (deterministic, no atomics).  blockIdx.x = (image, input row): the thread only splits its
// index into (column, channel group) - the flat-index version did three 64-bit divisions per element and a run-time
// `% stride` and `/ stride` per tap, 180 M VALU instructions per ResNet-50 launch (306 us for 565 MB).
template <int K, int S, int P>   // K == 0: run-time k / stride / pad
__global__ void maxpool_bwd_kernel(const bf16_t* __restrict__ gy, const unsigned char* __restrict__ idx,
                                   bf16_t* __restrict__ gx, int n, int h, int w, int c, int k_rt,
                                   int stride_rt, int pad_rt, int ho, int wo) {
  const int k = K ? K : k_rt, stride = K ? S : stride_rt, pad = K ? P : pad_rt;
  const unsigned c8 = (unsigned)c >> 3;
  const unsigned row_items = (unsigned)w * c8;
  const int img = blockIdx.x / h, iy = blockIdx.x - img * h;   // x: up to 2^31-1 blocks (y is limited to 65535)
  const bf16_t* gyi = gy + (size_t)img * ho * wo * c;
  const unsigned char* idxi = idx + (size_t)img * ho * wo * c;
  bf16_t* gxr = gx + ((size_t)img * h + iy) * w * c;
  for (unsigned i = blockIdx.y * blockDim.x + threadIdx.x; i < row_items; i += gridDim.y * blockDim.x) {
    const unsigned ix = i / c8, cc = i - ix * c8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < (K ? K : 1); ++r)
      for (int rr = K ? r : 0; rr < (K ? r + 1 : k); ++rr) {
        const int ty = iy + pad - rr;
        if (ty < 0 || ty % stride) continue;
        const int oy = ty / stride;
        if (oy >= ho) continue;
#pragma unroll
        for (int q = 0; q < (K ? K : 1); ++q)
          for (int s = K ? q : 0; s < (K ? q + 1 : k); ++s) {
            const int tx = (int)ix + pad - s;
            if (tx < 0 || tx % stride) continue;
            const int ox = tx / stride;
            if (ox >= wo) continue;
            const size_t o = (((size_t)oy * wo + ox) * c8 + cc) * 8;
            const u32x2_t pk = *(const u32x2_t*)(idxi + o);
            float gv[8];
            unpack8(*(const u32x4_t*)(gyi + o), gv);
            const unsigned want = (unsigned)(rr * k + s);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const unsigned bsel = (pk[j >> 2] >> ((j & 3) * 8)) & 0xffu;
              acc[j] += (bsel == want) ? gv[j] : 0.f;
            }
          }
      }
    *(u32x4_t*)(gxr + (size_t)i * 8) = pack8(acc);
  }
}

// The 3x3 / stride 2 / pad 1 form (every torchvision ResNet's stem pool), one thread per PAIR of input pixels (2a, 2a + 1)
// of a row and 8 channels.  The pair sees window columns a (taps 1 and 2) and a + 1 (tap 0) of the one or two window rows
// that cover input row iy - the same for every lane, so there is no divergence (the per-pixel kernel above runs all three
// tap columns with half the lanes masked and is VALU-bound at 274 us for ResNet-50's 565 MB at batch 256) and each
// (arg-max bytes, gradient) pair is loaded once for the two pixels.  Same sums in the same order as the per-pixel kernel.
__global__ __launch_bounds__(256) void maxpool3s2_bwd_pair_kernel(const bf16_t* __restrict__ gy,
                                                                  const unsigned char* __restrict__ idx,
                                                                  bf16_t* __restrict__ gx, int h, int w, int c, int ho, int wo) {
  const unsigned c8 = (unsigned)c >> 3;
  const unsigned row_items = (unsigned)(w >> 1) * c8;
  const int img = blockIdx.x / h, iy = blockIdx.x - img * h;
  const bf16_t* gyi = gy + (size_t)img * ho * wo * c;
  const unsigned char* idxi = idx + (size_t)img * ho * wo * c;
  bf16_t* gxr = gx + ((size_t)img * h + iy) * w * c;
  // window rows over input row iy: odd iy -> rows (iy + 1) / 2 at tap row 0 and (iy - 1) / 2 at tap row 2; even -> iy / 2 at tap row 1
  const int odd = iy & 1;
  const int oyA = odd ? (iy + 1) >> 1 : iy >> 1, trA = odd ? 0 : 1;   // first in the per-pixel kernel's order (tap row ascending)
  const int oyB = (iy - 1) >> 1;                                      // second (odd rows only), tap row 2
  for (unsigned i = blockIdx.y * blockDim.x + threadIdx.x; i < row_items; i += gridDim.y * blockDim.x) {
    const unsigned a = i / c8, cc = i - a * c8;
    float e[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // pixels 2a and 2a + 1
    for (int pass = 0; pass <= odd; ++pass) {
      const int oy = pass ? oyB : oyA, tr = pass ? 2 : trA;
      if (oy >= ho) continue;
      const size_t o0 = (((size_t)oy * wo + a) * c8 + cc) * 8;
      const bool has1 = (int)a + 1 < wo;
      const size_t o1 = has1 ? o0 + (size_t)c8 * 8 : o0;
      const u32x2_t p0 = *(const u32x2_t*)(idxi + o0), p1 = *(const u32x2_t*)(idxi + o1);
      float g0[8], g1[8];
      unpack8(*(const u32x4_t*)(gyi + o0), g0);
      unpack8(*(const u32x4_t*)(gyi + o1), g1);
      const unsigned t0 = (unsigned)(tr * 3);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned b0 = (p0[j >> 2] >> ((j & 3) * 8)) & 0xffu, b1 = (p1[j >> 2] >> ((j & 3) * 8)) & 0xffu;
        // per-pixel order: tap column ascending.  Even pixel 2a: window a at tap column 1.  Odd pixel 2a + 1: window a + 1 at
        // tap column 0, then window a at tap column 2.
        e[j] += (b0 == t0 + 1) ? g0[j] : 0.f;
        o[j] += (has1 && b1 == t0) ? g1[j] : 0.f;
        o[j] += (b0 == t0 + 2) ? g0[j] : 0.f;
      }
    }
    *(u32x4_t*)(gxr + ((size_t)(2 * a) * c8 + cc) * 8) = pack8(e);
    *(u32x4_t*)(gxr + ((size_t)(2 * a + 1) * c8 + cc) * 8) = pack8(o);
  }
}

// AdaptiveAvgPool2d(1) backward: gx[n,hw,c] = gy[n,c] / hw
__global__ void gavgpool_bwd_kernel(const float* __restrict__ gy, bf16_t* __restrict__ gx, int n,
                                    int hw, int c) {
  const int c8 = c >> 3;
  const size_t total = (size_t)n * hw * c8;
  const float inv = 1.0f / (float)hw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int img = (int)(i / ((size_t)hw * c8));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gy[(size_t)img * c + cc * 8 + j] * inv;
    *(u32x4_t*)(gx + i * 8) = pack8(v);
  }
}

// bias gradient: db[j] = sum_i dy[i][j]   (one thread per column, ordered)
__global__ void colsum_kernel(const float* __restrict__ dy, float* __restrict__ db, int n, int c) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= c) return;
  // 8 independent partial sums (8 loads in flight instead of a 256-long chain), combined in a fixed tree
  float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int i = 0;
  for (; i + 8 <= n; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += dy[(size_t)(i + u) * c + j];
  }
  for (; i < n; ++i) p[0] += dy[(size_t)i * c + j];
  db[j] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}

// split-K wgrad slabs -> gradient, fixed order.  8 independent partial sums
// keep 8 loads in flight per thread (the slabs are read exactly once); they
// are combined in a fixed tree, so the result is reproducible.
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t n,
                                   int splits, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += slabs[(size_t)(k + u) * n + i];
    }
    for (int u = 0; k < splits; ++k, ++u) p[u] += slabs[(size_t)k * n + i];
    out[i] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * scale;
  }
}

// stem wgrad image [Cout][8 rows][8 taps][4 ch] (x splits) -> [Cout][7][7][3]
__global__ void stem_wgrad_unpack_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                         int cout, int kh, int kw, int cin, int splits, float scale) {
  const int total = cout * kh * kw * cin;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ci = i % cin, s = (i / cin) % kw, r = (i / (cin * kw)) % kh, co = i / (cin * kw * kh);
  const size_t src = (size_t)co * 256 + r * 32 + (s + 1) * 4 + ci;
  float t = 0.f;
  for (int k = 0; k < splits; ++k) t += slabs[(size_t)k * cout * 256 + src];
  out[i] = t * scale;
}

// master fp32 [Cout][kh][kw][Cin] -> bf16 dgrad image [Cin][kh][kw][Cout]
__global__ void pack_dgrad_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout,
                                  int taps, int cin) {
  const size_t n = (size_t)cout * taps * cin;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % cout);
    const int t = (int)((i / cout) % taps);
    const int ci = (int)(i / ((size_t)cout * taps));
    out[i] = to_h16<DT>(w[((size_t)co * taps + t) * cin + ci]);
  }
}

// Both 16-bit weight images of up to 64 conv layers in ONE launch (every training step repacks every conv whose
// master weights the optimizer moved: 2 x 53 tiny launches for ResNet-50 otherwise).  blockIdx.y selects a table
// entry; kind 0: forward image [Cout][K] = the master layout, an element-wise conversion; kind 1: data-gradient
// image [Cin][taps][Cout].
// Element-wise entries move 8 values per lane (two 16-byte loads, one 16-byte store); the transposed image goes through a
// 64 x 64 LDS tile per (tap, cout block, cin block): 256-byte reads along cin, 128-byte writes along cout.  (One value per
// lane and, for the transposed image, reads strided by taps x cin floats: 105 us per launch, two per ResNet-50 step, for
// 306 MB - 0.21 ms; shapes that are not multiples of 64 keep that form.)
__global__ __launch_bounds__(256) void pack_multi_kernel(const float* __restrict__ pbuf, bf16_t* __restrict__ wpack, PackTable t) {
  __shared__ float tile[64][65];
  const PackEntry e = t.e[blockIdx.y];
  const float* w = pbuf + e.src;
  bf16_t* out = wpack + e.dst;
  const unsigned n = e.cout * e.taps * e.cin;
  if (e.kind == 0) {
    const unsigned n8 = ((e.src | e.dst) & 7) == 0 ? n >> 3 : 0;   // 16-byte aligned on both sides
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += gridDim.x * blockDim.x) {
      const f32x4_t a = *(const f32x4_t*)(w + (size_t)i * 8), b = *(const f32x4_t*)(w + (size_t)i * 8 + 4);
      u32x4_t o;
      o[0] = (unsigned)to_h16<DT>(a[0]) | ((unsigned)to_h16<DT>(a[1]) << 16);
      o[1] = (unsigned)to_h16<DT>(a[2]) | ((unsigned)to_h16<DT>(a[3]) << 16);
      o[2] = (unsigned)to_h16<DT>(b[0]) | ((unsigned)to_h16<DT>(b[1]) << 16);
      o[3] = (unsigned)to_h16<DT>(b[2]) | ((unsigned)to_h16<DT>(b[3]) << 16);
      *(u32x4_t*)(out + (size_t)i * 8) = o;
    }
    for (unsigned i = n8 * 8 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = to_h16<DT>(w[i]);
    return;
  }
  if ((e.cout & 63) || (e.cin & 63)) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const unsigned co = i % e.cout, tt = (i / e.cout) % e.taps, ci = i / (e.cout * e.taps);
      out[i] = to_h16<DT>(w[((size_t)co * e.taps + tt) * e.cin + ci]);
    }
    return;
  }
  const unsigned cbs = e.cout >> 6, ibs = e.cin >> 6, tiles = cbs * ibs * e.taps;
  const unsigned tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (unsigned q = blockIdx.x; q < tiles; q += gridDim.x) {
    const unsigned ib = q % ibs, cbk = (q / ibs) % cbs, tt = q / (ibs * cbs);
    const unsigned c0 = cbk * 64, i0 = ib * 64;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = w[((size_t)(c0 + ty + 4 * r) * e.taps + tt) * e.cin + i0 + tx];
    __syncthreads();   // the previous tile has been read out
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[ty + 4 * r][tx] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[((size_t)(i0 + ty + 4 * r) * e.taps + tt) * e.cout + c0 + tx] = to_h16<DT>(tile[tx][ty + 4 * r]);
  }
}

// ---- optimizers over one tensor of the flat buffers ----
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                           size_t n, float lr, float wd, float momentum, float gscale, int first) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float d = g[i] * gscale + wd * p[i];
    if (momentum != 0.f) {
      const float b = first ? d : momentum * mom[i] + d;
      mom[i] = b;
      d = b;
    }
    p[i] -= lr * d;
  }
}

// torch.optim.Adam (amsgrad=False, maximize=False): bias corrections from the
// per-tensor step count
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                            float wd, float gscale, float bc1, float bc2_sqrt) {
  const float step_size = lr / bc1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale + wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

// multi-tensor forms: blockIdx.y selects a tensor of the flat buffers
__global__ void sgd_multi_kernel(float* __restrict__ p, const float* __restrict__ g,
                                 float* __restrict__ mom, OptTable t, float wd, float momentum,
                                 float gscale) {
  const OptEntry e = t.e[blockIdx.y];
  float* pp = p + e.off;
  const float* gg = g + e.off;
  float* mm = mom ? mom + e.off : nullptr;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < e.n; i += gridDim.x * blockDim.x) {
    float d = gg[i] * gscale + wd * pp[i];
    if (momentum != 0.f) {
      const float b = e.first ? d : momentum * mm[i] + d;
      mm[i] = b;
      d = b;
    }
    pp[i] -= e.lr * d;
  }
}

__global__ void adam_multi_kernel(float* __restrict__ p, const float* __restrict__ g,
                                  float* __restrict__ m, float* __restrict__ v, OptTable t, float b1,
                                  float b2, float eps, float wd, float gscale) {
  const OptEntry e = t.e[blockIdx.y];
  float* pp = p + e.off;
  const float* gg = g + e.off;
  float* mm = m + e.off;
  float* vv = v + e.off;
  const float step_size = e.lr / e.bc1;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < e.n; i += gridDim.x * blockDim.x) {
    const float gi = gg[i] * gscale + wd * pp[i];
    const float mi = b1 * mm[i] + (1.f - b1) * gi;
    const float vi = b2 * vv[i] + (1.f - b2) * gi * gi;
    mm[i] = mi;
    vv[i] = vi;
    pp[i] -= step_size * mi / (sqrtf(vi) / e.bc2s + eps);
  }
}

// The other first-order torch.optim classes (reference: `getattr(optim, name)(groups)` with torch's default
// hyper-parameters, sykepic/train/train.py:131-138), one tensor of the flat buffers per blockIdx.y.  The per-tensor
// scalars that depend on the step count come precomputed in the table entry (bc1, bc2s, c0, c1).
//   2 AdamW     p *= 1 - lr*wd;  then Adam
//   3 RMSprop   v = a v + (1-a) g^2;  d = g / (sqrt(v) + eps);  momentum: buf = mom*buf + d, d = buf;  p -= lr d
//   4 Adagrad   v (+)= g^2 (first: init + g^2);  p -= c0 * g / (sqrt(v) + eps)              c0 = lr / (1 + (t-1) lr_decay)
//   5 Adamax    m = b1 m + (1-b1) g;  u = max(b2 u, |g| + eps);  p -= (lr / bc1) m / u
//   6 NAdam     m, v as Adam;  den = sqrt(v)/bc2s + eps;  p -= c0 g/den + c1 m/den   c0 = lr(1-mu_t)/(1-prod), c1 = lr mu_next/(1-prod mu_next)
//   7 RAdam     m, v as Adam;  c1 > 0 (rho_t > 5): p -= (m/bc1) c0 / (sqrt(v) + eps), c0 = lr rect sqrt(bc2);  else p -= lr m/bc1
//   8 Adadelta  sq = rho sq + (1-rho) g^2;  d = sqrt(acc + eps)/sqrt(sq + eps) g;  acc = rho acc + (1-rho) d^2;  p -= lr d
//   9 ASGD      p = p (1 - lambd eta) - eta g,  eta = lr / (1 + lambd lr (t-1))^alpha  (c0 = eta, c1 = lambd; the averaged
//               copy `ax` torch keeps beside the parameters is never read by a training loop and is not kept)
//  10 Rprop     s = g prev: > 0 step *= eta+, < 0 step *= eta-, g = 0;  step clamped to [min, max];  p -= sign(g) step;
//               prev = g   (mm = prev, vv = step size, first step: step size = lr; b1 = eta-, b2 = eta+, eps / alpha = bounds)
__global__ void opt_generic_kernel(int kind, float* __restrict__ p, const float* __restrict__ g,
                                   float* __restrict__ m, float* __restrict__ v, OptTable t, float b1, float b2,
                                   float eps, float wd, float momentum, float gscale, float alpha) {
  const OptEntry e = t.e[blockIdx.y];
  float* pp = p + e.off;
  const float* gg = g + e.off;
  float* mm = m + e.off;
  float* vv = v + e.off;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < e.n; i += gridDim.x * blockDim.x) {
    float pi = pp[i];
    float gi = gg[i] * gscale;
    if (kind == 2) pi *= 1.f - e.lr * wd;
    else if (kind != 10) gi += wd * pi;
    switch (kind) {
      case 9: {
        pi = pi * (1.f - e.c1 * e.c0) - e.c0 * gi;
        break;
      }
      case 10: {
        const float prev = e.first ? 0.f : mm[i];
        float ss = e.first ? e.lr : vv[i];
        const float sg = gi * prev;
        ss *= sg > 0.f ? b2 : (sg < 0.f ? b1 : 1.f);
        ss = fminf(fmaxf(ss, eps), alpha);
        if (sg < 0.f) gi = 0.f;
        pi -= (gi > 0.f ? ss : (gi < 0.f ? -ss : 0.f));
        mm[i] = gi;
        vv[i] = ss;
        break;
      }
      case 2: {
        const float mi = b1 * mm[i] + (1.f - b1) * gi, vi = b2 * vv[i] + (1.f - b2) * gi * gi;
        mm[i] = mi; vv[i] = vi;
        pi -= (e.lr / e.bc1) * mi / (sqrtf(vi) / e.bc2s + eps);
        break;
      }
      case 3: {
        const float vi = alpha * vv[i] + (1.f - alpha) * gi * gi;
        vv[i] = vi;
        float d = gi / (sqrtf(vi) + eps);
        if (momentum > 0.f) { d = (e.first ? 0.f : momentum * mm[i]) + d; mm[i] = d; }
        pi -= e.lr * d;
        break;
      }
      case 4: {
        const float vi = (e.first ? e.c1 : vv[i]) + gi * gi;
        vv[i] = vi;
        pi -= e.c0 * gi / (sqrtf(vi) + eps);
        break;
      }
      case 5: {
        const float mi = b1 * mm[i] + (1.f - b1) * gi;
        const float ui = fmaxf(b2 * vv[i], fabsf(gi) + eps);
        mm[i] = mi; vv[i] = ui;
        pi -= (e.lr / e.bc1) * mi / ui;
        break;
      }
      case 6: {
        const float mi = b1 * mm[i] + (1.f - b1) * gi, vi = b2 * vv[i] + (1.f - b2) * gi * gi;
        mm[i] = mi; vv[i] = vi;
        const float den = sqrtf(vi) / e.bc2s + eps;
        pi -= e.c0 * gi / den;
        pi -= e.c1 * mi / den;
        break;
      }
      case 7: {
        const float mi = b1 * mm[i] + (1.f - b1) * gi, vi = b2 * vv[i] + (1.f - b2) * gi * gi;
        mm[i] = mi; vv[i] = vi;
        const float mh = mi / e.bc1;
        pi -= e.c1 > 0.f ? mh * e.c0 / (sqrtf(vi) + eps) : mh * e.lr;
        break;
      }
      default: {  // 8 Adadelta: mm = square_avg, vv = acc_delta
        const float sq = alpha * mm[i] + (1.f - alpha) * gi * gi;
        const float acc = vv[i];
        const float d = sqrtf(acc + eps) / sqrtf(sq + eps) * gi;
        mm[i] = sq;
        vv[i] = alpha * acc + (1.f - alpha) * d * d;
        pi -= e.lr * d;
        break;
      }
    }
    pp[i] = pi;
  }
}

// ---- Dropout(p) in the head (reference network.py:59-61), train mode: y = x * keep / (1 - p) ----
// keep ~ Bernoulli(1 - p) from a counter hash of (seed, element index): reproducible for a given seed
// (spk_model_set_seed) and step, independent of the launch geometry.  torch draws from its own generator, so the
// masks differ from a torch run element by element (as they do between two torch runs with different seeds).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                   unsigned char* __restrict__ mask, size_t n, float p, unsigned long long seed) {
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float u = (float)(mix64(seed + 0x9E3779B97F4A7C15ull * (i + 1)) >> 40) * (1.0f / 16777216.0f);  // [0, 1)
    const bool keep = u >= p;
    mask[i] = keep ? 1 : 0;
    y[i] = keep ? x[i] * scale : 0.f;
  }
}
__global__ void dropout_bwd_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ mask,
                                   float* __restrict__ gx, size_t n, float p) {
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    gx[i] = mask[i] ? gy[i] * scale : 0.f;
}

// rows per block of the row-walking BatchNorm backward apply kernel: each thread takes >= 8 rows (its per-channel
// constants are loaded once), about 16 blocks per CU over the tensor
inline int pointwise_rows(int M, int C, int* rows_per_block) {
  const int c8 = C >> 3, tpr = c8 < 256 ? c8 : 256, rif = 256 / tpr;
  int rpb = rif * 8;
  while ((M + rpb - 1) / rpb > 256 * 16) rpb *= 2;
  *rows_per_block = rpb;
  return (M + rpb - 1) / rpb;
}

inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  if (g > 256 * 8 * 4) g = 256 * 8 * 4;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -1)

int spk_launch_bn_finalize(const float* partials, int m_tiles, int C, double M, const float* gamma,
                           const float* beta, float* rmean, float* rvar, float* mean, float* invstd,
                           float* scale, float* shift, float eps, float momentum, float* tmp,
                           hipStream_t s) {
  const BnFwdFin fin = {M, gamma, beta, rmean, rvar, mean, invstd, scale, shift, eps, momentum};
  return spk_reduce::reduce_finalize(partials, m_tiles, C, tmp, fin, s);
}

int spk_launch_bn_apply(const bf16_t* y, const float* scale, const float* shift, const bf16_t* res,
                        bf16_t* a, unsigned char* mask, size_t numel, int C, int relu, hipStream_t s) {
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(numel / 8, 256)), dim3(256), 0, s, y, scale, shift,
                     res, a, mask, numel / 8, C, relu);
  return LAUNCH_OK();
}

int spk_bn_bwd_blocks(int M, int C, int* rows_per_block) {
  int rpb = 1024;
  while (rpb > 64 && (M + rpb - 1) / rpb < 1024) rpb >>= 1;
  *rows_per_block = rpb;
  return (M + rpb - 1) / rpb;
}

int spk_launch_bn_bwd(const bf16_t* g, const unsigned char* a, const bf16_t* y, const float* mean,
                      const float* invstd, const float* gamma, float* partials, float* coef,
                      float* dgamma, float* dbeta, bf16_t* dy, bf16_t* g_res, int res_accumulate, int M,
                      int C, int relu, float* tmp, hipStream_t s, int pre_blocks) {
  int rpb;
  int nb = spk_bn_bwd_blocks(M, C, &rpb);
  if (pre_blocks > 0) {
    nb = pre_blocks;   // the sums came with the gradient (conv_igemm.hip dgrad epilogue)
  } else {
    static const bool tiled = !getenv("SPK_BN_CTILES") || atoi(getenv("SPK_BN_CTILES")) != 0;   // 0: one tile (round 2)
    const int c8 = C / 8, cts = tiled ? (c8 + 31) / 32 : 1, tile = (c8 + cts - 1) / cts;
    const int tpr = tile < 256 ? tile : 256;
    const size_t lds = (size_t)(256 / tpr) * 2 * tile * 8 * sizeof(float);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb, cts), dim3(256), lds, s, g, a, y, mean, invstd, partials,
                       M, C, relu, rpb);
  }
  const BnBwdFin fin = {C, (double)M, gamma, invstd, dgamma, dbeta, coef};
  if (spk_reduce::reduce_finalize(partials, nb, C, tmp, fin, s)) return -1;
  int arpb = 0;
  const int anb = pointwise_rows(M, C, &arpb);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(anb), dim3(256), 0, s, g, a, y, mean, invstd, coef, dy, g_res,
                     res_accumulate, M, C, relu, arpb);
  return LAUNCH_OK();
}

int spk_launch_maxpool_idx(const bf16_t* x, bf16_t* y, unsigned char* idx, int n, int h, int w, int c,
                           int k, int stride, int pad, int ho, int wo, hipStream_t s) {
  const size_t total = (size_t)n * ho * wo * (c / 8);
  hipLaunchKernelGGL(maxpool_idx_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, x, y, idx, n, h, w,
                     c, k, stride, pad, ho, wo);
  return LAUNCH_OK();
}

int spk_launch_maxpool_bwd(const bf16_t* gy, const unsigned char* idx, bf16_t* gx, int n, int h, int w,
                           int c, int k, int stride, int pad, int ho, int wo, hipStream_t s) {
  const unsigned row_items = (unsigned)w * (c / 8);
  const dim3 grid((unsigned)n * h, std::min(65535u, (row_items + 255) / 256));
  static const bool pair_on = !getenv("SPK_POOL_PAIR") || atoi(getenv("SPK_POOL_PAIR")) != 0;
  if (k == 3 && stride == 2 && pad == 1 && pair_on && w % 2 == 0 && wo == w / 2 && ho == (h + 1) / 2) {
    const unsigned items = (unsigned)(w / 2) * (c / 8);
    hipLaunchKernelGGL(maxpool3s2_bwd_pair_kernel, dim3((unsigned)n * h, (items + 255) / 256), dim3(256), 0, s, gy, idx, gx,
                       h, w, c, ho, wo);
  } else if (k == 3 && stride == 2 && pad == 1)
    hipLaunchKernelGGL((maxpool_bwd_kernel<3, 2, 1>), grid, dim3(256), 0, s, gy, idx, gx, n, h, w, c, k, stride, pad, ho, wo);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<0, 0, 0>), grid, dim3(256), 0, s, gy, idx, gx, n, h, w, c, k, stride, pad, ho, wo);
  return LAUNCH_OK();
}

int spk_launch_gavgpool_bwd(const float* gy, bf16_t* gx, int n, int hw, int c, hipStream_t s) {
  const size_t total = (size_t)n * hw * (c / 8);
  hipLaunchKernelGGL(gavgpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, gy, gx, n, hw, c);
  return LAUNCH_OK();
}

int spk_launch_dropout_fwd(const float* x, float* y, unsigned char* mask, size_t n, float p, unsigned long long seed,
                           hipStream_t s) {
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, x, y, mask, n, p, seed);
  return LAUNCH_OK();
}

int spk_launch_dropout_bwd(const float* gy, const unsigned char* mask, float* gx, size_t n, float p, hipStream_t s) {
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, gy, mask, gx, n, p);
  return LAUNCH_OK();
}

int spk_launch_colsum(const float* dy, float* db, int n, int c, hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3((c + 63) / 64), dim3(64), 0, s, dy, db, n, c);
  return LAUNCH_OK();
}

int spk_launch_slab_reduce(const float* slabs, float* out, size_t n, int splits, hipStream_t s, float scale) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, slabs, out, n, splits, scale);
  return LAUNCH_OK();
}

int spk_launch_stem_wgrad_unpack(const float* slabs, float* out, int cout, int kh, int kw, int cin,
                                 int splits, hipStream_t s, float scale) {
  const int total = cout * kh * kw * cin;
  hipLaunchKernelGGL(stem_wgrad_unpack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, slabs, out,
                     cout, kh, kw, cin, splits, scale);
  return LAUNCH_OK();
}

int spk_launch_pack_dgrad(const float* w, bf16_t* out, int cout, int taps, int cin, hipStream_t s) {
  const size_t n = (size_t)cout * taps * cin;
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, w, out, cout, taps, cin);
  return LAUNCH_OK();
}

int spk_launch_pack_multi(const float* pbuf, bf16_t* wpack, const PackTable& t, hipStream_t s) {
  if (t.count <= 0) return 0;
  hipLaunchKernelGGL(pack_multi_kernel, dim3(48, t.count), dim3(256), 0, s, pbuf, wpack, t);
  return LAUNCH_OK();
}

int spk_launch_opt_multi(int kind, float* p, const float* g, float* m, float* v, const OptTable& t,
                         float b1, float b2, float eps, float wd, float momentum, float gscale, float alpha,
                         hipStream_t s) {
  if (t.count <= 0) return 0;
  dim3 grid(96, t.count);
  if (kind == 1)
    hipLaunchKernelGGL(adam_multi_kernel, grid, dim3(256), 0, s, p, g, m, v, t, b1, b2, eps, wd, gscale);
  else if (kind == 0)
    hipLaunchKernelGGL(sgd_multi_kernel, grid, dim3(256), 0, s, p, g, m, t, wd, momentum, gscale);
  else
    hipLaunchKernelGGL(opt_generic_kernel, grid, dim3(256), 0, s, kind, p, g, m, v, t, b1, b2, eps, wd, momentum,
                       gscale, alpha);
  return LAUNCH_OK();
}

int spk_launch_sgd(float* p, const float* g, float* mom, size_t n, float lr, float wd, float momentum,
                   float gscale, int first, hipStream_t s) {
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, mom, n, lr, wd, momentum,
                     gscale, first);
  return LAUNCH_OK();
}

int spk_launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                    float eps, float wd, float gscale, float bc1, float bc2_sqrt, hipStream_t s) {
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps,
                     wd, gscale, bc1, bc2_sqrt);
  return LAUNCH_OK();
}
