// Training kernels of the EfficientNet graphs (torchvision MBConv: expand 1x1 -> depthwise k3/k5 -> squeeze-excitation
// -> project 1x1, SiLU, stochastic depth on the residual branch).  The reference trains whatever torchvision model the
// config names (sykepic/train/network.py:48-55, train.py:239-243); these stand in for the autograd pieces that the
// ResNet kernels of train_kernels.hip do not cover: depthwise Conv2d forward / data gradient / weight gradient, the 3x3
// RGB stem, BatchNorm with SiLU (and a per-image stochastic-depth factor) forward and backward, squeeze-excitation
// forward and backward.
//
// Layout: bf16 NHWC with the channel count padded to a multiple of 64 (pad channels hold zeros and have zero weights,
// scale and shift), so the 1x1 convs run on the implicit-GEMM / wgrad kernels unchanged; fp32 statistics and
// parameters; per-channel reductions are two-stage and ordered (no atomics): a step is bitwise reproducible.
#include "train_effnet.h"

#include <algorithm>

namespace {

constexpr int DT = DT_BF16;
constexpr int SPK_ACT_RELU = 1, SPK_ACT_SILU = 2;   // include/sykepic_hip.h

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
__device__ __forceinline__ float bf16_round(float f) { return lo_f32<DT>(pack2<DT>(f, 0.f)); }
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }
__device__ __forceinline__ float act_fwd(float z, int act) {
  return act == SPK_ACT_SILU ? z * sigmoidf_(z) : (act == SPK_ACT_RELU ? fmaxf(z, 0.f) : z);
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == SPK_ACT_SILU) {
    const float s = sigmoidf_(z);
    return s * (1.f + z * (1.f - s));
  }
  return act == SPK_ACT_RELU ? (z > 0.f ? 1.f : 0.f) : 1.f;
}

// A thread owns 8-channel chunks (lane_c, lane_c + tpr, ...) and walks rows lane_r, lane_r + rif, ... of its block
struct RowWalk {
  int c8, tpr, rif, lane_c, lane_r, row0, row1;
  bool active;
  __device__ RowWalk(int M, int C, int rows_per_block) {
    c8 = C >> 3;
    tpr = c8 < 256 ? c8 : 256;
    rif = 256 / tpr;
    lane_c = threadIdx.x % tpr;
    lane_r = threadIdx.x / tpr;
    active = lane_r < rif;
    row0 = blockIdx.x * rows_per_block;
    row1 = min(M, row0 + rows_per_block);
  }
};

// ---- per-channel sum / sum of squares of a bf16 [M][C] tensor: partials[block][2][C] (the layout bn_finalize reads)
__global__ __launch_bounds__(256) void col_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ partials,
                                                        int M, int C, int rows_per_block) {
  extern __shared__ float sm[];  // [rif][2][C]
  const RowWalk w(M, C, rows_per_block);
  if (w.active)
    for (int cc = w.lane_c; cc < w.c8; cc += w.tpr) {
      float s1[8], s2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
      for (int r = w.row0 + w.lane_r; r < w.row1; r += w.rif) {
        float v[8];
        unpack8(*(const u32x4_t*)(x + (size_t)r * C + cc * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[(w.lane_r * 2 + 0) * C + cc * 8 + j] = s1[j];
        sm[(w.lane_r * 2 + 1) * C + cc * 8 + j] = s2[j];
      }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int r = 0; r < w.rif; ++r) t += sm[r * 2 * C + i];
    partials[(size_t)blockIdx.x * 2 * C + i] = t;
  }
}

// ---- BatchNorm apply + activation (+ per-image factor) (+ shortcut): a = act(raw*scale + shift) * rs[img] + res
__global__ __launch_bounds__(256) void bna_apply_kernel(const bf16_t* __restrict__ raw, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const bf16_t* __restrict__ res,
                                                        const float* __restrict__ rowscale, bf16_t* __restrict__ out,
                                                        int M, int C, int HW, int act, int rows_per_block) {
  const RowWalk w(M, C, rows_per_block);
  if (!w.active) return;
  for (int cc = w.lane_c; cc < w.c8; cc += w.tpr) {
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cc * 8 + j]; sh[j] = shift[cc * 8 + j]; }
    for (int r = w.row0 + w.lane_r; r < w.row1; r += w.rif) {
      const size_t o = (size_t)r * C + cc * 8;
      float v[8];
      unpack8(*(const u32x4_t*)(raw + o), v);
      const float rs = rowscale ? rowscale[r / HW] : 1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], act) * rs;
      if (res) {
        float q[8];
        unpack8(*(const u32x4_t*)(res + o), q);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += q[j];
      }
      *(u32x4_t*)(out + o) = pack8(v);
    }
  }
}

// ---- its backward, stage 1: partial sums of dz = g * rs * act'(z) and dz * xhat
__global__ __launch_bounds__(256) void bna_bwd_reduce_kernel(
    const bf16_t* __restrict__ g, const bf16_t* __restrict__ raw, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ rowscale, float* __restrict__ partials, int M, int C, int HW, int act, int rows_per_block) {
  extern __shared__ float sm[];  // [rif][2][C]
  const RowWalk w(M, C, rows_per_block);
  if (w.active)
    for (int cc = w.lane_c; cc < w.c8; cc += w.tpr) {
      float s1[8], s2[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s1[j] = s2[j] = 0.f;
        sc[j] = scale[cc * 8 + j]; sh[j] = shift[cc * 8 + j]; mu[j] = mean[cc * 8 + j]; is[j] = invstd[cc * 8 + j];
      }
      for (int r = w.row0 + w.lane_r; r < w.row1; r += w.rif) {
        const size_t o = (size_t)r * C + cc * 8;
        float gv[8], yv[8];
        unpack8(*(const u32x4_t*)(g + o), gv);
        unpack8(*(const u32x4_t*)(raw + o), yv);
        const float rs = rowscale ? rowscale[r / HW] : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dz = gv[j] * rs * act_grad(yv[j] * sc[j] + sh[j], act);
          s1[j] += dz;
          s2[j] += dz * (yv[j] - mu[j]) * is[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[(w.lane_r * 2 + 0) * C + cc * 8 + j] = s1[j];
        sm[(w.lane_r * 2 + 1) * C + cc * 8 + j] = s2[j];
      }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int r = 0; r < w.rif; ++r) t += sm[r * 2 * C + i];
    partials[(size_t)blockIdx.x * 2 * C + i] = t;
  }
}

// stage 3: dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); the shortcut receives g itself
__global__ __launch_bounds__(256) void bna_bwd_apply_kernel(
    const bf16_t* __restrict__ g, const bf16_t* __restrict__ raw, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ coef, const float* __restrict__ rowscale, bf16_t* __restrict__ dy,
    bf16_t* __restrict__ g_res, int res_accumulate, int M, int C, int HW, int act, int rows_per_block) {
  const RowWalk w(M, C, rows_per_block);
  if (!w.active) return;
  for (int cc = w.lane_c; cc < w.c8; cc += w.tpr) {
    float sc[8], sh[8], mu[8], is[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cc * 8 + j;
      sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
      k0[j] = coef[c]; k1[j] = coef[C + c]; k2[j] = coef[2 * C + c];
    }
    for (int r = w.row0 + w.lane_r; r < w.row1; r += w.rif) {
      const size_t o = (size_t)r * C + cc * 8;
      float gv[8], yv[8], ov[8];
      unpack8(*(const u32x4_t*)(g + o), gv);
      unpack8(*(const u32x4_t*)(raw + o), yv);
      if (g_res) {
        if (res_accumulate) {
          float rv[8];
          unpack8(*(const u32x4_t*)(g_res + o), rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) rv[j] += gv[j];
          *(u32x4_t*)(g_res + o) = pack8(rv);
        } else {
          *(u32x4_t*)(g_res + o) = pack8(gv);
        }
      }
      const float rs = rowscale ? rowscale[r / HW] : 1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dz = gv[j] * rs * act_grad(yv[j] * sc[j] + sh[j], act);
        const float xh = (yv[j] - mu[j]) * is[j];
        ov[j] = k2[j] * (dz - k0[j] - xh * k1[j]);
      }
      *(u32x4_t*)(dy + o) = pack8(ov);
    }
  }
}

// ---- ordered finalize steps for a channel-padded tensor: parameters and running statistics exist for c < c_log only;
// pad channels get mean 0, invstd 0, scale 0, shift 0 (their raw values are zeros) ----
__device__ __forceinline__ double colsum16(const float* __restrict__ partials, int count, int C, int which, int c, int r,
                                           bool valid, double* sm) {
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

// stage 1 of the ordered reduction when there are many partial rows (one per M tile of the conv kernels): blockIdx.y takes
// a contiguous slice of the rows and writes one row of out[slices][2][C]
__global__ __launch_bounds__(1024) void bna_presum_kernel(const float* __restrict__ partials, int count, int C,
                                                          int rows_per_slice, float* __restrict__ out) {
  __shared__ double sm[16 * 64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const bool valid = c < C;
  const int t0 = blockIdx.y * rows_per_slice;
  const int cnt = max(0, min(count - t0, rows_per_slice));
  const float* base = partials + (size_t)t0 * 2 * C;
  const double s1 = colsum16(base, cnt, C, 0, c, r, valid, sm);
  const double s2 = colsum16(base, cnt, C, 1, c, r, valid, sm);
  if (r == 0 && valid) {
    out[((size_t)blockIdx.y * 2 + 0) * C + c] = (float)s1;
    out[((size_t)blockIdx.y * 2 + 1) * C + c] = (float)s2;
  }
}

__global__ __launch_bounds__(1024) void bna_finalize_kernel(
    const float* __restrict__ partials, int count, int C, int c_log, double M, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ st, float eps,
    float momentum) {
  __shared__ double sm[16 * 64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const bool valid = c < C;
  const double s1 = colsum16(partials, count, C, 0, c, r, valid, sm);
  const double s2 = colsum16(partials, count, C, 1, c, r, valid, sm);
  if (r != 0 || !valid) return;
  if (c >= c_log) {
    st[c] = 0.f; st[C + c] = 0.f; st[2 * C + c] = 0.f; st[3 * C + c] = 0.f;
    return;
  }
  const double mean = s1 / M;
  double var = s2 / M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  st[c] = (float)mean;
  st[C + c] = invstd;
  const float sc = gamma[c] * invstd;
  st[2 * C + c] = sc;
  st[3 * C + c] = beta[c] - (float)mean * sc;
  const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
  rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
}

__global__ __launch_bounds__(1024) void bna_bwd_finalize_kernel(
    const float* __restrict__ partials, int count, int C, int c_log, double M, const float* __restrict__ gamma,
    const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
  __shared__ double sm[16 * 64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const bool valid = c < C;
  const double s1 = colsum16(partials, count, C, 0, c, r, valid, sm);
  const double s2 = colsum16(partials, count, C, 1, c, r, valid, sm);
  if (r != 0 || !valid) return;
  if (c >= c_log) {
    coef[c] = 0.f; coef[C + c] = 0.f; coef[2 * C + c] = 0.f;
    return;
  }
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
  coef[c] = (float)(s1 / M);
  coef[C + c] = (float)(s2 / M);
  coef[2 * C + c] = gamma[c] * invstd[c];
}

// master [cout][taps][cin] fp32 -> bf16 images of the channel-padded GEMM: kind 0 forward [cout_p][taps][cin_p],
// kind 1 data gradient [cin_p][taps][cout_p]; zeros outside the logical ranges
__global__ void pack_train_padded_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int taps,
                                         int cin, int cout_p, int cin_p, int kind) {
  const size_t n = (size_t)cout_p * taps * cin_p;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int co, t, ci;
    if (kind == 0) { ci = (int)(i % cin_p); t = (int)((i / cin_p) % taps); co = (int)(i / ((size_t)cin_p * taps)); }
    else { co = (int)(i % cout_p); t = (int)((i / cout_p) % taps); ci = (int)(i / ((size_t)cout_p * taps)); }
    out[i] = (co < cout && ci < cin) ? to_h16<DT>(w[((size_t)co * taps + t) * cin + ci]) : (bf16_t)0;
  }
}

// ---- stochastic depth ("row" mode, torchvision.ops.StochasticDepth): rs[img] = bernoulli(1-p) / (1-p)
__global__ void sd_rowscale_kernel(float* __restrict__ rs, int n, float p, unsigned long long seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);   // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.f / 16777216.f);
  rs[i] = u < p ? 0.f : 1.f / (1.f - p);
}

// ---- 3x3 stride-2 pad-1 stem on the NHWC4 input: raw[p][co] = sum x[p@tap][ci] * bf16(w[co][tap][ci])
__global__ __launch_bounds__(256) void stem3_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wgt,
                                                        bf16_t* __restrict__ y, int n, int h, int wd, int wstride,
                                                        int cin, int cout, int C, int ho, int wo) {
  extern __shared__ float ws[];  // [9][4][C]
  for (int i = threadIdx.x; i < 36 * C; i += 256) {
    const int co = i % C, ci = (i / C) & 3, tap = i / (4 * C);
    ws[i] = (co < cout && ci < cin) ? bf16_round(wgt[((size_t)co * 9 + tap) * cin + ci]) : 0.f;
  }
  __syncthreads();
  const int c8 = C >> 3;
  const size_t total = (size_t)n * ho * wo * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const size_t p = i / c8;
    const int ow = (int)(p % wo), oh = (int)((p / wo) % ho), img = (int)(p / ((size_t)wo * ho));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 + kh - 1;
      if (ih < 0 || ih >= h) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 + kw - 1;
        if (iw < 0 || iw >= wd) continue;
        const uint2 xv = *(const uint2*)(x + (((size_t)img * h + ih) * wstride + iw) * 4);
        const float xf[4] = {lo_f32<DT>(xv.x), hi_f32<DT>(xv.x), lo_f32<DT>(xv.y), hi_f32<DT>(xv.y)};
        const float* wt = ws + (size_t)(kh * 3 + kw) * 4 * C + cc * 8;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += xf[ci] * wt[ci * C + j];
      }
    }
    *(u32x4_t*)(y + p * C + cc * 8) = pack8(acc);
  }
}

// stem weight gradient: partial[block][co][tap][ci] over the block's pixels.  A thread owns (tap, co) items and the four
// input channels of each: per pixel one LDS float of dy and one float4 of the gathered patch feed four FMAs.
__global__ __launch_bounds__(256) void stem3_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          float* __restrict__ partials, int n, int h, int wd, int wstride,
                                                          int cin, int cout, int C, int ho, int wo, int pix_per_block) {
  constexpr int P = 64;
  extern __shared__ float sm[];  // dys[P][C], xs[P][9][4]
  float* dys = sm;
  float4* xs = (float4*)(sm + P * C);
  const int M = n * ho * wo;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(M, p0 + pix_per_block);
  const int nitem = cout * 9;
  float4 acc[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int pb = p0; pb < p1; pb += P) {
    __syncthreads();
    for (int i = threadIdx.x; i < P * C; i += 256) {
      const int pp = pb + i / C;
      dys[i] = pp < p1 ? lo_f32<DT>((unsigned)dy[(size_t)pp * C + i % C]) : 0.f;
    }
    for (int i = threadIdx.x; i < P * 9; i += 256) {
      const int pp = pb + i / 9, tap = i % 9;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pp < p1) {
        const int ow = pp % wo, oh = (pp / wo) % ho, img = pp / (wo * ho);
        const int ih = oh * 2 + tap / 3 - 1, iw = ow * 2 + tap % 3 - 1;
        if (ih >= 0 && ih < h && iw >= 0 && iw < wd) {
          const uint2 q = *(const uint2*)(x + (((size_t)img * h + ih) * wstride + iw) * 4);
          v = make_float4(lo_f32<DT>(q.x), hi_f32<DT>(q.x), lo_f32<DT>(q.y), hi_f32<DT>(q.y));
        }
      }
      xs[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int o = threadIdx.x + u * 256;
      if (o < nitem) {     // (no break: the loop must unroll for acc[] to stay in registers)
        const int co = o % cout, tap = o / cout;
        float4 a = acc[u];
#pragma unroll 8
        for (int pp = 0; pp < P; ++pp) {
          const float g = dys[pp * C + co];
          const float4 v = xs[pp * 9 + tap];
          a.x += g * v.x; a.y += g * v.y; a.z += g * v.z; a.w += g * v.w;
        }
        acc[u] = a;
      }
    }
  }
  float* out = partials + (size_t)blockIdx.x * cout * 9 * cin;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int o = threadIdx.x + u * 256;
    if (o < nitem) {
      const int co = o % cout, tap = o / cout;
      float* q = out + ((size_t)co * 9 + tap) * cin;
      q[0] = acc[u].x;
      if (cin > 1) q[1] = acc[u].y;
      if (cin > 2) q[2] = acc[u].z;
      if (cin > 3) q[3] = acc[u].w;
    }
  }
}

// ---- depthwise conv ----
// master [C_log][taps] fp32 -> tap-major [taps][C] rounded to bf16 values (zeros in the pad channels)
__global__ void dw_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int c_log, int C, int taps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= taps * C) return;
  const int c = i % C, t = i / C;
  wt[i] = c < c_log ? bf16_round(w[(size_t)c * taps + t]) : 0.f;
}

__global__ __launch_bounds__(256) void dw_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wt,
                                                     bf16_t* __restrict__ y, int n, int h, int wd, int C, int k, int stride,
                                                     int pad, int ho, int wo) {
  const int c8 = C >> 3;
  const size_t total = (size_t)n * ho * wo * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const size_t p = i / c8;
    const int ow = (int)(p % wo), oh = (int)((p / wo) % ho), img = (int)(p / ((size_t)wo * ho));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < k; ++kh) {
      const int ih = oh * stride + kh - pad;
      if (ih < 0 || ih >= h) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int iw = ow * stride + kw - pad;
        if (iw < 0 || iw >= wd) continue;
        float xv[8];
        unpack8(*(const u32x4_t*)(x + (((size_t)img * h + ih) * wd + iw) * C + cc * 8), xv);
        const float4 w0 = *(const float4*)(wt + (size_t)(kh * k + kw) * C + cc * 8);
        const float4 w1 = *(const float4*)(wt + (size_t)(kh * k + kw) * C + cc * 8 + 4);
        acc[0] += xv[0] * w0.x; acc[1] += xv[1] * w0.y; acc[2] += xv[2] * w0.z; acc[3] += xv[3] * w0.w;
        acc[4] += xv[4] * w1.x; acc[5] += xv[5] * w1.y; acc[6] += xv[6] * w1.z; acc[7] += xv[7] * w1.w;
      }
    }
    *(u32x4_t*)(y + p * C + cc * 8) = pack8(acc);
  }
}

// dx[n,ih,iw,c] (+)= sum over taps with (ih + pad - kh) divisible by the stride of dy[.., (ih+pad-kh)/s, ..] * w[c][kh][kw]
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ wt,
                                                       bf16_t* __restrict__ dx, int accumulate, int n, int h, int wd, int C,
                                                       int k, int stride, int pad, int ho, int wo) {
  const int c8 = C >> 3;
  const size_t total = (size_t)n * h * wd * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const size_t p = i / c8;
    const int iw = (int)(p % wd), ih = (int)((p / wd) % h), img = (int)(p / ((size_t)wd * h));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < k; ++kh) {
      const int a = ih + pad - kh;
      if (a < 0 || a % stride) continue;
      const int oh = a / stride;
      if (oh >= ho) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int b = iw + pad - kw;
        if (b < 0 || b % stride) continue;
        const int ow = b / stride;
        if (ow >= wo) continue;
        float gv[8];
        unpack8(*(const u32x4_t*)(dy + (((size_t)img * ho + oh) * wo + ow) * C + cc * 8), gv);
        const float4 w0 = *(const float4*)(wt + (size_t)(kh * k + kw) * C + cc * 8);
        const float4 w1 = *(const float4*)(wt + (size_t)(kh * k + kw) * C + cc * 8 + 4);
        acc[0] += gv[0] * w0.x; acc[1] += gv[1] * w0.y; acc[2] += gv[2] * w0.z; acc[3] += gv[3] * w0.w;
        acc[4] += gv[4] * w1.x; acc[5] += gv[5] * w1.y; acc[6] += gv[6] * w1.z; acc[7] += gv[7] * w1.w;
      }
    }
    bf16_t* o = dx + p * C + cc * 8;
    if (accumulate) {
      float q[8];
      unpack8(*(const u32x4_t*)o, q);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += q[j];
    }
    *(u32x4_t*)o = pack8(acc);
  }
}

// gw[c][kh][kw] = sum over output pixels of dy * x(shifted): blockIdx.y = kh; the rows-in-flight of a block are combined
// through LDS in a fixed order, one partial row [c_log][taps] per block
template <int K>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                       float* __restrict__ partials, int n, int h, int wd, int C, int c_log,
                                                       int stride, int pad, int ho, int wo, int rows_per_block) {
  extern __shared__ float sm[];   // [rif][tpr][K][8]
  const int M = n * ho * wo;
  const RowWalk w(M, C, rows_per_block);
  const int kh = blockIdx.y;
  float* out = partials + (size_t)blockIdx.x * c_log * K * K;
  for (int c0 = 0; c0 < w.c8; c0 += w.tpr) {     // uniform trip count: the barriers below are reached by every thread
    const int cc = c0 + w.lane_c;
    const bool mine = w.active && cc < w.c8;
    float acc[K][8];
#pragma unroll
    for (int q = 0; q < K; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
    if (mine)
      for (int r = w.row0 + w.lane_r; r < w.row1; r += w.rif) {
        const int ow = r % wo, oh = (r / wo) % ho, img = r / (wo * ho);
        const int ih = oh * stride + kh - pad;
        if (ih < 0 || ih >= h) continue;
        float gv[8];
        unpack8(*(const u32x4_t*)(dy + (size_t)r * C + cc * 8), gv);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int iw = ow * stride + kw - pad;
          if (iw < 0 || iw >= wd) continue;
          float xv[8];
          unpack8(*(const u32x4_t*)(x + (((size_t)img * h + ih) * wd + iw) * C + cc * 8), xv);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[kw][j] += gv[j] * xv[j];
        }
      }
    __syncthreads();
    if (w.active) {
      float* dst = sm + ((size_t)w.lane_r * w.tpr + w.lane_c) * K * 8;
#pragma unroll
      for (int q = 0; q < K; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[q * 8 + j] = acc[q][j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < w.tpr * K * 8; i += 256) {
      const int j = i & 7, q = (i >> 3) % K, lc = i / (8 * K);
      const int c = (c0 + lc) * 8 + j;
      if (c0 + lc >= w.c8 || c >= c_log) continue;
      float t = 0.f;
      for (int r = 0; r < w.rif; ++r) t += sm[((size_t)r * w.tpr + lc) * K * 8 + q * 8 + j];
      out[(size_t)c * K * K + kh * K + q] = t;
    }
  }
}

// ---- squeeze-excitation ----
// partial sums over a chunk of the HW rows of one image: part[img][chunk][C] = sum x (* y when given).  Every thread of
// the block works (8-channel chunk x row in flight); the rows in flight are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void pool_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                           float* __restrict__ part, int HW, int C, int chunks) {
  extern __shared__ float sm[];   // [rif][C]
  const int img = blockIdx.x, chunk = blockIdx.y;
  const int rows = (HW + chunks - 1) / chunks;
  const int r0 = chunk * rows, r1 = min(HW, r0 + rows);
  const int c8 = C >> 3;
  const int tpr = c8 < 256 ? c8 : 256, rif = 256 / tpr;
  const int lane_c = threadIdx.x % tpr, lane_r = threadIdx.x / tpr;
  if (lane_r < rif)
    for (int cc = lane_c; cc < c8; cc += tpr) {
      float s[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = 0.f;
      for (int r = r0 + lane_r; r < r1; r += rif) {
        const size_t o = ((size_t)img * HW + r) * C + cc * 8;
        float v[8];
        unpack8(*(const u32x4_t*)(x + o), v);
        if (y) {
          float q[8];
          unpack8(*(const u32x4_t*)(y + o), q);
#pragma unroll
          for (int j = 0; j < 8; ++j) s[j] += v[j] * q[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sm[(size_t)lane_r * C + cc * 8 + j] = s[j];
    }
  __syncthreads();
  float* o = part + ((size_t)img * chunks + chunk) * C;
  for (int i = threadIdx.x; i < C; i += 256) {
    float t = 0.f;
    for (int r = 0; r < rif; ++r) t += sm[(size_t)r * C + i];
    o[i] = t;
  }
}
__global__ void pool_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int chunks, int C,
                                   float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * C) return;
  const int img = i / C, c = i % C;
  float t = 0.f;
  for (int k = 0; k < chunks; ++k) t += part[((size_t)img * chunks + k) * C + c];
  out[i] = t * scale;
}

// gates of one image per block: u1 = W1 pooled + b1, h1 = silu(u1), gate = sigmoid(W2 h1 + b2); fp32
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ W1,
                                                          const float* __restrict__ b1, const float* __restrict__ W2,
                                                          const float* __restrict__ b2, float* __restrict__ u1,
                                                          float* __restrict__ h1, float* __restrict__ gate, int C, int Cl,
                                                          int S) {
  extern __shared__ float sm[];   // pooled[Cl], h[S]
  float* sp = sm;
  float* shh = sm + Cl;
  const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < Cl; c += 256) sp[c] = pooled[(size_t)img * C + c];
  __syncthreads();
  for (int sidx = wave; sidx < S; sidx += 4) {
    float acc = 0.f;
    for (int c = lane; c < Cl; c += 64) acc += W1[(size_t)sidx * Cl + c] * sp[c];
    for (int d = 32; d; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) {
      const float u = acc + b1[sidx];
      const float hv = u / (1.f + expf(-u));
      u1[(size_t)img * S + sidx] = u;
      h1[(size_t)img * S + sidx] = hv;
      shh[sidx] = hv;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Cl; c += 256) {
    float acc = b2[c];
    for (int sidx = 0; sidx < S; ++sidx) acc += W2[(size_t)c * S + sidx] * shh[sidx];
    gate[(size_t)img * C + c] = 1.f / (1.f + expf(-acc));
  }
}

// backward of the gate path of one image per block.  in: dgate[img][c] = sum_hw g*a; out (in place): du2 = dgate*s(1-s);
// du1[img][s] = silu'(u1) * sum_c du2[c] W2[c][s];  dpool[img][c] = sum_s du1[s] W1[s][c]
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(float* __restrict__ dgate, const float* __restrict__ gate,
                                                          const float* __restrict__ u1, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, float* __restrict__ du1,
                                                          float* __restrict__ dpool, int C, int Cl, int S) {
  extern __shared__ float sm[];   // du2[Cl], du1[S]
  float* s2 = sm;
  float* s1 = sm + Cl;
  const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < Cl; c += 256) {
    const float g = gate[(size_t)img * C + c];
    const float v = dgate[(size_t)img * C + c] * g * (1.f - g);
    dgate[(size_t)img * C + c] = v;
    s2[c] = v;
  }
  __syncthreads();
  for (int sidx = wave; sidx < S; sidx += 4) {
    float acc = 0.f;
    for (int c = lane; c < Cl; c += 64) acc += s2[c] * W2[(size_t)c * S + sidx];
    for (int d = 32; d; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) {
      const float z = u1[(size_t)img * S + sidx], sg = 1.f / (1.f + expf(-z));
      const float v = acc * sg * (1.f + z * (1.f - sg));
      du1[(size_t)img * S + sidx] = v;
      s1[sidx] = v;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Cl; c += 256) {
    float acc = 0.f;
    for (int sidx = 0; sidx < S; ++sidx) acc += s1[sidx] * W1[(size_t)sidx * Cl + c];
    dpool[(size_t)img * C + c] = acc;
  }
}

// parameter gradients of the two 1x1 convs of the gate path (sums over the batch, fixed order); null: not wanted
__global__ void se_wgrad_kernel(const float* __restrict__ du2, const float* __restrict__ h1, const float* __restrict__ du1,
                                const float* __restrict__ pooled, float* __restrict__ gW1, float* __restrict__ gb1,
                                float* __restrict__ gW2, float* __restrict__ gb2, int n, int C, int Cl, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cl * S) return;
  if (gW2) {   // [Cl][S]
    const int c = i / S, sidx = i % S;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < n; ++k) acc += du2[(size_t)k * C + c] * h1[(size_t)k * S + sidx];
    gW2[i] = acc;
  }
  if (gW1) {   // [S][Cl]
    const int sidx = i / Cl, c = i % Cl;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < n; ++k) acc += du1[(size_t)k * S + sidx] * pooled[(size_t)k * C + c];
    gW1[i] = acc;
  }
  if (gb2 && i < Cl) {
    float acc = 0.f;
    for (int k = 0; k < n; ++k) acc += du2[(size_t)k * C + i];
    gb2[i] = acc;
  }
  if (gb1 && i < S) {
    float acc = 0.f;
    for (int k = 0; k < n; ++k) acc += du1[(size_t)k * S + i];
    gb1[i] = acc;
  }
}

// out = a * gate[img][c]
__global__ __launch_bounds__(256) void se_scale_kernel(const bf16_t* __restrict__ a, const float* __restrict__ gate,
                                                       bf16_t* __restrict__ out, int n, int HW, int C) {
  const int c8 = C >> 3;
  const size_t total = (size_t)n * HW * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const int img = (int)(i / ((size_t)HW * c8));
    float v[8];
    unpack8(*(const u32x4_t*)(a + i * 8), v);
    const float* gt = gate + (size_t)img * C + cc * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= gt[j];
    *(u32x4_t*)(out + i * 8) = pack8(v);
  }
}
// da = g * gate[img][c] + dpool[img][c] * inv_hw
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const bf16_t* __restrict__ g, const float* __restrict__ gate,
                                                           const float* __restrict__ dpool, bf16_t* __restrict__ da, int n,
                                                           int HW, int C, float inv_hw) {
  const int c8 = C >> 3;
  const size_t total = (size_t)n * HW * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const int img = (int)(i / ((size_t)HW * c8));
    float v[8];
    unpack8(*(const u32x4_t*)(g + i * 8), v);
    const float* gt = gate + (size_t)img * C + cc * 8;
    const float* dp = dpool + (size_t)img * C + cc * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * gt[j] + dp[j] * inv_hw;
    *(u32x4_t*)(da + i * 8) = pack8(v);
  }
}

// small element-wise maps on the [n][squeeze] / [n][C] vectors
constexpr int EW_SILU = SPK_EW_SILU, EW_SIGMOID = SPK_EW_SIGMOID, EW_SILU_BWD = SPK_EW_SILU_BWD;
__global__ void ew_kernel(int mode, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                          size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i];
  float r;
  if (mode == EW_SILU) r = x / (1.f + expf(-x));
  else if (mode == EW_SIGMOID) r = 1.f / (1.f + expf(-x));
  else if (mode == EW_SILU_BWD) {                      // a: gradient, b: pre-activation
    const float z = b[i], s = 1.f / (1.f + expf(-z));
    r = x * s * (1.f + z * (1.f - s));
  } else {                                             // a: gradient, b: sigmoid output
    const float s = b[i];
    r = x * s * (1.f - s);
  }
  out[i] = r;
}
// db[j] = sum_i dy[i*stride + j]
__global__ void colsum_strided_kernel(const float* __restrict__ dy, float* __restrict__ db, int n, int c, int stride) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= c) return;
  float t = 0.f;
  for (int i = 0; i < n; ++i) t += dy[(size_t)i * stride + j];
  db[j] = t;
}

// wgrad slabs of a channel-padded GEMM [splits][cout_p][taps*cin_p] -> gradient [cout][taps][cin], fixed order
__global__ void slab_reduce_sub_kernel(const float* __restrict__ slabs, float* __restrict__ out, int cout, int taps,
                                       int cin, int cout_p, int cin_p, int splits) {
  const size_t n = (size_t)cout * taps * cin;
  const size_t slab = (size_t)cout_p * taps * cin_p;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin), t = (int)((i / cin) % taps), co = (int)(i / ((size_t)cin * taps));
    const size_t src = ((size_t)co * taps + t) * cin_p + ci;
    float acc = 0.f;
    for (int k = 0; k < splits; ++k) acc += slabs[(size_t)k * slab + src];
    out[i] = acc;
  }
}

inline int grid_of(size_t total, int block) {
  const size_t g = (total + block - 1) / block;
  return (int)std::min<size_t>(g, 65535 * 16);
}
inline int walk_rows(int M, int* rows_per_block) {
  int rpb = 1024;
  while (rpb > 64 && (M + rpb - 1) / rpb < 1024) rpb >>= 1;
  *rows_per_block = rpb;
  return (M + rpb - 1) / rpb;
}
inline size_t walk_lds(int C) {
  const int c8 = C / 8, tpr = c8 < 256 ? c8 : 256;
  return (size_t)(256 / tpr) * 2 * C * sizeof(float);
}
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -1)

}  // namespace

int spk_eff_stat_blocks(int M, int* rows_per_block) { return walk_rows(M, rows_per_block); }

int spk_launch_col_stats(const bf16_t* x, float* partials, int M, int C, int* blocks, hipStream_t s) {
  int rpb;
  const int nb = walk_rows(M, &rpb);
  hipLaunchKernelGGL(col_stats_kernel, dim3(nb), dim3(256), walk_lds(C), s, x, partials, M, C, rpb);
  *blocks = nb;
  return LAUNCH_OK();
}

int spk_launch_bna_apply(const bf16_t* raw, const float* scale, const float* shift, const bf16_t* res,
                         const float* rowscale, bf16_t* out, int M, int C, int HW, int act, hipStream_t s) {
  int rpb;
  const int nb = walk_rows(M, &rpb);
  hipLaunchKernelGGL(bna_apply_kernel, dim3(nb), dim3(256), 0, s, raw, scale, shift, res, rowscale, out, M, C, HW, act,
                     rpb);
  return LAUNCH_OK();
}

int spk_launch_bna_bwd_reduce(const bf16_t* g, const bf16_t* raw, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const float* rowscale, float* partials, int M,
                              int C, int HW, int act, int* blocks, hipStream_t s) {
  int rpb;
  const int nb = walk_rows(M, &rpb);
  hipLaunchKernelGGL(bna_bwd_reduce_kernel, dim3(nb), dim3(256), walk_lds(C), s, g, raw, scale, shift, mean, invstd,
                     rowscale, partials, M, C, HW, act, rpb);
  *blocks = nb;
  return LAUNCH_OK();
}

int spk_launch_bna_bwd_apply(const bf16_t* g, const bf16_t* raw, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* coef, const float* rowscale,
                             bf16_t* dy, bf16_t* g_res, int res_accumulate, int M, int C, int HW, int act,
                             hipStream_t s) {
  int rpb;
  const int nb = walk_rows(M, &rpb);
  hipLaunchKernelGGL(bna_bwd_apply_kernel, dim3(nb), dim3(256), 0, s, g, raw, scale, shift, mean, invstd, coef, rowscale,
                     dy, g_res, res_accumulate, M, C, HW, act, rpb);
  return LAUNCH_OK();
}

int spk_launch_sd_rowscale(float* rs, int n, float p, unsigned long long seed, hipStream_t s) {
  hipLaunchKernelGGL(sd_rowscale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rs, n, p, seed);
  return LAUNCH_OK();
}

int spk_launch_stem3_train_fwd(const bf16_t* x, const float* w, bf16_t* y, int n, int h, int wd, int wstride, int cin,
                               int cout, int C, int ho, int wo, hipStream_t s) {
  const size_t total = (size_t)n * ho * wo * (C / 8);
  hipLaunchKernelGGL(stem3_fwd_kernel, dim3(grid_of(total, 256)), dim3(256), (size_t)36 * C * 4, s, x, w, y, n, h, wd,
                     wstride, cin, cout, C, ho, wo);
  return LAUNCH_OK();
}

int spk_stem3_wgrad_blocks(int M, int* pix_per_block) {
  int ppb = 4096;
  while (ppb > 256 && (M + ppb - 1) / ppb < 512) ppb >>= 1;
  *pix_per_block = ppb;
  return (M + ppb - 1) / ppb;
}

int spk_launch_stem3_wgrad(const bf16_t* x, const bf16_t* dy, float* partials, int n, int h, int wd, int wstride,
                           int cin, int cout, int C, int ho, int wo, int* blocks, hipStream_t s) {
  if (cout * 9 > 768 || cin > 4) return -1;
  int ppb;
  const int nb = spk_stem3_wgrad_blocks(n * ho * wo, &ppb);
  hipLaunchKernelGGL(stem3_wgrad_kernel, dim3(nb), dim3(256), (size_t)(64 * C + 64 * 36) * 4, s, x, dy, partials, n, h,
                     wd, wstride, cin, cout, C, ho, wo, ppb);
  *blocks = nb;
  return LAUNCH_OK();
}

int spk_launch_dw_pack(const float* w, float* wt, int c_log, int C, int taps, hipStream_t s) {
  hipLaunchKernelGGL(dw_pack_kernel, dim3((taps * C + 255) / 256), dim3(256), 0, s, w, wt, c_log, C, taps);
  return LAUNCH_OK();
}

int spk_launch_dw_train_fwd(const bf16_t* x, const float* wt, bf16_t* y, int n, int h, int wd, int C, int k, int stride,
                            int pad, int ho, int wo, hipStream_t s) {
  const size_t total = (size_t)n * ho * wo * (C / 8);
  hipLaunchKernelGGL(dw_fwd_kernel, dim3(grid_of(total, 256)), dim3(256), 0, s, x, wt, y, n, h, wd, C, k, stride, pad, ho,
                     wo);
  return LAUNCH_OK();
}

int spk_launch_dw_dgrad(const bf16_t* dy, const float* wt, bf16_t* dx, int accumulate, int n, int h, int wd, int C, int k,
                        int stride, int pad, int ho, int wo, hipStream_t s) {
  const size_t total = (size_t)n * h * wd * (C / 8);
  hipLaunchKernelGGL(dw_dgrad_kernel, dim3(grid_of(total, 256)), dim3(256), 0, s, dy, wt, dx, accumulate, n, h, wd, C, k,
                     stride, pad, ho, wo);
  return LAUNCH_OK();
}

// partial rows the weight-gradient kernel writes for an [M][C] problem ([rows][c_log][k*k] floats): one per block
static int dw_wgrad_blocks(int M, int* rows_per_block) {
  int rpb = 8192;
  while (rpb > 256 && (M + rpb - 1) / rpb < 256) rpb >>= 1;
  *rows_per_block = rpb;
  return (M + rpb - 1) / rpb;
}
int spk_dw_wgrad_rows(int M, int C) {
  int rpb;
  return dw_wgrad_blocks(M, &rpb);
}

int spk_launch_dw_wgrad(const bf16_t* x, const bf16_t* dy, float* partials, int n, int h, int wd, int C, int c_log,
                        int k, int stride, int pad, int ho, int wo, int* rows, hipStream_t s) {
  int rpb;
  const int M = n * ho * wo;
  const int nb = dw_wgrad_blocks(M, &rpb);
  *rows = nb;
  const int c8 = C / 8, tpr = c8 < 256 ? c8 : 256;
  const size_t lds = (size_t)(256 / tpr) * tpr * k * 8 * sizeof(float);
  if (k == 3)
    hipLaunchKernelGGL(dw_wgrad_kernel<3>, dim3(nb, 3), dim3(256), lds, s, x, dy, partials, n, h, wd, C, c_log, stride, pad,
                       ho, wo, rpb);
  else if (k == 5)
    hipLaunchKernelGGL(dw_wgrad_kernel<5>, dim3(nb, 5), dim3(256), lds, s, x, dy, partials, n, h, wd, C, c_log, stride, pad,
                       ho, wo, rpb);
  else
    return -1;
  return LAUNCH_OK();
}

int spk_se_chunks(int HW) { return HW >= 3136 ? 16 : (HW >= 196 ? 4 : 1); }

// out[n][C] = scale * sum over HW of x (* y)
int spk_launch_pool_rows(const bf16_t* x, const bf16_t* y, float* part, float* out, int n, int HW, int C, float scale,
                         hipStream_t s) {
  const int chunks = spk_se_chunks(HW);
  hipLaunchKernelGGL(pool_partial_kernel, dim3(n, chunks), dim3(256), walk_lds(C) / 2, s, x, y, part, HW, C, chunks);
  hipLaunchKernelGGL(pool_finish_kernel, dim3((n * C + 255) / 256), dim3(256), 0, s, part, out, n, chunks, C, scale);
  return LAUNCH_OK();
}

int spk_launch_se_scale(const bf16_t* a, const float* gate, bf16_t* out, int n, int HW, int C, hipStream_t s) {
  hipLaunchKernelGGL(se_scale_kernel, dim3(grid_of((size_t)n * HW * (C / 8), 256)), dim3(256), 0, s, a, gate, out, n, HW,
                     C);
  return LAUNCH_OK();
}

int spk_launch_se_bwd_apply(const bf16_t* g, const float* gate, const float* dpool, bf16_t* da, int n, int HW, int C,
                            hipStream_t s) {
  hipLaunchKernelGGL(se_bwd_apply_kernel, dim3(grid_of((size_t)n * HW * (C / 8), 256)), dim3(256), 0, s, g, gate, dpool,
                     da, n, HW, C, 1.f / (float)HW);
  return LAUNCH_OK();
}

int spk_launch_ew(int mode, const float* a, const float* b, float* out, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(ew_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mode, a, b, out, n);
  return LAUNCH_OK();
}

int spk_launch_colsum_strided(const float* dy, float* db, int n, int c, int stride, hipStream_t s) {
  hipLaunchKernelGGL(colsum_strided_kernel, dim3((c + 63) / 64), dim3(64), 0, s, dy, db, n, c, stride);
  return LAUNCH_OK();
}

int spk_launch_slab_reduce_sub(const float* slabs, float* out, int cout, int taps, int cin, int cout_p, int cin_p,
                               int splits, hipStream_t s) {
  hipLaunchKernelGGL(slab_reduce_sub_kernel, dim3(grid_of((size_t)cout * taps * cin, 256)), dim3(256), 0, s, slabs, out,
                     cout, taps, cin, cout_p, cin_p, splits);
  return LAUNCH_OK();
}

static const float* bna_presum(const float* partials, int* count, int C, float* tmp, hipStream_t s) {
  if (*count <= 128 || !tmp) return partials;
  const int slices = 64;
  const int rps = (*count + slices - 1) / slices;
  hipLaunchKernelGGL(bna_presum_kernel, dim3((C + 63) / 64, slices), dim3(1024), 0, s, partials, *count, C, rps, tmp);
  *count = slices;
  return tmp;
}

int spk_launch_bna_finalize(const float* partials, int count, int C, int c_log, double M, const float* gamma,
                            const float* beta, float* rmean, float* rvar, float* st, float eps, float momentum,
                            float* tmp, hipStream_t s) {
  partials = bna_presum(partials, &count, C, tmp, s);
  hipLaunchKernelGGL(bna_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, partials, count, C, c_log, M, gamma,
                     beta, rmean, rvar, st, eps, momentum);
  return LAUNCH_OK();
}

int spk_launch_bna_bwd_finalize(const float* partials, int count, int C, int c_log, double M, const float* gamma,
                                const float* invstd, float* dgamma, float* dbeta, float* coef, float* tmp, hipStream_t s) {
  partials = bna_presum(partials, &count, C, tmp, s);
  hipLaunchKernelGGL(bna_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, partials, count, C, c_log, M, gamma,
                     invstd, dgamma, dbeta, coef);
  return LAUNCH_OK();
}

int spk_launch_pack_train_padded(const float* w, bf16_t* out, int cout, int taps, int cin, int cout_p, int cin_p,
                                 int kind, hipStream_t s) {
  hipLaunchKernelGGL(pack_train_padded_kernel, dim3(grid_of((size_t)cout_p * taps * cin_p, 256)), dim3(256), 0, s, w, out,
                     cout, taps, cin, cout_p, cin_p, kind);
  return LAUNCH_OK();
}

int spk_launch_se_gate_fwd(const float* pooled, const float* W1, const float* b1, const float* W2, const float* b2,
                           float* u1, float* h1, float* gate, int n, int C, int Cl, int S, hipStream_t s) {
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(n), dim3(256), (size_t)(Cl + S) * 4, s, pooled, W1, b1, W2, b2, u1, h1, gate,
                     C, Cl, S);
  return LAUNCH_OK();
}

int spk_launch_se_gate_bwd(float* dgate, const float* gate, const float* u1, const float* W1, const float* W2, float* du1,
                           float* dpool, int n, int C, int Cl, int S, hipStream_t s) {
  hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(n), dim3(256), (size_t)(Cl + S) * 4, s, dgate, gate, u1, W1, W2, du1, dpool,
                     C, Cl, S);
  return LAUNCH_OK();
}

int spk_launch_se_wgrad(const float* du2, const float* h1, const float* du1, const float* pooled, float* gW1, float* gb1,
                        float* gW2, float* gb2, int n, int C, int Cl, int S, hipStream_t s) {
  if (!gW1 && !gb1 && !gW2 && !gb2) return 0;
  hipLaunchKernelGGL(se_wgrad_kernel, dim3((Cl * S + 255) / 256), dim3(256), 0, s, du2, h1, du1, pooled, gW1, gb1, gW2, gb2,
                     n, C, Cl, S);
  return LAUNCH_OK();
}
