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
#include "ordered_reduce.h"

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
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas, v_div_fixup per
// element): the BatchNorm + SiLU passes are as much VALU- as HBM-bound
__device__ __forceinline__ float sigmoidf_(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
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

// two channels at a time: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue two lanes' worth of fp32 work per instruction,
// and the BatchNorm + SiLU passes are as much VALU- as HBM-bound
typedef __attribute__((ext_vector_type(2))) float f2_t;
__device__ __forceinline__ f2_t fma2(f2_t a, f2_t b, f2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2_t act_grad2(f2_t z, int act) {
  if (act == SPK_ACT_SILU) {
    const f2_t e = {__expf(-z[0]), __expf(-z[1])};
    const f2_t d = e + 1.f;
    const f2_t s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return s * fma2(z, 1.f - s, (f2_t)1.f);
  }
  if (act == SPK_ACT_RELU) return f2_t{z[0] > 0.f ? 1.f : 0.f, z[1] > 0.f ? 1.f : 0.f};
  return (f2_t)1.f;
}

// A block owns rows [row0, row1) x the 8-channel groups [cb, ce) of channel tile `ct` of `cts`; a thread owns groups
// cb + lane_c, cb + lane_c + tpr, ... and walks rows lane_r, lane_r + rif, ...  Wide tensors on few rows (the 7x7 and
// 14x14 layers: 1152 channels x 6272 rows) are tiled along the channels: without tiles one thread per group walked all
// the rows of its block one after the other, rif = 1, on 98 blocks - 40-60 us per pass whatever the tensor size.
struct RowWalk {
  int c8, tpr, rif, lane_c, lane_r, row0, row1, cb, ce;
  bool active;
  __device__ RowWalk(int M, int C, int rows_per_block, int ct = 0, int cts = 1) {
    c8 = C >> 3;
    const int tile = (c8 + cts - 1) / cts;
    cb = ct * tile;
    ce = min(c8, cb + tile);
    tpr = tile < 256 ? tile : 256;
    rif = 256 / tpr;
    lane_c = threadIdx.x % tpr;
    lane_r = threadIdx.x / tpr;
    active = lane_r < rif;
    row0 = blockIdx.x * rows_per_block;
    row1 = min(M, row0 + rows_per_block);
  }
  __device__ int tw() const { return ((c8 + (int)gridDim.y - 1) / (int)gridDim.y) * 8; }   // channels of a tile (grid.y tiles)
};

// rows in flight of a block [rif][2][TW] -> partials[block][2][C] (the tile's channels), fixed order
__device__ __forceinline__ void walk_combine(const float* sm, float* __restrict__ partials, const RowWalk& w, int TW,
                                             int C) {
  const int nch = (w.ce - w.cb) * 8;
  for (int i = threadIdx.x; i < 2 * nch; i += 256) {
    const int which = i >= nch, col = i - which * nch;
    float t = 0.f;
    for (int r = 0; r < w.rif; ++r) t += sm[(r * 2 + which) * TW + col];
    partials[((size_t)blockIdx.x * 2 + which) * C + w.cb * 8 + col] = t;
  }
}

// ---- per-channel sum / sum of squares of a bf16 [M][C] tensor: partials[block][2][C] (the layout bn_finalize reads)
__global__ __launch_bounds__(256) void col_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ partials,
                                                        int M, int C, int rows_per_block) {
  extern __shared__ float sm[];  // [rif][2][TW]
  const RowWalk w(M, C, rows_per_block, blockIdx.y, gridDim.y);
  const int TW = w.tw();
  if (w.active)
    for (int cc = w.cb + w.lane_c; cc < w.ce; cc += w.tpr) {
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
        sm[(w.lane_r * 2 + 0) * TW + (cc - w.cb) * 8 + j] = s1[j];
        sm[(w.lane_r * 2 + 1) * TW + (cc - w.cb) * 8 + j] = s2[j];
      }
    }
  __syncthreads();
  walk_combine(sm, partials, w, TW, C);
}

// ---- BatchNorm apply + activation (+ per-image factor) (+ shortcut): a = act(raw*scale + shift) * rs[img] + res
__global__ __launch_bounds__(256) void bna_apply_kernel(const bf16_t* __restrict__ raw, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const bf16_t* __restrict__ res,
                                                        const float* __restrict__ rowscale, bf16_t* __restrict__ out,
                                                        int M, int C, int HW, int act, int rows_per_block) {
  const RowWalk w(M, C, rows_per_block, blockIdx.y, gridDim.y);
  if (!w.active) return;
  for (int cc = w.cb + w.lane_c; cc < w.ce; cc += w.tpr) {
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
  extern __shared__ float sm[];  // [rif][2][TW]
  const RowWalk w(M, C, rows_per_block, blockIdx.y, gridDim.y);
  const int TW = w.tw();
  if (w.active)
    for (int cc = w.cb + w.lane_c; cc < w.ce; cc += w.tpr) {
      f2_t s1[4], s2[4], sc[4], sh[4], mu[4], is[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = cc * 8 + 2 * j;
        s1[j] = s2[j] = (f2_t)0.f;
        sc[j] = f2_t{scale[c], scale[c + 1]}; sh[j] = f2_t{shift[c], shift[c + 1]};
        mu[j] = f2_t{mean[c], mean[c + 1]}; is[j] = f2_t{invstd[c], invstd[c + 1]};
      }
      for (int r = w.row0 + w.lane_r; r < w.row1; r += w.rif) {
        const size_t o = (size_t)r * C + cc * 8;
        float gv[8], yv[8];
        unpack8(*(const u32x4_t*)(g + o), gv);
        unpack8(*(const u32x4_t*)(raw + o), yv);
        const float rs = rowscale ? rowscale[r / HW] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f2_t y2 = {yv[2 * j], yv[2 * j + 1]}, g2 = {gv[2 * j], gv[2 * j + 1]};
          const f2_t dz = g2 * rs * act_grad2(fma2(y2, sc[j], sh[j]), act);
          s1[j] += dz;
          s2[j] = fma2(dz, (y2 - mu[j]) * is[j], s2[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[(w.lane_r * 2 + 0) * TW + (cc - w.cb) * 8 + j] = s1[j >> 1][j & 1];
        sm[(w.lane_r * 2 + 1) * TW + (cc - w.cb) * 8 + j] = s2[j >> 1][j & 1];
      }
    }
  __syncthreads();
  walk_combine(sm, partials, w, TW, C);
}

// stage 3: dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); the shortcut receives g itself
__global__ __launch_bounds__(256) void bna_bwd_apply_kernel(
    const bf16_t* __restrict__ g, const bf16_t* __restrict__ raw, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ coef, const float* __restrict__ rowscale, bf16_t* __restrict__ dy,
    bf16_t* __restrict__ g_res, int res_accumulate, int M, int C, int HW, int act, int rows_per_block) {
  const RowWalk w(M, C, rows_per_block, blockIdx.y, gridDim.y);
  if (!w.active) return;
  for (int cc = w.cb + w.lane_c; cc < w.ce; cc += w.tpr) {
    f2_t sc[4], sh[4], mu[4], is[4], k0[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cc * 8 + 2 * j;
      sc[j] = f2_t{scale[c], scale[c + 1]}; sh[j] = f2_t{shift[c], shift[c + 1]};
      mu[j] = f2_t{mean[c], mean[c + 1]}; is[j] = f2_t{invstd[c], invstd[c + 1]};
      k0[j] = f2_t{coef[c], coef[c + 1]}; k1[j] = f2_t{coef[C + c], coef[C + c + 1]};
      k2[j] = f2_t{coef[2 * C + c], coef[2 * C + c + 1]};
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
      for (int j = 0; j < 4; ++j) {
        const f2_t y2 = {yv[2 * j], yv[2 * j + 1]}, g2 = {gv[2 * j], gv[2 * j + 1]};
        const f2_t dz = g2 * rs * act_grad2(fma2(y2, sc[j], sh[j]), act);
        const f2_t xh = (y2 - mu[j]) * is[j];
        const f2_t o2 = k2[j] * (dz - k0[j] - xh * k1[j]);
        ov[2 * j] = o2[0];
        ov[2 * j + 1] = o2[1];
      }
      *(u32x4_t*)(dy + o) = pack8(ov);
    }
  }
}

// ---- ordered finalize steps for a channel-padded tensor: parameters and running statistics exist for c < c_log only;
// pad channels get mean 0, invstd 0, scale 0, shift 0 (their raw values are zeros) ----
struct BnaFwdFin {
  int C, c_log;
  double M;
  const float *gamma, *beta;
  float *rmean, *rvar, *st;
  float eps, momentum;
  __device__ __forceinline__ void operator()(int c, double s1, double s2) const {
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
};

struct BnaBwdFin {
  int C, c_log;
  double M;
  const float *gamma, *invstd;
  float *dgamma, *dbeta, *coef;
  __device__ __forceinline__ void operator()(int c, double s1, double s2) const {
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
};

// One launch for up to 48 weight images of the step (blockIdx.y = table entry).  kind 0 / 1: master [cout][taps][cin] fp32 ->
// bf16 image of the channel-padded GEMM, forward [cout_p][taps][cin_p] / data gradient [cin_p][taps][cout_p], zeros outside
// the logical ranges; kind 2: depthwise master [C_log][taps] -> tap-major [taps][C] floats rounded to bf16 values (zeros in
// the pad channels) and, behind it, the same with the window flipped (the stride-1 data gradient is a depthwise conv of dy
// with it)
__global__ void pack_padded_multi_kernel(const float* __restrict__ pbuf, bf16_t* __restrict__ wpack, float* __restrict__ dwt,
                                         PadPackTable t) {
  const PadPackEntry e = t.e[blockIdx.y];
  const float* w = pbuf + e.src;
  const int cout = (int)e.cout, taps = (int)e.taps, cin = (int)e.cin, cout_p = (int)e.cout_p, cin_p = (int)e.cin_p;
  if (e.kind == 2) {   // dw_pack_kernel
    float* wt = dwt + e.dst;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < taps * cout_p; i += gridDim.x * blockDim.x) {
      const int c = i % cout_p, tp = i / cout_p;
      const float v = c < cout ? bf16_round(w[(size_t)c * taps + tp]) : 0.f;
      wt[i] = v;
      wt[(size_t)(2 * taps - 1 - tp) * cout_p + c] = v;
    }
    return;
  }
  bf16_t* out = wpack + e.dst;   // pack_train_padded_kernel
  const size_t n = (size_t)cout_p * taps * cin_p;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int co, tp, ci;
    if (e.kind == 0) { ci = (int)(i % cin_p); tp = (int)((i / cin_p) % taps); co = (int)(i / ((size_t)cin_p * taps)); }
    else { co = (int)(i % cout_p); tp = (int)((i / cout_p) % taps); ci = (int)(i / ((size_t)cout_p * taps)); }
    out[i] = (co < cout && ci < cin) ? to_h16<DT>(w[((size_t)co * taps + tp) * cin + ci]) : (bf16_t)0;
  }
}

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
// A thread owns 8 output channels of STEM_PX horizontally adjacent pixels: a weight read from LDS serves all of them and
// the 2*STEM_PX+1 input columns of a row are loaded once (one pixel per thread re-read the 36 x 8 weights per pixel and
// was bound by LDS reads: 412 us for 128 x 112 x 112 x 64).  Channel groups that are padding only store zeros.
constexpr int STEM_PX = 4;
__global__ __launch_bounds__(256) void stem3_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wgt,
                                                        bf16_t* __restrict__ y, int n, int h, int wd, int wstride,
                                                        int cin, int cout, int C, int ho, int wo, float out_scale) {
  extern __shared__ float ws[];  // [9][4][C]
  for (int i = threadIdx.x; i < 36 * C; i += 256) {
    const int co = i % C, ci = (i / C) & 3, tap = i / (4 * C);
    ws[i] = (co < cout && ci < cin) ? bf16_round(wgt[((size_t)co * 9 + tap) * cin + ci]) : 0.f;
  }
  __syncthreads();
  constexpr int COLS = 2 * STEM_PX + 1;
  const int c8 = C >> 3, c8l = (cout + 7) >> 3;
  const int gpr = (wo + STEM_PX - 1) / STEM_PX;
  const size_t total = (size_t)n * ho * gpr * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const size_t g = i / c8;
    const int ow0 = (int)(g % gpr) * STEM_PX, oh = (int)((g / gpr) % ho), img = (int)(g / ((size_t)gpr * ho));
    float acc[STEM_PX][8];
#pragma unroll
    for (int u = 0; u < STEM_PX; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[u][j] = 0.f;
    if (cc < c8l)
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 + kh - 1;
        if (ih < 0 || ih >= h) continue;
        uint2 raw[COLS];
#pragma unroll
        for (int col = 0; col < COLS; ++col) {
          const int iw = ow0 * 2 - 1 + col;
          const int iwc = min(max(iw, 0), wd - 1);
          raw[col] = *(const uint2*)(x + (((size_t)img * h + ih) * wstride + iwc) * 4);
          if ((unsigned)iw >= (unsigned)wd) raw[col] = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float* wt = ws + (size_t)(kh * 3 + kw) * 4 * C + cc * 8;
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) {
            const float4 w0 = *(const float4*)(wt + ci * C), w1 = *(const float4*)(wt + ci * C + 4);
#pragma unroll
            for (int u = 0; u < STEM_PX; ++u) {
              const uint2 q = raw[2 * u + kw];
              const float xv = ci == 0 ? lo_f32<DT>(q.x) : (ci == 1 ? hi_f32<DT>(q.x) : lo_f32<DT>(q.y));
              acc[u][0] += xv * w0.x; acc[u][1] += xv * w0.y; acc[u][2] += xv * w0.z; acc[u][3] += xv * w0.w;
              acc[u][4] += xv * w1.x; acc[u][5] += xv * w1.y; acc[u][6] += xv * w1.z; acc[u][7] += xv * w1.w;
            }
          }
          if (cin > 3) {
            const float4 w0 = *(const float4*)(wt + 3 * C), w1 = *(const float4*)(wt + 3 * C + 4);
#pragma unroll
            for (int u = 0; u < STEM_PX; ++u) {
              const float xv = hi_f32<DT>(raw[2 * u + kw].y);
              acc[u][0] += xv * w0.x; acc[u][1] += xv * w0.y; acc[u][2] += xv * w0.z; acc[u][3] += xv * w0.w;
              acc[u][4] += xv * w1.x; acc[u][5] += xv * w1.y; acc[u][6] += xv * w1.z; acc[u][7] += xv * w1.w;
            }
          }
        }
      }
#pragma unroll
    for (int u = 0; u < STEM_PX; ++u)
      if (ow0 + u < wo) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[u][j] *= out_scale;   // (the input holds pixel values x 255: exact products, one scale)
        *(u32x4_t*)(y + ((((size_t)img * ho + oh) * wo) + ow0 + u) * C + cc * 8) = pack8(acc[u]);
      }
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

// The same sums with a register tile: a thread owns 4 output channels x 1 tap (x the 4 input channels) of every G-th
// pixel of the staged tile - one float4 of dy and one float4 of the patch from LDS feed 16 FMAs (the kernel above: 4 FMAs
// per 20 bytes of LDS reads, and 288 items on 3 x 256 thread slots: 470 us for 128 x 112 x 112 x 32, all of it exposed at the
// end of the step behind the last BatchNorm backward).  items = 9 * cout/4 <= 256, G = 256 / items pixel groups, combined
// through LDS in a fixed order.
__global__ __launch_bounds__(256) void stem3_wgrad_rt_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             float* __restrict__ partials, int n, int h, int wd,
                                                             int wstride, int cin, int cout, int C, int ho, int wo,
                                                             int pix_per_block) {
  constexpr int P = 64;
  extern __shared__ float sm[];  // dys[P][cout], xs[P][9][4]; then [G][items][16]
  float* dys = sm;
  float4* xs = (float4*)(sm + P * cout);
  const int M = n * ho * wo;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(M, p0 + pix_per_block);
  const int cg = cout >> 2, items = 9 * cg, G = 256 / items;
  const int item = threadIdx.x % items, pg = threadIdx.x / items;
  const int tap = item / cg, cog = item - tap * cg;
  const bool mine = pg < G;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int pb = p0; pb < p1; pb += P) {
    __syncthreads();
    for (int i = threadIdx.x; i < P * cout; i += 256) {
      const int pp = pb + i / cout;
      dys[i] = pp < p1 ? lo_f32<DT>((unsigned)dy[(size_t)pp * C + i % cout]) : 0.f;
    }
    for (int i = threadIdx.x; i < P * 9; i += 256) {
      const int pp = pb + i / 9, tp = i % 9;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pp < p1) {
        const int ow = pp % wo, oh = (pp / wo) % ho, img = pp / (wo * ho);
        const int ih = oh * 2 + tp / 3 - 1, iw = ow * 2 + tp % 3 - 1;
        if (ih >= 0 && ih < h && iw >= 0 && iw < wd) {
          const uint2 q = *(const uint2*)(x + (((size_t)img * h + ih) * wstride + iw) * 4);
          v = make_float4(lo_f32<DT>(q.x), hi_f32<DT>(q.x), lo_f32<DT>(q.y), hi_f32<DT>(q.y));
        }
      }
      xs[i] = v;
    }
    __syncthreads();
    if (mine)
#pragma unroll 4
      for (int pp = pg; pp < P; pp += G) {
        const float4 g = *(const float4*)(dys + pp * cout + cog * 4);
        const float4 v = xs[pp * 9 + tap];
        const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          acc[a][0] += gg[a] * v.x; acc[a][1] += gg[a] * v.y; acc[a][2] += gg[a] * v.z; acc[a][3] += gg[a] * v.w;
        }
      }
  }
  __syncthreads();
  if (mine)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) sm[((size_t)pg * items + item) * 16 + a * 4 + b] = acc[a][b];
  __syncthreads();
  float* out = partials + (size_t)blockIdx.x * cout * 9 * cin;
  for (int i = threadIdx.x; i < items * 16; i += 256) {
    const int it = i >> 4, a = (i >> 2) & 3, b = i & 3;
    if (b >= cin) continue;
    float t = 0.f;
    for (int g = 0; g < G; ++g) t += sm[((size_t)g * items + it) * 16 + a * 4 + b];
    const int tp = it / cg, co = (it - tp * cg) * 4 + a;
    out[((size_t)co * 9 + tp) * cin + b] = t;
  }
}

// ---- depthwise conv ----
// (the tap-major depthwise windows [taps][C], rounded to bf16 values with zeros in the pad channels, and their flipped
// copies come from pack_padded_multi_kernel)

// Loads are unconditional (coordinates clamped into the image, the value zeroed by a select): the K loads of a window row
// are in flight together.  With a branch around every tap each load was waited for before the next was issued.
template <int K>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wt,
                                                     bf16_t* __restrict__ y, int n, int h, int wd, int C, int stride,
                                                     int pad, int ho, int wo) {
  const int c8 = C >> 3;
  const size_t total = (size_t)n * ho * wo * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const size_t p = i / c8;
    const int ow = (int)(p % wo), oh = (int)((p / wo) % ho), img = (int)(p / ((size_t)wo * ho));
    const bf16_t* xi = x + (size_t)img * h * wd * C + cc * 8;
    const float* wc = wt + cc * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll(K == 3 ? 3 : 1)
    for (int kh = 0; kh < K; ++kh) {
      const int ih = oh * stride + kh - pad;
      const bool rowok = (unsigned)ih < (unsigned)h;
      const int ihc = min(max(ih, 0), h - 1);
      u32x4_t raw[K];
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int iw = ow * stride + kw - pad;
        const int iwc = min(max(iw, 0), wd - 1);
        raw[kw] = *(const u32x4_t*)(xi + ((size_t)ihc * wd + iwc) * C);
        if (!rowok || (unsigned)iw >= (unsigned)wd) raw[kw] = u32x4_t{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        float xv[8];
        unpack8(raw[kw], xv);
        const float4 w0 = *(const float4*)(wc + (size_t)(kh * K + kw) * C);
        const float4 w1 = *(const float4*)(wc + (size_t)(kh * K + kw) * C + 4);
        acc[0] += xv[0] * w0.x; acc[1] += xv[1] * w0.y; acc[2] += xv[2] * w0.z; acc[3] += xv[3] * w0.w;
        acc[4] += xv[4] * w1.x; acc[5] += xv[5] * w1.y; acc[6] += xv[6] * w1.z; acc[7] += xv[7] * w1.w;
      }
    }
    *(u32x4_t*)(y + p * C + cc * 8) = pack8(acc);
  }
}

// dx[n,ih,iw,c] (+)= sum over taps with (ih + pad - kh) divisible by the stride of dy[.., (ih+pad-kh)/s, ..] * w[c][kh][kw]:
// the taps of a pixel are kh0 + t*S with kh0 = (ih + pad) % S, visited in ascending order
template <int K, int S>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ wt,
                                                       bf16_t* __restrict__ dx, int accumulate, int n, int h, int wd, int C,
                                                       int pad, int ho, int wo) {
  constexpr int T = (K + S - 1) / S;
  const int c8 = C >> 3;
  const size_t total = (size_t)n * h * wd * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const size_t p = i / c8;
    const int iw = (int)(p % wd), ih = (int)((p / wd) % h), img = (int)(p / ((size_t)wd * h));
    const bf16_t* gi = dy + (size_t)img * ho * wo * C + cc * 8;
    const float* wc = wt + cc * 8;
    const int a0 = ih + pad, b0 = iw + pad;
    const int kh0 = a0 % S, kw0 = b0 % S;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll(T <= 3 ? T : 1)
    for (int t = 0; t < T; ++t) {
      const int kh = kh0 + t * S, da = a0 - kh;
      const int oh = da / S;
      const bool rowok = kh < K && da >= 0 && oh < ho;
      const int ohc = min(max(oh, 0), ho - 1), khc = min(kh, K - 1);
      u32x4_t raw[T];
      int kwc[T];
#pragma unroll
      for (int u = 0; u < T; ++u) {
        const int kw = kw0 + u * S, db = b0 - kw;
        const int ow = db / S;
        const bool ok = rowok && kw < K && db >= 0 && ow < wo;
        const int owc = min(max(ow, 0), wo - 1);
        kwc[u] = min(kw, K - 1);
        raw[u] = *(const u32x4_t*)(gi + ((size_t)ohc * wo + owc) * C);
        if (!ok) raw[u] = u32x4_t{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < T; ++u) {
        float gv[8];
        unpack8(raw[u], gv);
        const float4 w0 = *(const float4*)(wc + (size_t)(khc * K + kwc[u]) * C);
        const float4 w1 = *(const float4*)(wc + (size_t)(khc * K + kwc[u]) * C + 4);
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

// Pad (K-1)/2: a thread owns 8 channels x 4 adjacent input pixels (iw0 a multiple of 4); a window row's K weight vectors
// and the dy columns the four pixels touch are loaded once.  Stride 2: which (pixel, kw) pairs meet - (u + pad - kw) even -
// and which of the four dy columns iw0/2 - 1 ... iw0/2 + 2 each pair reads are compile-time.  Stride 1 (only used when the
// result accumulates into a trunk gradient; otherwise the forward kernel runs on dy): K + 3 columns from iw0 - pad.
// The gather kernel above re-reads 32 bytes of weights and 16 of dy per tap and pixel and is bound by those L1 requests
// (170-210 us on the 112 x 112 stride-2 layers, 400 us on B4's accumulating stage-1 layer).  Taps are visited in the
// gather kernel's order: same sums.
template <int K, int S>
__global__ __launch_bounds__(256) void dw_dgrad_px_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ wt,
                                                          bf16_t* __restrict__ dx, int accumulate, int n, int h, int wd,
                                                          int C, int ho, int wo) {
  constexpr int PAD = (K - 1) / 2, T = S == 2 ? (K + 1) / 2 : K, PX = 4, COLS = S == 2 ? 4 : K + 3;
  const int c8 = C >> 3, gpr = (wd + PX - 1) / PX;
  const size_t total = (size_t)n * h * gpr * c8;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    const size_t g = i / c8;
    const int iw0 = (int)(g % gpr) * PX, ih = (int)((g / gpr) % h), img = (int)(g / ((size_t)gpr * h));
    const bf16_t* gi = dy + (size_t)img * ho * wo * C + cc * 8;
    const float* wc = wt + cc * 8;
    const int a0 = ih + PAD, kh0 = S == 2 ? (a0 & 1) : 0, owb = S == 2 ? (iw0 >> 1) - 1 : iw0 - PAD;
    float acc[PX][8];
#pragma unroll
    for (int u = 0; u < PX; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[u][j] = 0.f;
    for (int t = 0; t < T; ++t) {
      const int kh = kh0 + S * t, da = a0 - kh;
      const int oh = S == 2 ? da >> 1 : da;
      if (kh >= K || da < 0 || oh >= ho) continue;
      u32x4_t raw[COLS];
#pragma unroll
      for (int c = 0; c < COLS; ++c) {
        const int ow = owb + c;
        raw[c] = *(const u32x4_t*)(gi + ((size_t)oh * wo + min(max(ow, 0), wo - 1)) * C);
        if ((unsigned)ow >= (unsigned)wo) raw[c] = u32x4_t{0u, 0u, 0u, 0u};
      }
      float4 w0[K], w1[K];
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        w0[kw] = *(const float4*)(wc + (size_t)(kh * K + kw) * C);
        w1[kw] = *(const float4*)(wc + (size_t)(kh * K + kw) * C + 4);
      }
      float gv[COLS][8];
#pragma unroll
      for (int c = 0; c < COLS; ++c) unpack8(raw[c], gv[c]);
#pragma unroll
      for (int u = 0; u < PX; ++u)
#pragma unroll
        for (int kw = 0; kw < K; ++kw)
          if (S == 1 || ((u + PAD - kw) & 1) == 0) {
            const int c = S == 2 ? (u + PAD - kw) / 2 + 1 : u + 2 * PAD - kw;   // compile-time after unrolling
            acc[u][0] += gv[c][0] * w0[kw].x; acc[u][1] += gv[c][1] * w0[kw].y;
            acc[u][2] += gv[c][2] * w0[kw].z; acc[u][3] += gv[c][3] * w0[kw].w;
            acc[u][4] += gv[c][4] * w1[kw].x; acc[u][5] += gv[c][5] * w1[kw].y;
            acc[u][6] += gv[c][6] * w1[kw].z; acc[u][7] += gv[c][7] * w1[kw].w;
          }
    }
#pragma unroll
    for (int u = 0; u < PX; ++u) {
      if (iw0 + u >= wd) continue;
      bf16_t* o = dx + ((((size_t)img * h + ih) * wd) + iw0 + u) * C + cc * 8;
      if (accumulate) {
        float q[8];
        unpack8(*(const u32x4_t*)o, q);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[u][j] += q[j];
      }
      *(u32x4_t*)o = pack8(acc[u]);
    }
  }
}

// gw[c][kh][kw] = sum over output pixels of dy * x(shifted): blockIdx.y = kh.  A work item is DW_PX horizontally adjacent
// output pixels x 8 channels: the (DW_PX-1)*S + K input columns they touch are loaded once (all loads of an item in
// flight together, clamped and zeroed by selects), 2-3 vector loads per pixel and window row instead of K + 1.  The items
// in flight of a block are combined through LDS in a fixed order, one partial row [c_log][taps] per block.
constexpr int DW_PX = 4;
template <int K, int S>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                       float* __restrict__ partials, int n, int h, int wd, int C, int c_log,
                                                       int pad, int ho, int wo, int rows_per_block) {
  constexpr int COLS = (DW_PX - 1) * S + K;
  extern __shared__ float sm[];   // [rif][tpr][K][8]
  const int gpr = (wo + DW_PX - 1) / DW_PX;
  const int M = n * ho * gpr;     // items
  const RowWalk w(M, C, rows_per_block, blockIdx.z, gridDim.z);
  const int kh = blockIdx.y;
  float* out = partials + (size_t)blockIdx.x * c_log * K * K;
  for (int c0 = w.cb; c0 < w.ce; c0 += w.tpr) {     // uniform trip count: the barriers below are reached by every thread
    const int cc = c0 + w.lane_c;
    const bool mine = w.active && cc < w.ce;
    float acc[K][8];
#pragma unroll
    for (int q = 0; q < K; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
    if (mine)
      for (int r = w.row0 + w.lane_r; r < w.row1; r += w.rif) {
        const int ow0 = (r % gpr) * DW_PX, oh = (r / gpr) % ho, img = r / (gpr * ho);
        const int ih = oh * S + kh - pad;
        if (ih < 0 || ih >= h) continue;
        const bf16_t* gp = dy + (((size_t)img * ho + oh) * wo) * C + cc * 8;
        const bf16_t* xp = x + (((size_t)img * h + ih) * wd) * C + cc * 8;
        u32x4_t graw[DW_PX], raw[COLS];
#pragma unroll
        for (int u = 0; u < DW_PX; ++u) {
          graw[u] = *(const u32x4_t*)(gp + (size_t)min(ow0 + u, wo - 1) * C);
          if (ow0 + u >= wo) graw[u] = u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int col = 0; col < COLS; ++col) {
          const int iw = ow0 * S - pad + col;
          raw[col] = *(const u32x4_t*)(xp + (size_t)min(max(iw, 0), wd - 1) * C);
          if ((unsigned)iw >= (unsigned)wd) raw[col] = u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < DW_PX; ++u) {
          float gv[8];
          unpack8(graw[u], gv);
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            float xv[8];
            unpack8(raw[u * S + kw], xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[kw][j] += gv[j] * xv[j];
          }
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
      if (c0 + lc >= w.ce || c >= c_log) continue;
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
  extern __shared__ float sm[];   // [rif][TW]
  const int img = blockIdx.x, chunk = blockIdx.y;
  const int rows = (HW + chunks - 1) / chunks;
  const int r0 = chunk * rows, r1 = min(HW, r0 + rows);
  const int c8 = C >> 3;
  const int tile = (c8 + (int)gridDim.z - 1) / (int)gridDim.z;   // channel tile blockIdx.z (as RowWalk)
  const int cb = blockIdx.z * tile, ce = min(c8, cb + tile), TW = tile * 8;
  const int tpr = tile < 256 ? tile : 256, rif = 256 / tpr;
  const int lane_c = threadIdx.x % tpr, lane_r = threadIdx.x / tpr;
  if (lane_r < rif)
    for (int cc = cb + lane_c; cc < ce; cc += tpr) {
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
      for (int j = 0; j < 8; ++j) sm[(size_t)lane_r * TW + (cc - cb) * 8 + j] = s[j];
    }
  __syncthreads();
  float* o = part + ((size_t)img * chunks + chunk) * C + cb * 8;
  for (int i = threadIdx.x; i < (ce - cb) * 8; i += 256) {
    float t = 0.f;
    for (int r = 0; r < rif; ++r) t += sm[(size_t)r * TW + i];
    o[i] = t;
  }
}
// BatchNorm apply + activation of a depthwise layer AND the squeeze of the squeeze-excitation layer behind it: the block
// layout of pool_partial_kernel (one chunk of one image's rows), the sums taken over the rounded values it stores, in
// pool_partial_kernel's order - the separate pooling pass over the tensor just written is gone.
__global__ __launch_bounds__(256) void bna_apply_pool_kernel(const bf16_t* __restrict__ raw, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, bf16_t* __restrict__ out,
                                                             float* __restrict__ part, int HW, int C, int act, int chunks) {
  extern __shared__ float sm[];   // [rif][TW]
  const int img = blockIdx.x, chunk = blockIdx.y;
  const int rows = (HW + chunks - 1) / chunks;
  const int r0 = chunk * rows, r1 = min(HW, r0 + rows);
  const int c8 = C >> 3;
  const int tile = (c8 + (int)gridDim.z - 1) / (int)gridDim.z;
  const int cb = blockIdx.z * tile, ce = min(c8, cb + tile), TW = tile * 8;
  const int tpr = tile < 256 ? tile : 256, rif = 256 / tpr;
  const int lane_c = threadIdx.x % tpr, lane_r = threadIdx.x / tpr;
  if (lane_r < rif)
    for (int cc = cb + lane_c; cc < ce; cc += tpr) {
      float sc[8], sh[8], sum[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc[j] = scale[cc * 8 + j]; sh[j] = shift[cc * 8 + j]; sum[j] = 0.f; }
      for (int r = r0 + lane_r; r < r1; r += rif) {
        const size_t o = ((size_t)img * HW + r) * C + cc * 8;
        float v[8], q[8];
        unpack8(*(const u32x4_t*)(raw + o), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], act);
        const u32x4_t pk = pack8(v);
        *(u32x4_t*)(out + o) = pk;
        unpack8(pk, q);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] += q[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sm[(size_t)lane_r * TW + (cc - cb) * 8 + j] = sum[j];
    }
  __syncthreads();
  float* o = part + ((size_t)img * chunks + chunk) * C + cb * 8;
  for (int i = threadIdx.x; i < (ce - cb) * 8; i += 256) {
    float t = 0.f;
    for (int r = 0; r < rif; ++r) t += sm[(size_t)r * TW + i];
    o[i] = t;
  }
}

// rows of W2 [Cl][S] staged per pass through LDS (row stride S|1: conflict-free both ways), about 47 KB
__host__ __device__ inline int se_tile_rows(int S) {
  const int r = 12032 / (S | 1);
  return r > 256 ? 256 : r;
}

// nrow rows of S floats (contiguous in global memory) -> tile rows of stride ld: flat coalesced loads, four in flight
__device__ __forceinline__ void se_stage_rows(float* __restrict__ tile, const float* __restrict__ src, int nrow, int S,
                                              int ld) {
  const int total = nrow * S;
  for (int i0 = threadIdx.x; i0 < total; i0 += 1024) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + 256 * k < total ? src[i0 + 256 * k] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + 256 * k;
      if (i < total) {
        const int r = i / S;
        tile[r * ld + (i - r * S)] = v[k];
      }
    }
  }
}

// Gates of the squeeze-excitation layer, fp32: u1 = W1 pooled + b1, h1 = silu(u1), gate = sigmoid(W2 h1 + b2).
// Two launches, both (image x tile) grids: with one block per image the block walked both matrices by itself - a chain of
// dependent L2 round trips, 60-150 us for the 2688-channel layers of B4 however little arithmetic that is.
// fc1: block = (image, 16 hidden units); W1 [S][Cl] read with the lanes along Cl, 64 loads in flight per thread.
__global__ __launch_bounds__(256) void se_fc1_train_kernel(const float* __restrict__ part, int chunks, float scale,
                                                           float* __restrict__ pooled, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, float* __restrict__ u1,
                                                           float* __restrict__ h1, int C, int Cl, int S) {
  extern __shared__ float sm[];   // pooled[Cl], red[4][16]
  float* sp = sm;
  float* red = sm + Cl;
  const int img = blockIdx.x, s0 = blockIdx.y * 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the squeeze: mean over HW from the per-chunk channel sums of the pooling pass (every block; block 0 keeps it for the
  // backward pass)
  for (int c = threadIdx.x; c < Cl; c += 256) {
    float t = 0.f;
    for (int k = 0; k < chunks; ++k) t += part[((size_t)img * chunks + k) * C + c];
    t *= scale;
    sp[c] = t;
    if (blockIdx.y == 0) pooled[(size_t)img * C + c] = t;
  }
  __syncthreads();
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  for (int c0 = threadIdx.x; c0 < Cl; c0 += 1024) {
    float wv[4][16], pv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = min(c0 + 256 * q, Cl - 1);
      pv[q] = c0 + 256 * q < Cl ? sp[c] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) wv[q][k] = W1[(size_t)min(s0 + k, S - 1) * Cl + c];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] += wv[q][k] * pv[q];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    for (int d = 32; d; d >>= 1) acc[k] += __shfl_xor(acc[k], d);
    if (lane == 0) red[wave * 16 + k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 16 && s0 + threadIdx.x < S) {
    const int sidx = s0 + threadIdx.x;
    const float u = ((red[threadIdx.x] + red[16 + threadIdx.x]) + red[32 + threadIdx.x]) + red[48 + threadIdx.x] + b1[sidx];
    u1[(size_t)img * S + sidx] = u;
    h1[(size_t)img * S + sidx] = u / (1.f + expf(-u));
  }
}

// fc2: block = (image, tile of R channels): the tile's rows of W2 [Cl][S] go through LDS (coalesced loads of whole rows,
// then one row per thread; a thread walking its own row in global memory touches 64 cache lines per wave and load)
__global__ __launch_bounds__(256) void se_fc2_train_kernel(const float* __restrict__ h1, const float* __restrict__ W2,
                                                           const float* __restrict__ b2, float* __restrict__ gate, int C,
                                                           int Cl, int S, int R) {
  extern __shared__ float sm[];   // h[S], tile[R][S|1]
  float* shh = sm;
  float* tile = sm + S;
  const int img = blockIdx.x, c0 = blockIdx.y * R, nrow = min(R, Cl - c0), ld = S | 1;
  for (int i = threadIdx.x; i < S; i += 256) shh[i] = h1[(size_t)img * S + i];
  se_stage_rows(tile, W2 + (size_t)c0 * S, nrow, S, ld);
  __syncthreads();
  for (int r = threadIdx.x; r < nrow; r += 256) {
    float acc = b2[c0 + r];
    for (int sidx = 0; sidx < S; ++sidx) acc += tile[r * ld + sidx] * shh[sidx];
    gate[(size_t)img * C + c0 + r] = 1.f / (1.f + expf(-acc));
  }
}

// Backward of the gate path.  in: pool_part[img][chunk][c], the per-chunk sums over HW of g*a.
// bwd1: block = (image, tile of R channels): du2 = (sum of the chunks)*s(1-s) -> dgate, and the tile's share of W2^T du2:
// part[img][tile][s] = sum_{c in tile} du2[c] W2[c][s] (lanes along s, the four waves take rows r, r+4, ...)
__global__ __launch_bounds__(256) void se_bwd1_kernel(const float* __restrict__ pool_part, int chunks,
                                                      float* __restrict__ dgate, const float* __restrict__ gate,
                                                      const float* __restrict__ W2, float* __restrict__ part, int C, int Cl,
                                                      int S, int R) {
  extern __shared__ float sm[];   // du2[R], tile[R][S|1] (then the four row-group partials [4][S])
  float* s2 = sm;
  float* tile = sm + R;
  const int img = blockIdx.x, c0 = blockIdx.y * R, nrow = min(R, Cl - c0), ld = S | 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = threadIdx.x; r < nrow; r += 256) {
    const float g = gate[(size_t)img * C + c0 + r];
    float dg = 0.f;   // sum over HW of g*a from the per-chunk sums of the pooling pass
    for (int k = 0; k < chunks; ++k) dg += pool_part[((size_t)img * chunks + k) * C + c0 + r];
    const float v = dg * g * (1.f - g);
    dgate[(size_t)img * C + c0 + r] = v;
    s2[r] = v;
  }
  se_stage_rows(tile, W2 + (size_t)c0 * S, nrow, S, ld);
  __syncthreads();
  float acc[3] = {0.f, 0.f, 0.f};   // hidden units lane, lane + 64, lane + 128 (S <= 192)
  for (int r = wave; r < nrow; r += 4) {
    const float v = s2[r];
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (lane + 64 * q < S) acc[q] += v * tile[r * ld + lane + 64 * q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (lane + 64 * q < S) tile[wave * S + lane + 64 * q] = acc[q];
  __syncthreads();
  float* o = part + ((size_t)img * gridDim.y + blockIdx.y) * S;
  for (int sidx = threadIdx.x; sidx < S; sidx += 256)
    o[sidx] = ((tile[sidx] + tile[S + sidx]) + tile[2 * S + sidx]) + tile[3 * S + sidx];
}

// bwd2: block = (image, 1024 channels): du1[s] = silu'(u1[s]) * sum over the tiles of part (every block, block 0 stores
// it), dpool[c] = sum_s du1[s] W1[s][c] for the block's channels (lanes along Cl, 16 loads in flight)
__global__ __launch_bounds__(256) void se_bwd2_kernel(const float* __restrict__ part, int tiles, const float* __restrict__ u1,
                                                      const float* __restrict__ W1, float* __restrict__ du1,
                                                      float* __restrict__ dpool, int C, int Cl, int S) {
  extern __shared__ float s1[];   // du1[S]
  const int img = blockIdx.x;
  for (int sidx = threadIdx.x; sidx < S; sidx += 256) {
    float acc = 0.f;
    for (int t = 0; t < tiles; ++t) acc += part[((size_t)img * tiles + t) * S + sidx];
    const float z = u1[(size_t)img * S + sidx], sg = 1.f / (1.f + expf(-z));
    const float v = acc * sg * (1.f + z * (1.f - sg));
    s1[sidx] = v;
    if (blockIdx.y == 0) du1[(size_t)img * S + sidx] = v;
  }
  __syncthreads();
  const int cbase = blockIdx.y * 1024 + threadIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int cq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) cq[q] = min(cbase + 256 * q, Cl - 1);
#pragma unroll 4
  for (int sidx = 0; sidx < S; ++sidx) {
    const float v = s1[sidx];
    const float* wr = W1 + (size_t)sidx * Cl;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += v * wr[cq[q]];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (cbase + 256 * q < Cl) dpool[(size_t)img * C + cbase + 256 * q] = acc[q];
}

// parameter gradients of the two 1x1 convs of the gate path (sums over the batch in ascending order); null: not wanted.
// Two small GEMMs G[c][s] = sum_k A[k][c] B[k][s]: blockIdx.y = 0: gW2 [Cl][S] from A = du2, B = h1 (+ gb2 = column sums
// of A); 1: gW1 [S][Cl] from A = pooled, B = du1 (+ gb1 = column sums of B).  A block owns 64 channels x all S: the batch
// is staged through LDS 32 images at a time, a thread keeps 4 channels x ceil(S/16) hidden units in registers (one
// thread per output reading both operands from global memory took 85 us per layer).
constexpr int SEW_KC = 32, SEW_SJ = 12;   // S <= 192
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ du2, const float* __restrict__ h1,
                                                       const float* __restrict__ du1, const float* __restrict__ pooled,
                                                       float* __restrict__ gW1, float* __restrict__ gb1,
                                                       float* __restrict__ gW2, float* __restrict__ gb2, int n, int C, int Cl,
                                                       int S) {
  extern __shared__ float sm[];   // As[KC][64], Bs[KC][S]
  const int job = blockIdx.y;
  const float* A = job == 0 ? du2 : pooled;
  const float* B = job == 0 ? h1 : du1;
  float* G = job == 0 ? gW2 : gW1;
  float* gbA = job == 0 ? gb2 : nullptr;
  float* gbB = job == 0 ? nullptr : gb1;
  if (!G && !gbA && !gbB) return;
  float* As = sm;
  float* Bs = sm + SEW_KC * 64;
  const int c0 = blockIdx.x * 64, tc = threadIdx.x & 15, ts = threadIdx.x >> 4;
  const int sj = (S + 15) >> 4;
  float acc[4][SEW_SJ], sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[SEW_SJ];
#pragma unroll
  for (int j = 0; j < SEW_SJ; ++j) {
    sb[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] = 0.f;
  }
  for (int k0 = 0; k0 < n; k0 += SEW_KC) {
    const int kc = min(SEW_KC, n - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < kc * 64; i += 256) {
      const int kk = i >> 6, c = c0 + (i & 63);
      As[i] = c < Cl ? A[(size_t)(k0 + kk) * C + c] : 0.f;
    }
    for (int i = threadIdx.x; i < kc * S; i += 256) Bs[i] = B[(size_t)k0 * S + i];
    __syncthreads();
    for (int kk = 0; kk < kc; ++kk) {
      const float4 a = *(const float4*)(As + kk * 64 + tc * 4);
      sa[0] += a.x; sa[1] += a.y; sa[2] += a.z; sa[3] += a.w;
#pragma unroll
      for (int j = 0; j < SEW_SJ; ++j)
        if (j < sj) {
          const int sidx = ts + 16 * j;
          const float bv = sidx < S ? Bs[kk * S + sidx] : 0.f;
          sb[j] += bv;
          acc[0][j] += a.x * bv; acc[1][j] += a.y * bv; acc[2][j] += a.z * bv; acc[3][j] += a.w * bv;
        }
    }
  }
#pragma unroll
  for (int j = 0; j < SEW_SJ; ++j) {
    const int sidx = ts + 16 * j;
    if (j >= sj || sidx >= S) continue;
    if (gbB && blockIdx.x == 0 && tc == 0) gbB[sidx] = sb[j];
    if (G) {
      const float v[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + tc * 4 + i;
        if (c < Cl) G[job == 0 ? (size_t)c * S + sidx : (size_t)sidx * Cl + c] = v[i];
      }
    }
  }
  if (gbA && ts == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c0 + tc * 4 + i < Cl) gbA[c0 + tc * 4 + i] = sa[i];
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

// wgrad slabs of a channel-padded GEMM [splits][cout_p][taps*cin_p] -> gradient [cout][taps][cin], fixed order.
// A block owns 32 consecutive outputs x 8 split lanes (lane l sums splits l, l + 8, ...; the eight partial sums are added
// in order through LDS): with one thread per output walking all (up to 384) splits, the small layers - 16 x 96 outputs on
// six blocks - were a chain of 384 dependent loads, 89 us each, 2 ms of the second stream per EfficientNet-B0 step.
__global__ __launch_bounds__(256) void slab_reduce_sub_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                              int cout, int taps, int cin, int cout_p, int cin_p,
                                                              int splits) {
  __shared__ float part[8][32];
  const size_t n = (size_t)cout * taps * cin;
  const size_t slab = (size_t)cout_p * taps * cin_p;
  const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
  for (size_t i0 = (size_t)blockIdx.x * 32; i0 < n; i0 += (size_t)gridDim.x * 32) {   // uniform trip count per block
    const size_t i = i0 + ol;
    float acc = 0.f;
    if (i < n) {
      const int ci = (int)(i % cin), t = (int)((i / cin) % taps), co = (int)(i / ((size_t)cin * taps));
      const size_t src = ((size_t)co * taps + t) * cin_p + ci;
      for (int k = sl; k < splits; k += 8) acc += slabs[(size_t)k * slab + src];
    }
    part[sl][ol] = acc;
    __syncthreads();
    if (sl == 0 && i < n) {
      float t = part[0][ol];
#pragma unroll
      for (int q = 1; q < 8; ++q) t += part[q][ol];
      out[i] = t;
    }
    __syncthreads();
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
// channel tiles of about 256 channels once a tensor is wider than that
inline int walk_ctiles(int C) { return (C / 8 + 31) / 32; }
inline size_t walk_lds(int C) {
  const int c8 = C / 8, cts = walk_ctiles(C), tile = (c8 + cts - 1) / cts, tpr = tile < 256 ? tile : 256;
  return (size_t)(256 / tpr) * 2 * tile * 8 * sizeof(float);
}
inline size_t pool_lds(int C) { return walk_lds(C) / 2; }   // [rif][TW] floats
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -1)

}  // namespace

int spk_launch_col_stats(const bf16_t* x, float* partials, int M, int C, int* blocks, hipStream_t s) {
  int rpb;
  const int nb = walk_rows(M, &rpb);
  hipLaunchKernelGGL(col_stats_kernel, dim3(nb, walk_ctiles(C)), dim3(256), walk_lds(C), s, x, partials, M, C, rpb);
  *blocks = nb;
  return LAUNCH_OK();
}

int spk_launch_bna_apply(const bf16_t* raw, const float* scale, const float* shift, const bf16_t* res,
                         const float* rowscale, bf16_t* out, int M, int C, int HW, int act, hipStream_t s) {
  int rpb;
  const int nb = walk_rows(M, &rpb);
  hipLaunchKernelGGL(bna_apply_kernel, dim3(nb, walk_ctiles(C)), dim3(256), 0, s, raw, scale, shift, res, rowscale, out, M, C, HW, act,
                     rpb);
  return LAUNCH_OK();
}

int spk_launch_bna_bwd_reduce(const bf16_t* g, const bf16_t* raw, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const float* rowscale, float* partials, int M,
                              int C, int HW, int act, int* blocks, hipStream_t s) {
  int rpb;
  const int nb = walk_rows(M, &rpb);
  hipLaunchKernelGGL(bna_bwd_reduce_kernel, dim3(nb, walk_ctiles(C)), dim3(256), walk_lds(C), s, g, raw, scale, shift, mean, invstd,
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
  hipLaunchKernelGGL(bna_bwd_apply_kernel, dim3(nb, walk_ctiles(C)), dim3(256), 0, s, g, raw, scale, shift, mean, invstd, coef, rowscale,
                     dy, g_res, res_accumulate, M, C, HW, act, rpb);
  return LAUNCH_OK();
}

int spk_launch_sd_rowscale(float* rs, int n, float p, unsigned long long seed, hipStream_t s) {
  hipLaunchKernelGGL(sd_rowscale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rs, n, p, seed);
  return LAUNCH_OK();
}

int spk_launch_stem3_train_fwd(const bf16_t* x, const float* w, bf16_t* y, int n, int h, int wd, int wstride, int cin,
                               int cout, int C, int ho, int wo, hipStream_t s, float out_scale) {
  const size_t total = (size_t)n * ho * ((wo + STEM_PX - 1) / STEM_PX) * (C / 8);
  hipLaunchKernelGGL(stem3_fwd_kernel, dim3(grid_of(total, 256)), dim3(256), (size_t)36 * C * 4, s, x, w, y, n, h, wd,
                     wstride, cin, cout, C, ho, wo, out_scale);
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
  if (cout % 4 == 0 && 9 * (cout / 4) <= 256) {
    const int items = 9 * (cout / 4), G = 256 / items;
    const size_t lds = (size_t)std::max(64 * cout + 64 * 36, G * items * 16) * 4;
    hipLaunchKernelGGL(stem3_wgrad_rt_kernel, dim3(nb), dim3(256), lds, s, x, dy, partials, n, h, wd, wstride, cin, cout, C,
                       ho, wo, ppb);
    *blocks = nb;
    return LAUNCH_OK();
  }
  hipLaunchKernelGGL(stem3_wgrad_kernel, dim3(nb), dim3(256), (size_t)(64 * C + 64 * 36) * 4, s, x, dy, partials, n, h,
                     wd, wstride, cin, cout, C, ho, wo, ppb);
  *blocks = nb;
  return LAUNCH_OK();
}

int spk_launch_dw_train_fwd(const bf16_t* x, const float* wt, bf16_t* y, int n, int h, int wd, int C, int k, int stride,
                            int pad, int ho, int wo, hipStream_t s) {
  const size_t total = (size_t)n * ho * wo * (C / 8);
  if (k == 3)
    hipLaunchKernelGGL(dw_fwd_kernel<3>, dim3(grid_of(total, 256)), dim3(256), 0, s, x, wt, y, n, h, wd, C, stride, pad, ho,
                       wo);
  else if (k == 5)
    hipLaunchKernelGGL(dw_fwd_kernel<5>, dim3(grid_of(total, 256)), dim3(256), 0, s, x, wt, y, n, h, wd, C, stride, pad, ho,
                       wo);
  else
    return -1;
  return LAUNCH_OK();
}

int spk_launch_dw_dgrad(const bf16_t* dy, const float* wt, bf16_t* dx, int accumulate, int n, int h, int wd, int C, int k,
                        int stride, int pad, int ho, int wo, hipStream_t s) {
  const size_t total = (size_t)n * h * wd * (C / 8);
#define DW_DGRAD(K_, S_)                                                                                              \
  hipLaunchKernelGGL((dw_dgrad_kernel<K_, S_>), dim3(grid_of(total, 256)), dim3(256), 0, s, dy, wt, dx, accumulate, n, h, \
                     wd, C, pad, ho, wo)
  if ((stride == 1 || stride == 2) && pad == (k - 1) / 2 && (k == 3 || k == 5)) {
    const dim3 grid(grid_of((size_t)n * h * ((wd + 3) / 4) * (C / 8), 256));
#define DW_DGRAD_PX(K_, S_)                                                                                             \
  hipLaunchKernelGGL((dw_dgrad_px_kernel<K_, S_>), grid, dim3(256), 0, s, dy, wt, dx, accumulate, n, h, wd, C, ho, wo)
    if (k == 3 && stride == 1) DW_DGRAD_PX(3, 1);
    else if (k == 3) DW_DGRAD_PX(3, 2);
    else if (stride == 1) DW_DGRAD_PX(5, 1);
    else DW_DGRAD_PX(5, 2);
#undef DW_DGRAD_PX
    return LAUNCH_OK();
  }
  if (k == 3 && stride == 1) DW_DGRAD(3, 1);
  else if (k == 3 && stride == 2) DW_DGRAD(3, 2);
  else if (k == 5 && stride == 1) DW_DGRAD(5, 1);
  else if (k == 5 && stride == 2) DW_DGRAD(5, 2);
  else return -1;
#undef DW_DGRAD
  return LAUNCH_OK();
}

// partial rows the weight-gradient kernel writes for an [M][C] problem ([rows][c_log][k*k] floats): one per block
static int dw_wgrad_blocks(int M, int* rows_per_block) {
  int rpb = 8192;
  while (rpb > 64 && (M + rpb - 1) / rpb < 256) rpb >>= 1;
  *rows_per_block = rpb;
  return (M + rpb - 1) / rpb;
}
int spk_dw_wgrad_rows(int M, int C) {   // upper bound at planning time: the launch walks M / DW_PX or more items
  int rpb;
  return std::max(dw_wgrad_blocks(M, &rpb), std::min(512, (M + 63) / 64 + 1));
}

int spk_launch_dw_wgrad(const bf16_t* x, const bf16_t* dy, float* partials, int n, int h, int wd, int C, int c_log,
                        int k, int stride, int pad, int ho, int wo, int* rows, hipStream_t s) {
  int rpb;
  const int M = n * ho * ((wo + DW_PX - 1) / DW_PX);
  const int nb = dw_wgrad_blocks(M, &rpb);
  *rows = nb;
  const int c8 = C / 8, cts = walk_ctiles(C), tile = (c8 + cts - 1) / cts, tpr = tile < 256 ? tile : 256;
  const size_t lds = (size_t)(256 / tpr) * tpr * k * 8 * sizeof(float);
#define DW_WGRAD(K_, S_)                                                                                          \
  hipLaunchKernelGGL((dw_wgrad_kernel<K_, S_>), dim3(nb, K_, cts), dim3(256), lds, s, x, dy, partials, n, h, wd, C, \
                     c_log, pad, ho, wo, rpb)
  if (k == 3 && stride == 1) DW_WGRAD(3, 1);
  else if (k == 3 && stride == 2) DW_WGRAD(3, 2);
  else if (k == 5 && stride == 1) DW_WGRAD(5, 1);
  else if (k == 5 && stride == 2) DW_WGRAD(5, 2);
  else return -1;
#undef DW_WGRAD
  return LAUNCH_OK();
}

int spk_se_chunks(int HW) { return HW >= 3136 ? 16 : (HW >= 196 ? 4 : 1); }

// part[n][chunks][C] = per-chunk sums over HW of x (* y); the squeeze-excitation gate kernels sum the chunks themselves
int spk_launch_pool_rows(const bf16_t* x, const bf16_t* y, float* part, int n, int HW, int C, hipStream_t s) {
  const int chunks = spk_se_chunks(HW);
  hipLaunchKernelGGL(pool_partial_kernel, dim3(n, chunks, walk_ctiles(C)), dim3(256), pool_lds(C), s, x, y, part, HW, C, chunks);
  return LAUNCH_OK();
}

// a = act(raw*scale + shift) and part[n][chunks][C] = per-chunk sums over HW of a (the squeeze of the layer behind)
int spk_launch_bna_apply_pool(const bf16_t* raw, const float* scale, const float* shift, bf16_t* out, float* part, int n,
                              int HW, int C, int act, hipStream_t s) {
  const int chunks = spk_se_chunks(HW);
  hipLaunchKernelGGL(bna_apply_pool_kernel, dim3(n, chunks, walk_ctiles(C)), dim3(256), pool_lds(C), s, raw, scale, shift, out, part,
                     HW, C, act, chunks);
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

int spk_launch_slab_reduce_sub(const float* slabs, float* out, int cout, int taps, int cin, int cout_p, int cin_p,
                               int splits, hipStream_t s) {
  hipLaunchKernelGGL(slab_reduce_sub_kernel, dim3(grid_of((size_t)cout * taps * cin, 32)), dim3(256), 0, s, slabs, out,
                     cout, taps, cin, cout_p, cin_p, splits);
  return LAUNCH_OK();
}

// the ordered reductions themselves (one launch each: ordered_reduce.h)
int spk_launch_bna_finalize(const float* partials, int count, int C, int c_log, double M, const float* gamma,
                            const float* beta, float* rmean, float* rvar, float* st, float eps, float momentum,
                            float* tmp, hipStream_t s) {
  const BnaFwdFin fin = {C, c_log, M, gamma, beta, rmean, rvar, st, eps, momentum};
  return spk_reduce::reduce_finalize(partials, count, C, tmp, fin, s);
}

int spk_launch_bna_bwd_finalize(const float* partials, int count, int C, int c_log, double M, const float* gamma,
                                const float* invstd, float* dgamma, float* dbeta, float* coef, float* tmp, hipStream_t s) {
  const BnaBwdFin fin = {C, c_log, M, gamma, invstd, dgamma, dbeta, coef};
  return spk_reduce::reduce_finalize(partials, count, C, tmp, fin, s);
}

int spk_launch_pack_padded_multi(const float* pbuf, bf16_t* wpack, float* dwt, const PadPackTable& t, hipStream_t s) {
  if (t.count == 0) return 0;
  hipLaunchKernelGGL(pack_padded_multi_kernel, dim3(64, t.count), dim3(256), 0, s, pbuf, wpack, dwt, t);
  return LAUNCH_OK();
}

int spk_se_gate_tiles(int Cl, int S) {
  const int R = se_tile_rows(S);
  return (Cl + R - 1) / R;
}

// part: [n][chunks][C] channel sums of the pooling pass (spk_launch_pool_rows / spk_launch_bna_apply_pool),
// pooled = scale * their sum is stored for the backward pass
int spk_launch_se_gate_fwd(const float* part, int chunks, float scale, float* pooled, const float* W1, const float* b1,
                           const float* W2, const float* b2, float* u1, float* h1, float* gate, int n, int C, int Cl, int S,
                           hipStream_t s) {
  if (S > 192) return -1;
  const int R = se_tile_rows(S);
  hipLaunchKernelGGL(se_fc1_train_kernel, dim3(n, (S + 15) / 16), dim3(256), (size_t)(Cl + 64) * 4, s, part, chunks, scale,
                     pooled, W1, b1, u1, h1, C, Cl, S);
  hipLaunchKernelGGL(se_fc2_train_kernel, dim3(n, spk_se_gate_tiles(Cl, S)), dim3(256),
                     ((size_t)S + (size_t)R * (S | 1)) * 4, s, h1, W2, b2, gate, C, Cl, S, R);
  return LAUNCH_OK();
}

// part: [n][spk_se_gate_tiles(Cl, S)][S] floats of scratch
int spk_launch_se_gate_bwd(const float* pool_part, int chunks, float* dgate, const float* gate, const float* u1,
                           const float* W1, const float* W2, float* du1, float* dpool, float* part, int n, int C, int Cl,
                           int S, hipStream_t s) {
  if (S > 192) return -1;
  const int R = se_tile_rows(S), tiles = spk_se_gate_tiles(Cl, S);
  hipLaunchKernelGGL(se_bwd1_kernel, dim3(n, tiles), dim3(256), ((size_t)R + (size_t)std::max(R, 4) * (S | 1)) * 4, s,
                     pool_part, chunks, dgate, gate, W2, part, C, Cl, S, R);
  hipLaunchKernelGGL(se_bwd2_kernel, dim3(n, (Cl + 1023) / 1024), dim3(256), (size_t)S * 4, s, part, tiles, u1, W1, du1,
                     dpool, C, Cl, S);
  return LAUNCH_OK();
}

int spk_launch_se_wgrad(const float* du2, const float* h1, const float* du1, const float* pooled, float* gW1, float* gb1,
                        float* gW2, float* gb2, int n, int C, int Cl, int S, hipStream_t s) {
  if (!gW1 && !gb1 && !gW2 && !gb2) return 0;
  if (S > 16 * SEW_SJ) return -1;
  hipLaunchKernelGGL(se_wgrad_kernel, dim3((Cl + 63) / 64, 2), dim3(256), (size_t)SEW_KC * (64 + S) * sizeof(float), s, du2,
                     h1, du1, pooled, gW1, gb1, gW2, gb2, n, C, Cl, S);
  return LAUNCH_OK();
}
