// HBM-bound helper kernels of the inference path (gfx950): image batch ->
// NHWC bf16, 3x3/2 max-pool, global average pool, BatchNorm(eval) folding and
// weight packing.  All of them move 16 B (or 8 B) per lane, coalesced along
// the channel-minor NHWC axis; none has data reuse worth LDS.
//
// Reference ops they stand in for (torch modules reached through `net(x)`,
// sykepic/compute/probability.py:189): MaxPool2d(3,2,1), AdaptiveAvgPool2d(1),
// BatchNorm2d in eval mode; `x.to(device)` layout conversion at :188.
#include "spk_common.h"

namespace {

// one thread = one output pixel (4 bf16 channels = 8 B)
template <typename T, bool NHWC, int DT>
__global__ void to_nhwc4_kernel(const T* __restrict__ x, bf16_t* __restrict__ out, int n, int c,
                                int h, int w, int wp, float mul) {
  const size_t total = (size_t)n * h * wp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int px = (int)(i % wp);
    const size_t row = i / wp;  // img*h + y
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (px < w) {
      const int img = (int)(row / h), y = (int)(row % h);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        if (ch < c) {
          const size_t src = NHWC ? (((size_t)img * h + y) * w + px) * c + ch
                                  : (((size_t)img * c + ch) * h + y) * w + px;
          v[ch] = (float)x[src] * mul;   // (u8: mul = scale / 255, exactly 1 for scale 255)
        }
      }
    }
    u32x2_t o;
    o[0] = pack2<DT>(v[0], v[1]);
    o[1] = pack2<DT>(v[2], v[3]);
    *(u32x2_t*)(out + i * 4) = o;
  }
}

// float NCHW planes, w % 4 == 0 (the benched 224x224 input): one thread = four pixels of a row - three 16-byte loads, two
// 16-byte stores, 32-bit index arithmetic (blockIdx.y = image).  The general kernel above spends its time on three 64-bit
// divisions per pixel: 3.7 TB/s on the 256 x 3 x 224 x 224 batch against ~5 for a copy.
template <int DT>
__global__ __launch_bounds__(256) void to_nhwc4_planes_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int c,
                                                              int hw, float mul) {
  const int q = blockIdx.x * 256 + threadIdx.x;   // group of four pixels inside the image
  if (q * 4 >= hw) return;
  const float* xi = x + (size_t)blockIdx.y * c * hw + q * 4;
  f32x4_t v[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) v[ch] = ch < c ? *(const f32x4_t*)(xi + (size_t)ch * hw) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t o[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    o[p >> 1][(p & 1) * 2] = pack2<DT>(v[0][p] * mul, v[1][p] * mul);
    o[p >> 1][(p & 1) * 2 + 1] = pack2<DT>(v[2][p] * mul, 0.f);
  }
  u32x4_t* d = (u32x4_t*)(out + ((size_t)blockIdx.y * hw + q * 4) * 4);
  d[0] = o[0];
  d[1] = o[1];
}

template <int DT>
__device__ __forceinline__ void max8(u32x4_t& acc, const u32x4_t v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // max of two exactly representable values is exactly representable
    const bool l = lo_f32<DT>(v[j]) > lo_f32<DT>(acc[j]);
    const bool h = hi_f32<DT>(v[j]) > hi_f32<DT>(acc[j]);
    acc[j] = ((l ? v[j] : acc[j]) & 0xffffu) | ((h ? v[j] : acc[j]) & 0xffff0000u);
  }
}

// one thread = 8 channels of one output pixel
template <int DT>
__global__ void maxpool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int n, int h,
                               int w, int c, int k, int stride, int pad, int ho, int wo) {
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
    const unsigned NEG = DT == DT_BF16 ? 0xff80ff80u : 0xfc00fc00u;  // two -inf
    u32x4_t acc = {NEG, NEG, NEG, NEG};
    for (int r = 0; r < k; ++r) {
      const int iy = oy * stride - pad + r;
      if ((unsigned)iy >= (unsigned)h) continue;
      for (int s = 0; s < k; ++s) {
        const int ix = ox * stride - pad + s;
        if ((unsigned)ix >= (unsigned)w) continue;
        max8<DT>(acc, *(const u32x4_t*)(x + (((size_t)img * h + iy) * w + ix) * c + cc * 8));
      }
    }
    *(u32x4_t*)(y + i * 8) = acc;
  }
}

// block = 64 channel groups (8 channels each) of one image x 4 pixel slices: thread (q, cc) sums pixels q, q+4, ... in fp32,
// the four slice sums are added in index order through LDS (a fixed grouping: an image's mean does not depend on the
// batch it is in).  One thread walking all hw pixels of its channels (rounds 1-3) kept 12 % of the CUs busy on the 7x7
// maps of a 128-image half batch with a 49-deep load chain each: 31 us for 26 MB.
template <int DT>
__global__ __launch_bounds__(256) void gavgpool_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int n, int hw,
                                                       int c) {
  __shared__ float part[4][64][9];
  const int c8 = c >> 3;
  const int img = blockIdx.y, cc = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cc < c8) {
    const bf16_t* p = x + (size_t)img * hw * c + cc * 8;
#pragma unroll 4
    for (int t = q; t < hw; t += 4) {
      const u32x4_t v = *(const u32x4_t*)(p + (size_t)t * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[2 * j] += lo_f32<DT>(v[j]);
        s[2 * j + 1] += hi_f32<DT>(v[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[q][threadIdx.x & 63][j] = s[j];
  __syncthreads();
  // 512 outputs of the block, two per thread
  const float inv = 1.0f / (float)hw;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int o = threadIdx.x + h * 256, g = o >> 3, j = o & 7;
    const int gc = blockIdx.x * 64 + g;
    if (gc < c8)
      y[(size_t)img * c + gc * 8 + j] = (((part[0][g][j] + part[1][g][j]) + part[2][g][j]) + part[3][g][j]) * inv;
  }
}

__global__ void bn_fold_kernel(const float* g, const float* b, const float* mean, const float* var,
                               float eps, float* scale, float* bias, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const float sc = g[i] / sqrtf(var[i] + eps);
  scale[i] = sc;
  bias[i] = b[i] - mean[i] * sc;
}

template <int DT>
__device__ __forceinline__ float h16_to_f32(unsigned short v) { return lo_f32<DT>((unsigned int)v); }

template <int DT>
__global__ void pack_generic_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, size_t n,
                                    int splitw) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const unsigned short hi = to_h16<DT>(w[i]);
    out[i] = hi;
    if (splitw) out[n + i] = to_h16<DT>(w[i] - h16_to_f32<DT>(hi));
  }
}

// stem image: [Cout][8 filter rows][8 taps][4 ch]; tap t <-> filter column
// t-1 (tap 0 is the zero that 16-B-aligns the pixel pairs), row 7 / ch 3 zero.
template <int DT>
__global__ void pack_stem_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout,
                                 int kh, int kw, int cin, int splitw, float wscale) {
  const int total = cout * 256;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ch = i & 3, t = (i >> 2) & 7, r = (i >> 5) & 7, co = i >> 8;
  float v = 0.f;
  const int s = t - 1;
  if (r < kh && s >= 0 && s < kw && ch < cin) v = w[(((size_t)co * kh + r) * kw + s) * cin + ch] * wscale;
  const unsigned short hi = to_h16<DT>(v);
  out[i] = hi;
  if (splitw) out[total + i] = to_h16<DT>(v - h16_to_f32<DT>(hi));
}

inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  if (g > 256 * 8 * 4) g = 256 * 8 * 4;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define DT_DISPATCH(dt, CALL_BF16, CALL_F16) \
  do { if ((dt) == DT_F16) { CALL_F16; } else { CALL_BF16; } } while (0)

template <typename T, bool NHWC>
static void launch_to_nhwc4(const void* x, bf16_t* out, int n, int c, int h, int w, int wp, int dt, int g,
                            hipStream_t s, float mul) {
  DT_DISPATCH(dt,
              hipLaunchKernelGGL((to_nhwc4_kernel<T, NHWC, DT_BF16>), dim3(g), dim3(256), 0, s, (const T*)x, out, n, c, h, w, wp, mul),
              hipLaunchKernelGGL((to_nhwc4_kernel<T, NHWC, DT_F16>), dim3(g), dim3(256), 0, s, (const T*)x, out, n, c, h, w, wp, mul));
}

int spk_launch_to_nhwc4(const void* x, int layout, int dtype, int n, int c, int h, int w,
                        bf16_t* out, int dt, hipStream_t s, float scale) {
  if (c < 1 || c > 4) return -1;
  const int wp = (w + 1) & ~1;
  const size_t total = (size_t)n * h * wp;
  const int g = grid_for(total, 256);
  if (dtype == 0 && layout == 0 && c <= 3 && w % 4 == 0 && n <= 65535 && ((size_t)x & 15) == 0) {
    const dim3 gp((h * w / 4 + 255) / 256, n);
    DT_DISPATCH(dt,
                hipLaunchKernelGGL(to_nhwc4_planes_kernel<DT_BF16>, gp, dim3(256), 0, s, (const float*)x, out, c, h * w, scale),
                hipLaunchKernelGGL(to_nhwc4_planes_kernel<DT_F16>, gp, dim3(256), 0, s, (const float*)x, out, c, h * w, scale));
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (dtype == 0) {
    if (layout == 0) launch_to_nhwc4<float, false>(x, out, n, c, h, w, wp, dt, g, s, scale);
    else launch_to_nhwc4<float, true>(x, out, n, c, h, w, wp, dt, g, s, scale);
  } else if (dtype == 2) {
    if (layout == 0) launch_to_nhwc4<unsigned char, false>(x, out, n, c, h, w, wp, dt, g, s, scale / 255.0f);
    else launch_to_nhwc4<unsigned char, true>(x, out, n, c, h, w, wp, dt, g, s, scale / 255.0f);
  } else {
    return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_maxpool(const bf16_t* x, bf16_t* y, int n, int h, int w, int c, int k, int stride,
                       int pad, int ho, int wo, int dt, hipStream_t s) {
  if (c % 8) return -1;
  const size_t total = (size_t)n * ho * wo * (c / 8);
  const int g = grid_for(total, 256);
  DT_DISPATCH(dt,
              hipLaunchKernelGGL(maxpool_kernel<DT_BF16>, dim3(g), dim3(256), 0, s, x, y, n, h, w, c, k, stride, pad, ho, wo),
              hipLaunchKernelGGL(maxpool_kernel<DT_F16>, dim3(g), dim3(256), 0, s, x, y, n, h, w, c, k, stride, pad, ho, wo));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_gavgpool(const bf16_t* x, float* y, int n, int hw, int c, int dt, hipStream_t s) {
  if (c % 8) return -1;
  const dim3 g((c / 8 + 63) / 64, n);
  DT_DISPATCH(dt,
              hipLaunchKernelGGL(gavgpool_kernel<DT_BF16>, g, dim3(256), 0, s, x, y, n, hw, c),
              hipLaunchKernelGGL(gavgpool_kernel<DT_F16>, g, dim3(256), 0, s, x, y, n, hw, c));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_bn_fold(const float* g, const float* b, const float* mean, const float* var,
                       float eps, float* scale, float* bias, int c, hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((c + 255) / 256), dim3(256), 0, s, g, b, mean, var, eps,
                     scale, bias, c);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void scale_inplace_kernel(float* __restrict__ x, float f, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= f;
}
int spk_launch_scale_inplace(float* x, float f, int n, hipStream_t s) {
  hipLaunchKernelGGL(scale_inplace_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, f, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_pack_weights(const float* w_krsc, bf16_t* out, int cout, int kh, int kw, int cin,
                            int mode, int dt, int splitw, hipStream_t s, float stem_wscale) {
  if (mode == CONV_MODE_STEM) {
    if (kh > 7 || kw > 7 || cin > 4) return -1;
    const int total = cout * 256;
    const int g = (total + 255) / 256;
    DT_DISPATCH(dt,
                hipLaunchKernelGGL(pack_stem_kernel<DT_BF16>, dim3(g), dim3(256), 0, s, w_krsc, out, cout, kh, kw, cin, splitw, stem_wscale),
                hipLaunchKernelGGL(pack_stem_kernel<DT_F16>, dim3(g), dim3(256), 0, s, w_krsc, out, cout, kh, kw, cin, splitw, stem_wscale));
  } else {
    const size_t n = (size_t)cout * kh * kw * cin;
    const int g = grid_for(n, 256);
    DT_DISPATCH(dt,
                hipLaunchKernelGGL(pack_generic_kernel<DT_BF16>, dim3(g), dim3(256), 0, s, w_krsc, out, n, splitw),
                hipLaunchKernelGGL(pack_generic_kernel<DT_F16>, dim3(g), dim3(256), 0, s, w_krsc, out, n, splitw));
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
