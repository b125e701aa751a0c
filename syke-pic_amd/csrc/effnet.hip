// EfficientNet (torchvision MBConv) pieces of the eval path that are not GEMM-shaped:
// the 3x3/2 RGB stem, depthwise k3/k5 convolutions, squeeze-excitation, and the
// zero-padded packing that lets the implicit-GEMM kernel (64-channel granularity) run
// the 1x1 expand / project / head convolutions of a network whose widths are
// multiples of 8 only (expanded C in {48,144,192,336,672,960,1632}).
// Reference: the torchvision backbone reached through sykepic/train/network.py:48;
// topology restated in oracle/backbones.py (parameter counts of B0-B4 match the
// published ones).  Every activation tensor carries its channels padded to a multiple
// of 64; padded channels are exactly zero everywhere (zero weights, zero BN scale/shift,
// SiLU(0) = 0, SE scale 0), so they never contribute.
// All of these are HBM-bound passes: 16-B per-lane accesses, channels innermost.
#include "spk_common.h"

namespace {

template <int DT>
__device__ __forceinline__ void unpack8f(const u32x4_t v, float* f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = lo_f32<DT>(v[j]);
    f[2 * j + 1] = hi_f32<DT>(v[j]);
  }
}
template <int DT>
__device__ __forceinline__ u32x4_t pack8f(const float* f) {
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack2<DT>(f[2 * j], f[2 * j + 1]);
  return o;
}
__device__ __forceinline__ float act_f(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return v / (1.f + __expf(-v));  // SiLU
  return v;
}

// [cout][taps][cin] fp32 -> [cout_p][taps][cin_p] 16-bit (hi, then lo halves), zero padded
template <int DT>
__global__ void pack_padded_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int taps,
                                   int cin, int cout_p, int cin_p, int splitw) {
  const size_t n = (size_t)cout_p * taps * cin_p;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin_p);
    const size_t r = i / cin_p;
    const int t = (int)(r % taps), co = (int)(r / taps);
    const float v = (co < cout && ci < cin) ? w[((size_t)co * taps + t) * cin + ci] : 0.f;
    const unsigned short hi = to_h16<DT>(v);
    out[i] = hi;
    if (splitw) out[n + i] = to_h16<DT>(v - lo_f32<DT>((unsigned int)hi));
  }
}

// depthwise / stem weights: [c][taps*cin1] fp32 -> [taps*cin1][c_p] fp32 (channel innermost), zero padded
__global__ void pack_tapmajor_kernel(const float* __restrict__ w, float* __restrict__ out, int c, int rows,
                                     int c_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c_p) return;
  const int ch = i % c_p, r = i / c_p;
  out[i] = ch < c ? w[(size_t)ch * rows + r] : 0.f;
}

// 3x3 stride-2 pad-1 stem on the NHWC4 input image (RGB + zero channel): one thread = 8 output channels of
// one output pixel; weights [9 taps][4 ch][c_p] fp32 (exact fp32 products of the 16-bit image).
template <int DT>
__global__ void stem3x3_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                               const float* __restrict__ scale, const float* __restrict__ bias,
                               bf16_t* __restrict__ y, int n, int h, int wid, int wstride, int ho, int wo, int c_p,
                               int act) {
  const int c8 = c_p >> 3;
  const size_t total = (size_t)n * ho * wo * c8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c8);
    size_t p = i / c8;
    const int ox = (int)(p % wo);
    p /= wo;
    const int oy = (int)(p % ho), img = (int)(p / ho);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = 2 * oy - 1 + r;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ix = 2 * ox - 1 + s;
        if ((unsigned)ix >= (unsigned)wid) continue;
        const uint2 px = *(const uint2*)(x + (((size_t)img * h + iy) * wstride + ix) * 4);
        const float ch[3] = {lo_f32<DT>(px.x), hi_f32<DT>(px.x), lo_f32<DT>(px.y)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* wp = w + (size_t)((r * 3 + s) * 4 + c) * c_p + cg * 8;
          const f32x4_t w0 = *(const f32x4_t*)wp, w1 = *(const f32x4_t*)(wp + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[j] += ch[c] * w0[j];
            acc[4 + j] += ch[c] * w1[j];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = act_f(acc[j] * scale[cg * 8 + j] + bias[cg * 8 + j], act);
    *(u32x4_t*)(y + i * 8) = pack8f<DT>(acc);
  }
}

// depthwise KxK conv + folded BN + activation; one thread = 8 channels of one output pixel
template <int DT, int K>
__global__ void dwconv_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                              const float* __restrict__ scale, const float* __restrict__ bias,
                              bf16_t* __restrict__ y, int n, int h, int wid, int c_p, int ho, int wo, int stride,
                              int act) {
  constexpr int PAD = (K - 1) / 2;
  const int c8 = c_p >> 3;
  const size_t total = (size_t)n * ho * wo * c8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c8);
    size_t p = i / c8;
    const int ox = (int)(p % wo);
    p /= wo;
    const int oy = (int)(p % ho), img = (int)(p / ho);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const int iy = oy * stride - PAD + r;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int s = 0; s < K; ++s) {
        const int ix = ox * stride - PAD + s;
        if ((unsigned)ix >= (unsigned)wid) continue;
        float xv[8];
        unpack8f<DT>(*(const u32x4_t*)(x + (((size_t)img * h + iy) * wid + ix) * c_p + cg * 8), xv);
        const float* wp = w + (size_t)(r * K + s) * c_p + cg * 8;
        const f32x4_t w0 = *(const f32x4_t*)wp, w1 = *(const f32x4_t*)(wp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] += xv[j] * w0[j];
          acc[4 + j] += xv[4 + j] * w1[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = act_f(acc[j] * scale[cg * 8 + j] + bias[cg * 8 + j], act);
    *(u32x4_t*)(y + i * 8) = pack8f<DT>(acc);
  }
}

// squeeze: per-image, per-channel sums of a pixel chunk -> partial[img][chunk][c_p] (ordered two-stage sum)
template <int DT>
__global__ __launch_bounds__(256) void se_pool_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                      int hw, int c_p, int chunks) {
  extern __shared__ float sm[];  // [rows in flight][c_p]
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int c8 = c_p >> 3;
  const int tpr = c8 < 256 ? c8 : 256, rif = 256 / tpr;
  const int lane_c = threadIdx.x % tpr, lane_r = threadIdx.x / tpr;
  const int per = (hw + chunks - 1) / chunks;
  const int p0 = chunk * per, p1 = min(hw, p0 + per);
  if (lane_r < rif) {
    for (int cg = lane_c; cg < c8; cg += tpr) {
      float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int p = p0 + lane_r; p < p1; p += rif) {
        float v[8];
        unpack8f<DT>(*(const u32x4_t*)(x + ((size_t)img * hw + p) * c_p + cg * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += v[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sm[lane_r * c_p + cg * 8 + j] = s[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < c_p; c += 256) {
    float t = 0.f;
    for (int r = 0; r < rif; ++r) t += sm[r * c_p + c];
    partial[((size_t)img * chunks + chunk) * c_p + c] = t;
  }
}

// excitation: s = sigmoid(W2 silu(W1 avg + b1) + b2); one block per image; scale[img][c_p] (0 on padding)
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ partial, int chunks, float inv_hw,
                                                    const float* __restrict__ w1, const float* __restrict__ b1,
                                                    const float* __restrict__ w2, const float* __restrict__ b2,
                                                    float* __restrict__ scale, int c, int c_p, int sq) {
  extern __shared__ float sm[];  // avg[c_p], hid[sq]
  float* avg = sm;
  float* hid = sm + c_p;
  const int img = blockIdx.x;
  for (int i = threadIdx.x; i < c_p; i += 256) {
    float t = 0.f;
    for (int k = 0; k < chunks; ++k) t += partial[((size_t)img * chunks + k) * c_p + i];
    avg[i] = t * inv_hw;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < sq; j += 4) {  // one wave per hidden unit: lanes stride the channels
    float t = 0.f;
    for (int i = lane; i < c; i += 64) t += w1[(size_t)j * c + i] * avg[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) {
      t += b1[j];
      hid[j] = t / (1.f + __expf(-t));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c_p; i += 256) {
    float s = 0.f;
    if (i < c) {
      float t = b2[i];
      for (int j = 0; j < sq; ++j) t += w2[(size_t)i * sq + j] * hid[j];
      s = 1.f / (1.f + __expf(-t));
    }
    scale[(size_t)img * c_p + i] = s;
  }
}

template <int DT>
__global__ void se_scale_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                bf16_t* __restrict__ y, size_t total8, int hw, int c_p) {
  const int c8 = c_p >> 3;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total8; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c8);
    const size_t img = i / c8 / hw;
    float v[8];
    unpack8f<DT>(*(const u32x4_t*)(x + i * 8), v);
    const float* s = scale + img * c_p + cg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= s[j];
    *(u32x4_t*)(y + i * 8) = pack8f<DT>(v);
  }
}

inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  if (g > 256 * 8 * 4) g = 256 * 8 * 4;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

#define DT_DISPATCH(dt, CALL_BF16, CALL_F16) \
  do { if ((dt) == DT_F16) { CALL_F16; } else { CALL_BF16; } } while (0)

int spk_launch_pack_padded(const float* w, bf16_t* out, int cout, int taps, int cin, int cout_p, int cin_p, int dt,
                           int splitw, hipStream_t s) {
  const int g = grid_for((size_t)cout_p * taps * cin_p, 256);
  DT_DISPATCH(dt,
              hipLaunchKernelGGL(pack_padded_kernel<DT_BF16>, dim3(g), dim3(256), 0, s, w, out, cout, taps, cin, cout_p, cin_p, splitw),
              hipLaunchKernelGGL(pack_padded_kernel<DT_F16>, dim3(g), dim3(256), 0, s, w, out, cout, taps, cin, cout_p, cin_p, splitw));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_pack_tapmajor(const float* w, float* out, int c, int rows, int c_p, hipStream_t s) {
  hipLaunchKernelGGL(pack_tapmajor_kernel, dim3((rows * c_p + 255) / 256), dim3(256), 0, s, w, out, c, rows, c_p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_stem3x3(const bf16_t* x, const float* w, const float* scale, const float* bias, bf16_t* y, int n, int h,
                       int wid, int wstride, int ho, int wo, int c_p, int act, int dt, hipStream_t s) {
  const int g = grid_for((size_t)n * ho * wo * (c_p / 8), 256);
  DT_DISPATCH(dt,
              hipLaunchKernelGGL(stem3x3_kernel<DT_BF16>, dim3(g), dim3(256), 0, s, x, w, scale, bias, y, n, h, wid, wstride, ho, wo, c_p, act),
              hipLaunchKernelGGL(stem3x3_kernel<DT_F16>, dim3(g), dim3(256), 0, s, x, w, scale, bias, y, n, h, wid, wstride, ho, wo, c_p, act));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_dwconv(const bf16_t* x, const float* w, const float* scale, const float* bias, bf16_t* y, int n, int h,
                      int wid, int c_p, int ho, int wo, int k, int stride, int act, int dt, hipStream_t s) {
  if ((k != 3 && k != 5) || dt != DT_F16) return -2;
  const int g = grid_for((size_t)n * ho * wo * (c_p / 8), 256);
  if (k == 3)
    hipLaunchKernelGGL((dwconv_kernel<DT_F16, 3>), dim3(g), dim3(256), 0, s, x, w, scale, bias, y, n, h, wid, c_p, ho, wo, stride, act);
  else
    hipLaunchKernelGGL((dwconv_kernel<DT_F16, 5>), dim3(g), dim3(256), 0, s, x, w, scale, bias, y, n, h, wid, c_p, ho, wo, stride, act);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// chunks of the squeeze pass for a given image size (same value at planning and at launch)
int spk_se_chunks(int hw) { return hw >= 4096 ? 16 : (hw >= 512 ? 4 : 1); }

int spk_launch_se(const bf16_t* x, bf16_t* y, float* partial, float* scale, const float* w1, const float* b1,
                  const float* w2, const float* b2, int n, int hw, int c, int c_p, int sq, int dt, hipStream_t s) {
  if (dt != DT_F16) return -2;
  const int chunks = spk_se_chunks(hw);
  const int c8 = c_p / 8, tpr = c8 < 256 ? c8 : 256, rif = 256 / tpr;
  hipLaunchKernelGGL(se_pool_kernel<DT_F16>, dim3(chunks, n), dim3(256), (size_t)rif * c_p * 4, s, x, partial, hw, c_p, chunks);
  hipLaunchKernelGGL(se_fc_kernel, dim3(n), dim3(256), (size_t)(c_p + sq) * 4, s, partial, chunks, 1.0f / (float)hw, w1, b1,
                     w2, b2, scale, c, c_p, sq);
  const size_t total8 = (size_t)n * hw * c8;
  hipLaunchKernelGGL(se_scale_kernel<DT_F16>, dim3(grid_for(total8, 256)), dim3(256), 0, s, x, scale, y, total8, hw, c_p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
