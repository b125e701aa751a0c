// Shared device helpers of the LDS-staged depthwise kernels (dwconv_lds.hip): the 16-byte channel
// group of a thread (8 fp16 or 16 e4m3 channels) to / from fp32.
#pragma once
#include "spk_common.h"

namespace dwu {

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int ET> struct DwT;
template <> struct DwT<0> { static constexpr int CPT = 8, PX = 4, ELEM = 2; };
template <> struct DwT<1> { static constexpr int CPT = 16, PX = 2, ELEM = 1; };

template <int ET>
__device__ __forceinline__ void unpack16B(const u32x4_t v, float* f) {
  if (ET == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = lo_f32<DT_F16>(v[j]); f[2 * j + 1] = hi_f32<DT_F16>(v[j]); }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[j], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[j], true);
      f[4 * j] = lo[0]; f[4 * j + 1] = lo[1]; f[4 * j + 2] = hi[0]; f[4 * j + 3] = hi[1];
    }
  }
}
template <int ET>
__device__ __forceinline__ u32x4_t pack16B(const float* f, float s) {
  u32x4_t o;
  if (ET == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack2<DT_F16>(f[2 * j], f[2 * j + 1]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = fminf(fmaxf(f[4 * j] * s, -448.f), 448.f), b = fminf(fmaxf(f[4 * j + 1] * s, -448.f), 448.f);
      float c = fminf(fmaxf(f[4 * j + 2] * s, -448.f), 448.f), d = fminf(fmaxf(f[4 * j + 3] * s, -448.f), 448.f);
      int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
      o[j] = (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    }
  }
  return o;
}

// Tap weights, folded-BN scales and shifts of one channel tile -> LDS, for every depthwise kernel.
// LDS layout [row][4-channel part][channel group][4] (row = tap, then scale, then shift): the lanes of a wave read
// consecutive 16-byte slots (the plain [row][channel] layout put a lane's 8 / 16 floats 32 / 64 B from its
// neighbour's: 2- / 4-way bank conflicts on every weight read, 13-15 conflict cycles per LDS instruction in the e4m3
// gather kernel).  All of a thread's 16-byte loads are issued before its first LDS store: as a plain
// `for (i = tid; ...; i += 256) lds[f(i)] = global[g(i)]` loop the (K*K + 2) * 256 / 256 = 11 ... 27 iterations were a
// chain of dependent L2 round trips at the start of EVERY block (~13 us for k5; a 7x7 or 14x14 layer has ~1800 blocks
// of ~1 us of work each).
// groups: channel groups of the tile that exist (ncg); gstride: group stride of the LDS layout (ncg, or the plan's
// tile width when the kernel addresses with that); channels of groups >= `groups` are stored as zeros.
template <int K, int CPT>
__device__ __forceinline__ void stage_dw_weights(float* __restrict__ lds, const float* __restrict__ w,
                                                 const float* __restrict__ scale, const float* __restrict__ bias, int c_p,
                                                 int ch0, int groups, int gstride, float w_scale, int tid) {
  constexpr int ROWS = K * K + 2;
  constexpr int WV = (ROWS * 256 / 4 + 255) / 256;    // float4 per thread for a full 256-channel tile
  const int tcs = gstride * CPT;                      // channels the layout holds per row
  const int total4 = ROWS * tcs / 4;
  f32x4_t v[WV];
  int slot[WV];
#pragma unroll
  for (int i = 0; i < WV; ++i) {
    const int j = min(tid + i * 256, total4 - 1) * 4;
    const int r = j / tcs, c = j - r * tcs;
    const bool ok = c < groups * CPT;
    const float* src = r < K * K ? w + (size_t)r * c_p : (r == K * K ? scale : bias);
    v[i] = *(const f32x4_t*)(src + ch0 + (ok ? c : 0));
    v[i] *= ok ? (r < K * K ? w_scale : 1.f) : 0.f;
    slot[i] = ((r * (CPT / 4) + ((c % CPT) >> 2)) * gstride + c / CPT) << 2;
  }
#pragma unroll
  for (int i = 0; i < WV; ++i)
    if (tid + i * 256 < total4) *(f32x4_t*)(lds + slot[i]) = v[i];
}

}  // namespace dwu
