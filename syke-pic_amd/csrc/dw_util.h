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

}  // namespace dwu
