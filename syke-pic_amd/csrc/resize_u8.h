// 8-bit bilinear resize arithmetic shared by the ROI preprocessing and the augmentation kernels: the host
// restatement of OpenCV's INTER_LINEAR uint8 path (sykepic_hip/preprocess.py:_coeffs / resize_linear_u8):
// 11-bit fixed-point coefficients computed in double without FMA contraction.
#pragma once
#include <hip/hip_runtime.h>

struct Axis {
  int i0, i1, a0, a1;
};

__device__ __forceinline__ Axis coeff(int d, int src, int dst) {
#pragma clang fp contract(off)  // no FMA fusion: the host computes mul, then sub
  const double scale = (double)src / (double)dst;
  double f = ((double)d + 0.5) * scale - 0.5;
  int s = (int)floor(f);
  f -= (double)s;
  if (s < 0) { f = 0.0; s = 0; }
  if (s >= src - 1) { f = 0.0; s = src - 1; }
  Axis r;
  r.i0 = s;
  r.i1 = min(s + 1, src - 1);
  r.a1 = (int)rint(f * 2048.0);
  r.a0 = (int)rint((1.0 - f) * 2048.0);
  return r;
}

// one resized sample of a [h][w] single-channel plane with pixel stride `ps` bytes
__device__ __forceinline__ int resize_u8_at(const unsigned char* src, int ps, int w, int h, int new_w, int new_h,
                                            int rx, int ry) {
  if (h == new_h && w == new_w) return src[((size_t)ry * w + rx) * ps];
  if (w == 2 * new_w && h == 2 * new_h) {  // OpenCV's exact-2x INTER_AREA shortcut
    const unsigned char* q = src + ((size_t)(2 * ry) * w + 2 * rx) * ps;
    return (q[0] + q[ps] + q[(size_t)w * ps] + q[(size_t)(w + 1) * ps] + 2) >> 2;
  }
  const Axis ax = coeff(rx, w, new_w), ay = coeff(ry, h, new_h);
  const int r0 = src[((size_t)ay.i0 * w + ax.i0) * ps] * ax.a0 + src[((size_t)ay.i0 * w + ax.i1) * ps] * ax.a1;
  const int r1 = src[((size_t)ay.i1 * w + ax.i0) * ps] * ax.a0 + src[((size_t)ay.i1 * w + ax.i1) * ps] * ax.a1;
  const int v = (((ay.a0 * (r0 >> 4)) >> 16) + ((ay.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
  return min(max(v, 0), 255);
}
