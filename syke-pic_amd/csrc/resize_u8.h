// 8-bit bilinear resize / affine warp arithmetic shared by the ROI preprocessing and the augmentation kernels.
// Restates OpenCV 4.5.5 (the reference's pinned cv2) exactly as sykepic_hip/preprocess.py does on the host:
//   cv::resize INTER_LINEAR, uint8   modules/imgproc/src/resize.cpp
//     fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx            (scale double, fx FLOAT)
//     x axis: sx < 0 -> fx = 0, sx = 0;  sx >= w-1 -> fx = 0, sx = w-1;  y axis: rows clamped, weights kept
//     ialpha = saturate_cast<short>((1.f - fx) * 2048), saturate_cast<short>(fx * 2048)   (cvRound: half to even)
//     dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2,  S = s[x0]*a0 + s[x1]*a1
//     scale == 2 on both axes: INTER_AREA fast path, (sum of the 2x2 box + 2) >> 2
//   cv::warpAffine INTER_LINEAR + BORDER_CONSTANT, uint8   modules/imgproc/src/imgwarp.cpp  (warp_affine_u8_at)
// No FMA contraction: the host (numpy) rounds every product and sum separately.
#pragma once
#include <hip/hip_runtime.h>

struct Axis {
  int i0, i1, a0, a1;
};

// scale: 1/inv_scale as cv::resize computes it (1.0 / ((double)dst / src) when dsize is given; 1/fx when the
// caller passes fx).  YAXIS: clamp the rows only.
template <bool YAXIS>
__device__ __forceinline__ Axis coeff(int d, int src, double scale) {
#pragma clang fp contract(off)
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  Axis r;
  if (YAXIS) {
    r.i0 = min(max(s, 0), src - 1);
    r.i1 = min(max(s + 1, 0), src - 1);
  } else {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    r.i0 = s;
    r.i1 = min(s + 1, src - 1);
  }
  r.a0 = (int)rintf((1.f - f) * 2048.f);
  r.a1 = (int)rintf(f * 2048.f);
  return r;
}

__device__ __forceinline__ double resize_scale(int src, int dst) {
  return 1.0 / ((double)dst / (double)src);
}

// one resized sample of a [h][w] single-channel plane with pixel stride `ps` bytes
__device__ __forceinline__ int resize_u8_at(const unsigned char* src, int ps, int w, int h, int new_w, int new_h,
                                            double scale_x, double scale_y, int rx, int ry) {
  if (h == new_h && w == new_w && scale_x == 1.0 && scale_y == 1.0) return src[((size_t)ry * w + rx) * ps];
  if (scale_x == 2.0 && scale_y == 2.0) {  // OpenCV's exact-2x INTER_AREA shortcut
    const int x0 = 2 * rx, y0 = 2 * ry, x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    return (src[((size_t)y0 * w + x0) * ps] + src[((size_t)y0 * w + x1) * ps] + src[((size_t)y1 * w + x0) * ps] +
            src[((size_t)y1 * w + x1) * ps] + 2) >> 2;
  }
  const Axis ax = coeff<false>(rx, w, scale_x), ay = coeff<true>(ry, h, scale_y);
  const int r0 = src[((size_t)ay.i0 * w + ax.i0) * ps] * ax.a0 + src[((size_t)ay.i0 * w + ax.i1) * ps] * ax.a1;
  const int r1 = src[((size_t)ay.i1 * w + ax.i0) * ps] * ax.a0 + src[((size_t)ay.i1 * w + ax.i1) * ps] * ax.a1;
  const int v = (((ay.a0 * (r0 >> 4)) >> 16) + ((ay.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
  return min(max(v, 0), 255);
}

// cv::warpAffine, INTER_LINEAR, BORDER_CONSTANT, one uint8 sample.  M[0..5]: the INVERTED 2x3 matrix exactly as
// cv::warpAffine derives it from the caller's forward matrix (the host does that in double: preprocess.invert_affine).
//   adelta = saturate_cast<int>(M[0]*x*1024), bdelta = saturate_cast<int>(M[3]*x*1024)          (AB_BITS 10)
//   X0 = saturate_cast<int>((M[1]*y + M[2])*1024) + 16,  Y0 likewise with M[4], M[5]            (round_delta)
//   X = (X0 + adelta) >> 5;  sx = X >> 5, fx = X & 31 (INTER_BITS 5);  likewise Y
//   weights = BilinearTab_i[fy][fx] = 32*(32-fy)*(32-fx), 32*(32-fy)*fx, 32*fy*(32-fx), 32*fy*fx  (15 bits; the
//   entry (0,0) is 32767,0,0,1: its 32768 saturates in a short and the table's sum repair lands on tap (1,1))
//   dst = (sum tap*weight + (1 << 14)) >> 15, taps outside the image = the border value
__device__ __forceinline__ int warp_affine_u8_at(const unsigned char* img, int h, int w, int c, int ch, const double* M,
                                                 int x, int y, int border) {
#pragma clang fp contract(off)
  const int adelta = (int)rint(M[0] * (double)x * 1024.0);
  const int bdelta = (int)rint(M[3] * (double)x * 1024.0);
  const int X0 = (int)rint((M[1] * (double)y + M[2]) * 1024.0) + 16;
  const int Y0 = (int)rint((M[4] * (double)y + M[5]) * 1024.0) + 16;
  const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
  const int fx = X & 31, fy = Y & 31;
  int w00 = 32 * (32 - fy) * (32 - fx), w01 = 32 * (32 - fy) * fx, w10 = 32 * fy * (32 - fx), w11 = 32 * fy * fx;
  if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }
  auto tap = [&](int yy, int xx) -> int {
    return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? img[((size_t)yy * w + xx) * c + ch] : border;
  };
  const int acc = tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11;
  return min(max((acc + (1 << 14)) >> 15, 0), 255);
}
