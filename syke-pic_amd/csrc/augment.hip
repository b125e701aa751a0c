// Training augmentations on the GPU: the per-sample random transforms the
// reference applies with cv2 in its DataLoader workers (sykepic/train/image.py:80-180 through
// Compose.__call__, image.py:25-56) - flip H/V, translate, zoom, rotate, brightness - on a batch of
// already resized+bordered uint8 NHWC images.  The random draws stay on the host (Python's `random`, in the
// reference's call order, so a seeded run augments exactly like the host pipeline); the kernels apply them.
// Arithmetic replicates sykepic_hip/preprocess.py byte for byte (integer / double without FMA contraction);
// every op rounds to uint8 like the host does between transforms.  One launch per op over the whole batch.
#include "../../include/sykepic_hip.h"
#include "resize_u8.h"

#include <string>

void spk_set_error(const std::string& s);

namespace {

__device__ __forceinline__ int tap_u8(const unsigned char* img, int h, int w, int c, int ch, long long yy,
                                      long long xx, int border) {
  return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? img[((size_t)yy * w + xx) * c + ch] : border;
}

// grid (pixel chunks, n); one thread = one pixel, all channels
__global__ __launch_bounds__(256) void augment_kernel(const unsigned char* __restrict__ in,
                                                      unsigned char* __restrict__ out, int h, int w, int c,
                                                      const spk_aug_op* __restrict__ ops,
                                                      const unsigned char* __restrict__ border) {
#pragma clang fp contract(off)  // numpy evaluates a*x + b*y + c with separate roundings
  const int img = blockIdx.y;
  const spk_aug_op op = ops[img];
  const unsigned char* src = in + (size_t)img * h * w * c;
  unsigned char* dst = out + (size_t)img * h * w * c;
  const unsigned char* bv = border + (size_t)img * 4;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < h * w; p += gridDim.x * blockDim.x) {
    const int y = p / w, x = p - y * w;
    for (int ch = 0; ch < c; ++ch) {
      int v;
      switch (op.kind) {
        case SPK_AUG_FLIP_H: v = src[((size_t)y * w + (op.i0 ? w - 1 - x : x)) * c + ch]; break;
        case SPK_AUG_FLIP_V: v = src[((size_t)(op.i0 ? h - 1 - y : y) * w + x) * c + ch]; break;
        case SPK_AUG_TRANSLATE:  // dst(x, y) = src(x - tx, y - ty), constant border
          // cv2.warpAffine with an integer shift lands on table entry (0,0): weights 32767 on the pixel and 1 on its
          // lower-right neighbour -> (32767 a + b + 16384) >> 15 = a for 8-bit a, b: an exact copy
          v = tap_u8(src, h, w, c, ch, (long long)y - op.i1, (long long)x - op.i0, bv[ch]);
          break;
        case SPK_AUG_ZOOM: {  // cv2.resize(img, None, fx=f, fy=f) to i0 x i0 (source coordinates with scale 1/f),
                              // then centred pad (f < 1) or crop; square images
          const int z = op.i0;
          const double sc = 1.0 / op.d[0];
          if (op.d[0] < 1.0) {
            const int p1 = (w - z) / 2;  // int((w - zw) / 2), non-negative
            const int ry = y - p1, rx = x - p1;
            v = (ry >= 0 && ry < z && rx >= 0 && rx < z) ? resize_u8_at(src + ch, c, w, h, z, z, sc, sc, rx, ry) : bv[ch];
          } else {
            const int c1 = (z - w) / 2;  // int((zw - w) / 2)
            v = resize_u8_at(src + ch, c, w, h, z, z, sc, sc, x + c1, y + c1);
          }
          break;
        }
        case SPK_AUG_ROTATE:  // cv2.warpAffine through the inverted matrix d[0..5] (resize_u8.h), constant border
          v = warp_affine_u8_at(src, h, w, c, ch, op.d, x, y, bv[ch]);
          break;
        case SPK_AUG_BRIGHT: {  // (img * v).clip(0, 255).astype(uint8): truncation
          const double o = (double)src[((size_t)y * w + x) * c + ch] * op.d[0];
          v = (int)fmin(fmax(o, 0.0), 255.0);
          break;
        }
        default: v = src[((size_t)y * w + x) * c + ch]; break;
      }
      dst[((size_t)y * w + x) * c + ch] = (unsigned char)v;
    }
  }
}

}  // namespace

extern "C" int spk_augment_batch(const unsigned char* in_dev, unsigned char* out_dev, unsigned char* tmp_dev, int n,
                                 int h, int w, int c, const spk_aug_op* ops_dev, int n_ops,
                                 const unsigned char* border_dev, void* stream) {
  if (!in_dev || !out_dev || n < 0 || h <= 0 || w <= 0 || c < 1 || c > 4 || n_ops < 0 ||
      (n_ops > 0 && (!ops_dev || !border_dev)) || (n_ops > 1 && !tmp_dev)) {
    spk_set_error("spk_augment_batch: bad arguments");
    return SPK_ERR_ARG;
  }
  if (n == 0) return SPK_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)n * h * w * c;
  if (n_ops == 0) {
    if (hipMemcpyAsync(out_dev, in_dev, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      spk_set_error("spk_augment_batch: copy failed");
      return SPK_ERR_HIP;
    }
    return SPK_OK;
  }
  // ping-pong so that the last op lands in out_dev
  const unsigned char* src = in_dev;
  int gx = (h * w + 255) / 256;
  if (gx > 64) gx = 64;
  for (int j = 0; j < n_ops; ++j) {
    unsigned char* dst = ((n_ops - 1 - j) % 2 == 0) ? out_dev : tmp_dev;
    hipLaunchKernelGGL(augment_kernel, dim3(gx, n), dim3(256), 0, s, src, dst, h, w, c, ops_dev + (size_t)j * n,
                       border_dev);
    src = dst;
  }
  if (hipGetLastError() != hipSuccess) {
    spk_set_error("spk_augment_batch: launch failed");
    return SPK_ERR_HIP;
  }
  return SPK_OK;
}
