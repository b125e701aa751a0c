// ROI preprocessing on the GPU, straight from the bytes of an IFCB `.roi` blob:
// modal grey level -> aspect-preserving bilinear resize -> centred constant
// border -> [n, H, W, 3] uint8 (three identical channels), the tensor the
// reference's DataLoader workers build one PNG at a time
// (sykepic/utils/ifcb.py:76-118 -> sykepic/train/data.py:210-231 ->
// sykepic/train/image.py:25-56,183-237).  SURVEY.md §8f rank 1: once the
// forward runs at >3e4 img/s the per-ROI host pipeline is the bottleneck.
//
// Arithmetic replicates sykepic_hip/preprocess.py (the host restatement of
// OpenCV 4.5.5's 8-bit INTER_LINEAR cv::resize: float32 source coordinates,
// 11-bit fixed-point coefficients, the (b*(S>>4))>>16 vertical pass, the
// exact-2x INTER_AREA shortcut; see resize_u8.h) bit for bit — tests compare
// the two byte-wise.
// One workgroup per ROI: LDS histogram for the mode, then 4 output pixels
// (12 B = three dword stores) per thread per step.
#include "../../include/sykepic_hip.h"
#include "spk_common.h"
#include "resize_u8.h"

namespace {

__global__ __launch_bounds__(256) void roi_preprocess_kernel(const unsigned char* __restrict__ blob,
                                                             long long blob_bytes,
                                                             const spk_roi* __restrict__ rois, int out_h,
                                                             int out_w, int border,
                                                             unsigned char* __restrict__ out) {
  __shared__ int hist[256];
  __shared__ int s_mode;
  const spk_roi roi = rois[blockIdx.x];
  const int w = roi.width, h = roi.height;
  const unsigned char* src = blob + roi.offset;
  const bool ok = w > 0 && h > 0 && roi.offset >= 0 && roi.offset + (long long)w * h <= blob_bytes;

  int grey = border;
  if (border < 0) {  // "mode": most common value, lowest on ties
    hist[threadIdx.x] = 0;
    __syncthreads();
    if (ok)
      for (int i = threadIdx.x; i < w * h; i += 256) atomicAdd(&hist[src[i]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      int best = 0;
      for (int v = 1; v < 256; ++v)
        if (hist[v] > hist[best]) best = v;
      s_mode = best;
    }
    __syncthreads();
    grey = s_mode;
  }

  // aspect-preserving size (image.py:183-198): same double arithmetic as the host
  int new_h, new_w;
  if (!ok) {
    new_h = new_w = 0;
  } else if (h > w) {
    new_h = out_h;
    new_w = (int)((double)w * ((double)out_h / (double)h));
  } else {
    new_h = (int)((double)h * ((double)out_w / (double)w));
    new_w = out_w;
  }
  if (ok) { new_h = max(new_h, 1); new_w = max(new_w, 1); }
  const int top = max(out_h - new_h, 0) / 2, left = max(out_w - new_w, 0) / 2;
  const bool identity = (h == new_h && w == new_w);
  const double scale_x = ok ? resize_scale(w, new_w) : 1.0, scale_y = ok ? resize_scale(h, new_h) : 1.0;
  const bool half = (scale_x == 2.0 && scale_y == 2.0);

  unsigned char* dst = out + (size_t)blockIdx.x * out_h * out_w * 3;
  const int quads = (out_h * out_w + 3) / 4;
  for (int qi = threadIdx.x; qi < quads; qi += 256) {
    unsigned char px[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = qi * 4 + e;
      const int oy = p / out_w, ox = p - oy * out_w;
      const int ry = oy - top, rx = ox - left;
      int v = grey;
      if (p < out_h * out_w && ry >= 0 && ry < new_h && rx >= 0 && rx < new_w) {
        if (identity) {
          v = src[ry * w + rx];
        } else if (half) {
          const unsigned char* q = src + (2 * ry) * w + 2 * rx;
          v = (q[0] + q[1] + q[w] + q[w + 1] + 2) >> 2;
        } else {
          const Axis ax = coeff<false>(rx, w, scale_x), ay = coeff<true>(ry, h, scale_y);
          const int r0 = src[ay.i0 * w + ax.i0] * ax.a0 + src[ay.i0 * w + ax.i1] * ax.a1;
          const int r1 = src[ay.i1 * w + ax.i0] * ax.a0 + src[ay.i1 * w + ax.i1] * ax.a1;
          v = (((ay.a0 * (r0 >> 4)) >> 16) + ((ay.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
          v = min(max(v, 0), 255);
        }
      }
      px[e] = (unsigned char)v;
    }
    const int p0 = qi * 4;
    if (p0 + 3 < out_h * out_w) {
      // 4 pixels x 3 identical channels = 12 bytes, dword aligned (p0 % 4 == 0)
      unsigned int* o32 = (unsigned int*)(dst + (size_t)p0 * 3);
      o32[0] = px[0] | (px[0] << 8) | (px[0] << 16) | ((unsigned)px[1] << 24);
      o32[1] = px[1] | (px[1] << 8) | (px[2] << 16) | ((unsigned)px[2] << 24);
      o32[2] = px[2] | (px[3] << 8) | (px[3] << 16) | ((unsigned)px[3] << 24);
    } else {
      for (int e = 0; e < 4 && p0 + e < out_h * out_w; ++e)
        for (int c = 0; c < 3; ++c) dst[(size_t)(p0 + e) * 3 + c] = px[e];
    }
  }
}

}  // namespace

void spk_set_error(const std::string& s);

extern "C" int spk_preprocess_rois(const unsigned char* blob_dev, int64_t blob_bytes, const spk_roi* rois_dev,
                                   int n, int out_h, int out_w, int border, unsigned char* out_dev,
                                   void* stream) {
  if (!blob_dev || !rois_dev || !out_dev || n < 0 || out_h <= 0 || out_w <= 0 || border > 255) {
    spk_set_error("spk_preprocess_rois: bad arguments");
    return SPK_ERR_ARG;
  }
  if (n == 0) return SPK_OK;
  if (((size_t)out_h * out_w * 3) % 4) {
    spk_set_error("spk_preprocess_rois: out_h*out_w*3 must be a multiple of 4 bytes");
    return SPK_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(roi_preprocess_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, blob_dev,
                     (long long)blob_bytes, rois_dev, out_h, out_w, border, out_dev);
  if (hipGetLastError() != hipSuccess) {
    spk_set_error("spk_preprocess_rois: launch failed");
    return SPK_ERR_HIP;
  }
  return SPK_OK;
}

int spk_launch_predict(const float* p, int n, int c, const float* thr, float scalar_thr, int* pred,
                       unsigned char* ok, hipStream_t s);

extern "C" int spk_predict_rows(const float* probs_dev, int n, int num_classes, const float* thresholds_dev,
                                float scalar_threshold, int32_t* pred_dev, unsigned char* classified_dev,
                                void* stream) {
  if (!probs_dev || !pred_dev || !classified_dev || n < 0 || num_classes <= 0) {
    spk_set_error("spk_predict_rows: bad arguments");
    return SPK_ERR_ARG;
  }
  if (n == 0) return SPK_OK;
  if (spk_launch_predict(probs_dev, n, num_classes, thresholds_dev, scalar_threshold, pred_dev, classified_dev,
                         (hipStream_t)stream)) {
    spk_set_error("spk_predict_rows: launch failed");
    return SPK_ERR_HIP;
  }
  return SPK_OK;
}
