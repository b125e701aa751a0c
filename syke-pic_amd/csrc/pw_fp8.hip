// fp8 (OCP e4m3) pointwise convolutions of the EfficientNet MBConv blocks on gfx950 — BASELINE config 5
// ("EfficientNet-B4 fp8 inference ... CDNA4 fp8 MFMA").  Reference op: the 1x1 Conv2d + BatchNorm2d (+SiLU)
// (+residual) modules of torchvision's MBConv, reached through `net(x)` (sykepic/compute/probability.py:189,
// model built at sykepic/train/network.py:48).
//
// A 1x1 conv is the GEMM  Y[m][n] = sum_k X[m][k] W[n][k]  over the M = N*H*W pixels.  In fp8 mode the tensors
// INSIDE an MBConv block are stored as e4m3 bytes with one power-of-two-free float scale per tensor
// (value = byte * scale; scales from a calibration pass, spk_model_calibrate_fp8), the residual trunk stays fp16:
//   expand   X = trunk (fp16, converted to e4m3 in the A-tile loader), Y = expanded tensor (e4m3)
//   project  X = depthwise output (e4m3), multiplied by the squeeze-excitation gate of its image and channel in
//            the loader (the separate scale pass over the largest tensor of the block disappears),
//            Y = trunk (fp16, + shortcut)
// Weights are e4m3 with one scale per output channel; every activation/weight scale is folded into the
// per-channel epilogue factor, so the MFMA (v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulate) runs on raw bytes.
//
// Tile 128 x 64, K step 64 (64-B LDS rows), 4 waves (2 x 2; four waves stacked in M with 64-column row segments
// measured 5-9 % slower), two LDS stages, register staging (the loader
// converts / gates, which LDS-DMA cannot).  LDS granule (8 B) index XORed with 2*((row>>2)&3): the ds_read_b64
// fragment reads of 16 rows x 2 k-groups then touch 32 distinct 8-B slots (conflict-free), and a 16-B store
// stays one aligned 16-B store.  (A persistent variant - 3 resident blocks per CU walking the tiles, the loads of the
// next (tile, K step) issued before the MFMAs and the epilogue of the current one - measured 10-35 % SLOWER on every
// expand layer: many small independent blocks, 5-6 per CU, hide the load-convert-MFMA-store latency chain better than
// one-deep prefetch in fewer, fatter blocks.)  These layers are HBM-bound (K = 24 ... 2688): the point of fp8 here is the
// halved bytes of the expanded tensors, not the MFMA rate.
#include "dw_util.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ unsigned int cvt4_fp8(float a, float b, float c, float d) {
  // saturating: e4m3fn has no infinity, 448 is the largest finite value
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned int)v;
}
__device__ __forceinline__ void cvt4_f32(unsigned int v, float* f) {
  const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v, true);
  f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}

struct PwArgs {
  const void* x;        // [M][cin_s] fp16 or e4m3
  const unsigned char* w;  // [Npad][Kpad] e4m3
  void* y;              // [M][cout_s] fp16 or e4m3
  const bf16_t* res;    // [M][cout_s] fp16 or null
  const float* scale;   // [Npad] epilogue factor (BN scale x weight scale x activation scale)
  const float* bias;    // [Npad]
  const float* gate;    // [images][gate_stride] or null (squeeze-excitation scale of the A operand)
  int M, Kpad, Npad, cin_s, cout_s, hw, gate_stride, act;
  float a_inv_scale;    // fp16 A: x * a_inv_scale -> e4m3
  float y_inv_scale;    // e4m3 output: value * y_inv_scale -> e4m3
  unsigned int x_bytes;
  // fp16 output only: an e4m3 copy of the (fp16-rounded) output, [M][y8_stride] bytes = e4m3(value * y8_inv_scale), columns
  // past cout_s zero - the A operand of the NEXT block's expand conv, which then skips its fp16 -> e4m3 conversion (that
  // conversion, repeated for each of its 3 ... 42 N tiles, was 35-45 % of the expand kernels' time)
  unsigned char* y8;
  int y8_stride;
  float y8_inv_scale;
};

__device__ __forceinline__ int lds_byte(int row, int granule) {   // granule: 8-B unit within the 64-B row
  return row * 64 + ((granule ^ (((row >> 2) & 3) << 1)) << 3);
}

template <bool A_FP8, bool GATED, bool OUT_FP8>
__global__ __launch_bounds__(256) void pw_fp8_kernel(PwArgs a, int m_tiles, int n_tiles) {
  constexpr int BM = 128, BN = 64, BK = 64;
  constexpr int A_BYTES = BM * BK, B_BYTES = BN * BK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];
  unsigned char* const sA = smem;
  unsigned char* const sB = smem + 2 * A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves: 64 x 32 each
  const int frow = lane & 15, fq = lane >> 4;

  // XCD-aware bijective tile map, N tiles of one M tile adjacent (as conv_igemm.hip)
  const int ntiles = m_tiles * n_tiles;
  const int w = blockIdx.x, q8 = ntiles >> 3, r8 = ntiles & 7, xcd = w & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (w >> 3);
  const int m0 = (swz / n_tiles) * BM, n0 = (swz % n_tiles) * BN;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const int KT = a.Kpad / BK;

  // ---- loaders ----
  // A fp8: 128 rows x 4 chunks of 16 B: 2 per thread.  A fp16: 128 rows x 8 chunks of 16 B (8 channels): 4 per thread.
  constexpr int A_IT = A_FP8 ? 2 : 4;
  constexpr int A_CPR = A_FP8 ? 4 : 8;              // 16-B global chunks per row per K step
  constexpr int A_RPP = 256 / A_CPR;
  const int a_chunk = tid % A_CPR, a_row = tid / A_CPR;
  int a_img[A_IT];
  u32x4_t ra[A_IT];
  f32x4_t rg[GATED ? A_IT : 1][4];
  u32x4_t rb;
  const int b_chunk = tid & 3, b_row = tid >> 2;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + a_row + i * A_RPP;
    a_img[i] = GATED ? min(m, a.M - 1) / a.hw : 0;
  }
  auto issue = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int m = m0 + a_row + i * A_RPP;
      const int kc = k0 + a_chunk * (A_FP8 ? 16 : 8);      // first channel of this chunk
      const bool ok = m < a.M && kc < a.cin_s;
      const unsigned off = ok ? (unsigned)(((size_t)m * a.cin_s + kc) * (A_FP8 ? 1 : 2)) : 0x80000000u;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      if (GATED) {
        // the 16 gates of this chunk travel with it (one K step ahead of their use): loaded in stash() they cost a
        // full L2 round trip per K step on the critical path of the long-K project convs
        const float* g = a.gate + (size_t)a_img[i] * a.gate_stride + kc;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rg[i][q] = kc + q * 4 < a.gate_stride ? *(const f32x4_t*)(g + q * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    rb = *(const u32x4_t*)(a.w + (size_t)(n0 + b_row) * a.Kpad + k0 + b_chunk * 16);
  };
  auto stash = [&](int buf) {
    unsigned char* dA = sA + buf * A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = a_row + i * A_RPP;
      if (A_FP8) {
        u32x4_t v = ra[i];
        if (GATED) {
          // x * gate[image][channel], re-rounded to e4m3 (one rounding, as a separate scale pass would make)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float f[4];
            cvt4_f32(v[q], f);
            const f32x4_t gq = rg[i][q];
            v[q] = cvt4_fp8(f[0] * gq[0], f[1] * gq[1], f[2] * gq[2], f[3] * gq[3]);
          }
        }
        *(u32x4_t*)(dA + lds_byte(row, a_chunk * 2)) = v;
      } else {
        float f[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { f[2 * q] = lo_f32<DT_F16>(ra[i][q]); f[2 * q + 1] = hi_f32<DT_F16>(ra[i][q]); }
        const float s = a.a_inv_scale;
        u32x2_t o;
        o[0] = cvt4_fp8(f[0] * s, f[1] * s, f[2] * s, f[3] * s);
        o[1] = cvt4_fp8(f[4] * s, f[5] * s, f[6] * s, f[7] * s);
        *(u32x2_t*)(dA + lds_byte(row, a_chunk)) = o;
      }
    }
    *(u32x4_t*)(sB + buf * B_BYTES + lds_byte(b_row, b_chunk * 2)) = rb;
  };

  f32x4_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  issue(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) issue(kt + 1);
    const unsigned char* pa = sA + buf * A_BYTES;
    const unsigned char* pb = sB + buf * B_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      long fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const long*)(pa + lds_byte(wm * 64 + i * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *(const long*)(pb + lds_byte(wn * 32 + j * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) stash(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: acc -> per-wave LDS (fp32) -> 8 consecutive channels of a row per lane ----
  constexpr int EPI_LD = 36;                      // 32 columns + pad
  float* const epi = (float*)smem + wave * (16 * EPI_LD);   // 4 x 2304 B <= the tile buffers
  const int ecol = (lane & 3) * 8, erow = lane >> 2;        // 4 lanes per row, 16 rows per pass
  const int gcol = n0 + wn * 32 + ecol;
  float sc[8], bi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = a.scale[gcol + j]; bi[j] = a.bias[gcol + j]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) epi[(fq * 4 + r) * EPI_LD + j * 16 + frow] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    {
    const int m = m0 + wm * 64 + i * 16 + erow;
    const f32x4_t v0 = *(const f32x4_t*)(epi + erow * EPI_LD + ecol);
    const f32x4_t v1 = *(const f32x4_t*)(epi + erow * EPI_LD + ecol + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (m < a.M && gcol < a.cout_s) {
      const size_t o = (size_t)m * a.cout_s + gcol;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + bi[j];
      if (a.res) {
        const u32x4_t rr = *(const u32x4_t*)(a.res + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] += lo_f32<DT_F16>(rr[j]); v[2 * j + 1] += hi_f32<DT_F16>(rr[j]); }
      }
      if (a.act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      } else if (a.act == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
      }
      if (OUT_FP8) {
        const float s = a.y_inv_scale;
        u32x2_t ov;
        ov[0] = cvt4_fp8(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
        ov[1] = cvt4_fp8(v[4] * s, v[5] * s, v[6] * s, v[7] * s);
        *(u32x2_t*)((unsigned char*)a.y + o) = ov;
      } else {
        u32x4_t ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = pack2<DT_F16>(v[2 * j], v[2 * j + 1]);
        *(u32x4_t*)((bf16_t*)a.y + o) = ov;
        if (a.y8) {   // the same bytes the expand loader would make of the stored fp16 values
          const float s8 = a.y8_inv_scale;
          float r[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { r[2 * j] = lo_f32<DT_F16>(ov[j]) * s8; r[2 * j + 1] = hi_f32<DT_F16>(ov[j]) * s8; }
          u32x2_t o8;
          o8[0] = cvt4_fp8(r[0], r[1], r[2], r[3]);
          o8[1] = cvt4_fp8(r[4], r[5], r[6], r[7]);
          *(u32x2_t*)(a.y8 + (size_t)m * a.y8_stride + gcol) = o8;
        }
      }
    } else if (!OUT_FP8 && a.y8 && m < a.M && gcol < a.y8_stride) {
      *(u32x2_t*)(a.y8 + (size_t)m * a.y8_stride + gcol) = u32x2_t{0u, 0u};   // padding columns of the e4m3 copy
    }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// fp32 [cout][cin] (1x1 conv master weights) -> e4m3 [Npad][Kpad] with one scale per output channel:
// byte = e4m3(w * col_scale / ws[n]), ws[n] = max_k |w * col_scale| / 448; wscale_out[n] = ws[n] (1 when the row is zero)
__global__ __launch_bounds__(256) void pack_fp8_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                       float* __restrict__ wscale_out, int cout, int cin, int Npad,
                                                       int Kpad, float col_scale) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  float mx = 0.f;
  if (n < cout)
    for (int k = threadIdx.x; k < cin; k += 256) mx = fmaxf(mx, fabsf(w[(size_t)n * cin + k] * col_scale));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  const float ws = red[0] > 0.f ? red[0] / 448.f : 1.f;
  if (threadIdx.x == 0) wscale_out[n] = ws;
  const float inv = 1.f / ws;
  for (int k4 = threadIdx.x; k4 < Kpad / 4; k4 += 256) {
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k4 * 4 + j;
      f[j] = (n < cout && k < cin) ? w[(size_t)n * cin + k] * col_scale * inv : 0.f;
    }
    *(unsigned int*)(out + (size_t)n * Kpad + k4 * 4) = cvt4_fp8(f[0], f[1], f[2], f[3]);
  }
}

// max |x| of a 16-bit tensor (calibration): per-block maxima -> atomicMax on the float bits (values >= 0)
__global__ void absmax_f16_kernel(const bf16_t* __restrict__ x, size_t n8, unsigned int* __restrict__ out) {
  float mx = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4_t v = *(const u32x4_t*)(x + i * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fmaxf(fabsf(lo_f32<DT_F16>(v[j])), fabsf(hi_f32<DT_F16>(v[j]))));
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned int, mx));
}

template <bool A_FP8, bool GATED, bool OUT_FP8>
int launch_pw(const PwArgs& a, hipStream_t s) {
  const int m_tiles = (a.M + 127) / 128, n_tiles = a.Npad / 64;
  hipLaunchKernelGGL((pw_fp8_kernel<A_FP8, GATED, OUT_FP8>), dim3(m_tiles * n_tiles), dim3(256), 0, s, a, m_tiles, n_tiles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// x: [M][cin_s] (fp16, or e4m3 when a_fp8), w: e4m3 [Npad][Kpad], y: [M][cout_s] (fp16, or e4m3 when out_fp8).
// gate (a_fp8 only): fp32 [M / hw][gate_stride] multiplied into the A operand.  Returns 0 / -1 / -2 (unsupported).
int spk_launch_pw_fp8(const void* x, int a_fp8, const unsigned char* w, void* y, int out_fp8, const bf16_t* res,
                      const float* scale, const float* bias, const float* gate, int gate_stride, int hw, int M, int Kpad,
                      int Npad, int cin_s, int cout_s, int act, float a_inv_scale, float y_inv_scale, hipStream_t s,
                      unsigned char* y8, int y8_stride, float y8_inv_scale) {
  if (M < 1 || Kpad % 64 || Npad % 64 || cout_s % 8 || (a_fp8 ? cin_s % 16 : cin_s % 8) || (out_fp8 && res) ||
      (gate && !a_fp8) || (size_t)M * cin_s * (a_fp8 ? 1 : 2) >= ((size_t)1 << 31))
    return -2;
  PwArgs a;
  a.x = x; a.w = w; a.y = y; a.res = res; a.scale = scale; a.bias = bias; a.gate = gate;
  a.M = M; a.Kpad = Kpad; a.Npad = Npad; a.cin_s = cin_s; a.cout_s = cout_s; a.hw = hw > 0 ? hw : 1;
  a.gate_stride = gate_stride; a.act = act; a.a_inv_scale = a_inv_scale; a.y_inv_scale = y_inv_scale;
  a.x_bytes = (unsigned)((size_t)M * cin_s * (a_fp8 ? 1 : 2));
  if (y8 && (out_fp8 || y8_stride % 8 || y8_stride < cout_s || y8_stride > Npad)) return -2;
  a.y8 = y8; a.y8_stride = y8_stride; a.y8_inv_scale = y8_inv_scale;
  if (a_fp8) {
    if (out_fp8) return gate ? -2 : launch_pw<true, false, true>(a, s);
    return gate ? launch_pw<true, true, false>(a, s) : launch_pw<true, false, false>(a, s);
  }
  return out_fp8 ? launch_pw<false, false, true>(a, s) : launch_pw<false, false, false>(a, s);
}

int spk_launch_pack_fp8(const float* w, unsigned char* out, float* wscale_out, int cout, int cin, int Npad, int Kpad,
                        float col_scale, hipStream_t s) {
  hipLaunchKernelGGL(pack_fp8_kernel, dim3(Npad), dim3(256), 0, s, w, out, wscale_out, cout, cin, Npad, Kpad, col_scale);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// *out_bits (device uint, zeroed by the caller) = float bits of max |x| over n8 * 8 fp16 values
int spk_launch_absmax_f16(const bf16_t* x, size_t n8, unsigned int* out_bits, hipStream_t s) {
  size_t g = (n8 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(absmax_f16_kernel, dim3((unsigned)(g < 1 ? 1 : g)), dim3(256), 0, s, x, n8, out_bits);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// Depthwise KxK conv + folded BN + activation on e4m3 tensors (fp8 mode of the MBConv interior): the fp16 kernel
// of effnet.hip with 16 channels (one 16-B access) per thread and 2 horizontally adjacent outputs.  The input scale
// is folded into the tap weights by the caller (w' = w * in_scale); the output is v * out_inv_scale -> e4m3.  The
// squeeze-excitation pool partials are the fp32 sums of v (before the 8-bit rounding), as in the fp16 kernel.
// ---------------------------------------------------------------------------------------------------------------
namespace {

template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_fp8_kernel(const unsigned char* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ scale, const float* __restrict__ bias,
                                                         unsigned char* __restrict__ y, float* __restrict__ partial,
                                                         int h, int wid, int c_p, int ho, int wo, int act, int chunks,
                                                         float in_scale, float out_inv_scale) {
  constexpr int PAD = (K - 1) / 2, PX = 2, CPT = 16;  // PX 4: 64 accumulators + 16 inputs per thread, measured 35 % slower
  constexpr int COLS = (PX - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [K*K + 2][tc]; reused for the pool reduce
  const int img = blockIdx.y / chunks, chunk = blockIdx.y % chunks;
  const int c16 = c_p / CPT;
  const int cg0 = blockIdx.x * 16;
  const int ncg = min(16, c16 - cg0), tc = ncg * CPT;
  dwu::stage_dw_weights<K, CPT>(sm, w, scale, bias, c_p, cg0 * CPT, ncg, ncg, in_scale, threadIdx.x);   // layout: dw_util.h
  __syncthreads();
  const int rows = 256 / ncg;
  const int cg = threadIdx.x % ncg, prow = threadIdx.x / ncg;
  const int gpr = (wo + PX - 1) / PX, ngroups = ho * gpr;
  const int per = (ngroups + chunks - 1) / chunks;
  const int g0 = chunk * per, g1 = min(ngroups, g0 + per);
  float pool[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) pool[j] = 0.f;
  if (prow < rows) {
    const float* wl = sm + cg * 4;   // + (row * 4 + quarter) * ncg * 4
    const unsigned char* xi = x + (size_t)img * h * wid * c_p + (cg0 + cg) * CPT;
    unsigned char* yi = y + (size_t)img * ho * wo * c_p + (cg0 + cg) * CPT;
    for (int g = g0 + prow; g < g1; g += rows) {
      const int oy = g / gpr, ox0 = (g - oy * gpr) * PX;
      float acc[PX][CPT];
#pragma unroll
      for (int u = 0; u < PX; ++u)
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[u][j] = 0.f;
#pragma unroll 1
      for (int r = 0; r < K; ++r) {
        const int iy = oy * S - PAD + r;
        if ((unsigned)iy >= (unsigned)h) continue;
        // all COLS loads of the row before the first use: unconditional, clamped column, zeroed by a select
        u32x4_t raw[COLS];
#pragma unroll
        for (int col = 0; col < COLS; ++col) {
          const int ix = ox0 * S - PAD + col;
          const int ixc = min(max(ix, 0), wid - 1);
          raw[col] = *(const u32x4_t*)(xi + ((size_t)iy * wid + ixc) * c_p);
          if ((unsigned)ix >= (unsigned)wid) raw[col] = u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int col = 0; col < COLS; ++col) {
          float xv[CPT];
#pragma unroll
          for (int q = 0; q < 4; ++q) cvt4_f32(raw[col][q], xv + 4 * q);
#pragma unroll
          for (int u = 0; u < PX; ++u) {
            const int q = col - u * S;  // tap of output u that this column feeds
            if (q >= 0 && q < K) {
#pragma unroll
              for (int j4 = 0; j4 < 4; ++j4) {
                const f32x4_t wt = *(const f32x4_t*)(wl + ((r * K + q) * 4 + j4) * ncg * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[u][4 * j4 + j] += xv[4 * j4 + j] * wt[j];
              }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < PX; ++u) {
        if (ox0 + u >= wo) continue;
#pragma unroll
        for (int j = 0; j < CPT; ++j)
          acc[u][j] = acc[u][j] * wl[((K * K) * 4 + (j >> 2)) * ncg * 4 + (j & 3)] + wl[((K * K + 1) * 4 + (j >> 2)) * ncg * 4 + (j & 3)];
        if (act == 2) {   // uniform branch: one activation's instructions, not both + selects
#pragma unroll
          for (int j = 0; j < CPT; ++j) acc[u][j] = silu_f(acc[u][j]);
        } else if (act == 1) {
#pragma unroll
          for (int j = 0; j < CPT; ++j) acc[u][j] = fmaxf(acc[u][j], 0.f);
        }
        u32x4_t o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pool[4 * q + j] += acc[u][4 * q + j];
            v[j] = acc[u][4 * q + j] * out_inv_scale;
          }
          o[q] = cvt4_fp8(v[0], v[1], v[2], v[3]);
        }
        *(u32x4_t*)(yi + ((size_t)oy * wo + ox0 + u) * c_p) = o;
      }
    }
  }
  if (!partial) return;
  __syncthreads();
  if (prow < rows) {
#pragma unroll
    for (int j = 0; j < CPT; ++j) sm[prow * tc + cg * CPT + j] = pool[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < tc; c += 256) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += sm[r * tc + c];
    partial[((size_t)img * chunks + chunk) * c_p + cg0 * CPT + c] = t;
  }
}

}  // namespace

// chunks: spk_dw_chunks(n, ho * ceil(wo / 2), c_p) (the caller passes it: the
// squeeze-excitation kernels read [n][chunks][c_p] partials)
int spk_launch_dwconv_fp8(const unsigned char* x, const float* w, const float* scale, const float* bias, unsigned char* y,
                          float* partial, int n, int h, int wid, int c_p, int ho, int wo, int k, int stride, int act,
                          int chunks, float in_scale, float out_inv_scale, hipStream_t s) {
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2) || c_p % 16) return -2;
  const int c16 = c_p / 16, ctiles = (c16 + 15) / 16;
  const int tc = (c16 < 16 ? c16 : 16) * 16;
  const size_t lds = (size_t)((k * k + 2) * tc > 256 / (tc / 16) * tc ? (k * k + 2) * tc : 256 / (tc / 16) * tc) * 4;
  const dim3 grid(ctiles, n * chunks);
#define SPK_DW8(K, S)                                                                                                  \
  hipLaunchKernelGGL((dwconv_fp8_kernel<K, S>), grid, dim3(256), lds, s, x, w, scale, bias, y, partial, h, wid, c_p, ho, wo, \
                     act, chunks, in_scale, out_inv_scale)
  if (k == 3 && stride == 1) SPK_DW8(3, 1);
  else if (k == 3) SPK_DW8(3, 2);
  else if (stride == 1) SPK_DW8(5, 1);
  else SPK_DW8(5, 2);
#undef SPK_DW8
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// out[i] = a[i] * b[i] * c   (folds BN scale x weight scale x activation scale into the epilogue factor)
namespace {
__global__ void mul3_kernel(const float* __restrict__ a, const float* __restrict__ b, float c, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i] * c;
}
}  // namespace
int spk_launch_mul3(const float* a, const float* b, float c, float* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(mul3_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, c, out, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
