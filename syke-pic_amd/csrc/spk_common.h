// Internal declarations shared by the HIP translation units of
// libsykepic_hip.so (gfx950 only).  Public C-ABI: include/sykepic_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>

// Raises a kernel's dynamic-LDS limit once per DEVICE (hipFuncSetAttribute applies to the current device only; a
// process may hold model handles on several GPUs).  `mask`: one function-local static per kernel instantiation, one bit
// per device; devices >= 64 set the attribute on every launch.  Returns false when the runtime refuses.
static inline bool spk_lds_limit_once(std::atomic<unsigned long long>& mask, const void* kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  if (mask.load(std::memory_order_acquire) & bit) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  mask.fetch_or(bit, std::memory_order_release);
  return true;
}

typedef unsigned short bf16_t;  // storage type of a bfloat16 value

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __builtin_bit_cast(float, ((unsigned int)v) << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  return (unsigned int)f32_to_bf16(lo) | ((unsigned int)f32_to_bf16(hi) << 16);
}

// 16-bit activation/weight storage type of a kernel instantiation.
// bf16: training (range for gradients).  f16: inference (3 more mantissa bits
// at the same MFMA rate; needed for the 1e-3 probability tolerance).
enum { DT_BF16 = 0, DT_F16 = 1 };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// SiLU x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division: `x / (1 + exp(-x))`
// compiles to ~12 VALU instructions per element (v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas, v_div_fixup), which made
// the epilogue - not the MFMA / FMA work - the longest part of EfficientNet's expand and depthwise layers.  Every
// caller rounds the result to fp16 or e4m3 next (>= 2^-11 relative), so the 1-ulp fp32 reciprocal is invisible.
// x -> -inf: exp overflows to +inf, rcp gives 0, the product is -0 like the division's.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

template <int DT> __device__ __forceinline__ float lo_f32(unsigned int u) {
  if (DT == DT_BF16) return __builtin_bit_cast(float, u << 16);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
}
template <int DT> __device__ __forceinline__ float hi_f32(unsigned int u) {
  if (DT == DT_BF16) return __builtin_bit_cast(float, u & 0xffff0000u);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
}
template <int DT> __device__ __forceinline__ unsigned short to_h16(float f) {
  if (DT == DT_BF16) return f32_to_bf16(f);
  // saturate instead of overflowing to inf
  return __builtin_bit_cast(unsigned short, (_Float16)fminf(fmaxf(f, -65504.f), 65504.f));
}
template <int DT> __device__ __forceinline__ unsigned int pack2(float lo, float hi) {
  return (unsigned int)to_h16<DT>(lo) | ((unsigned int)to_h16<DT>(hi) << 16);
}
template <int DT> __device__ __forceinline__ f32x4_t mfma16(const u32x4_t a, const u32x4_t b, f32x4_t c) {
  if (DT == DT_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution (conv_igemm.hip).
//   M = N*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin (tap-major, channel
//   minor), activations NHWC bf16, weights [Cout][K] bf16.
// Epilogue: v = acc*scale[c] + bias[c] (+ residual) (ReLU) -> bf16.
// ---------------------------------------------------------------------------
enum { CONV_MODE_GENERIC = 0, CONV_MODE_STEM = 1, CONV_MODE_DGRAD = 2, CONV_MODE_STEM3 = 3 /* 3x3/2 direct stem */,
       CONV_MODE_DGRAD_BNB = 4 /* kernel-internal: DGRAD + the BatchNorm-backward sums of the producer (spk_set_bnb) */ };

struct ConvArgs {
  const bf16_t* x;      // [N,H,W,Cin]   (stem mode: Cin stored = 4)
  const bf16_t* w;      // [Cout][K]
  bf16_t* y;            // [N,Ho,Wo,Cout]
  const bf16_t* res;    // [N,Ho,Wo,Cout] or null
  const bf16_t* res_lo; // rounding remainder of res (res_true = res + res_lo) or null
  bf16_t* y_lo;         // if set: y_true - y, kept for a later shortcut add
  const float* scale;   // [Cout] or null (=1)
  const float* bias;    // [Cout] or null (=0)
  float* stats;         // train: [m_tiles][2][Cout] partial sum / sum of squares of the raw output, or null
  int N, H, W, Cin, Ho, Wo, Cout;
  int kh, kw, stride, pad;
  int M, K;
  int relu;
  int dt;               // DT_BF16 / DT_F16
  int splitw;           // w holds [2][Cout][K]: hi then lo halves (f16 eval only)
  int cfg, dma;         // tile config / main loop (0 register-staged, 1 LDS-DMA, 2 register-staged persistent); -1 = autotuned or heuristic
  int cls_ph, cls_pw;   // dgrad of a stride-2 conv: output parity class of this launch (-1: none)
  int oH, oW;           // ... and the full output height/width (Ho/Wo are then the class grid)
  int kt_count;         // K steps of this launch when not K/64 (parity classes use a tap subset)
  unsigned int x_bytes, w_bytes;
  // channels per pixel as stored in HBM when they differ from the GEMM's 64-padded Cin / Cout (EfficientNet widths
  // are multiples of 8 only): 0 = same.  K chunks past cin_s read as zeros, output columns past cout_s are not stored.
  int cin_s, cout_s;
  // stem only (conv_stem.hip): when set, the kernel applies the 3x3 stride-2 pad-1 max-pool that follows the stem in
  // every torchvision ResNet to its own output tile and writes ONLY the pooled tensor [N,pool_ho,pool_wo,64] here
  // (y is not written).  Needs relu, no stats.
  bf16_t* pool_y;
  int pool_ho, pool_wo;
  // DGRAD mode with `stats` set (round 4): the tensor this launch completes, y = dL/d(activation t), is the input of the
  // BatchNorm backward of the layer P that produced t, and the epilogue also emits P's reduction - per row tile the
  // per-channel sums of dz = y * relu_mask and dz * xhat into `stats` ([m_tiles][2][Cout], the layout of the forward
  // statistics), from the fp32 values BEFORE they are rounded to 16 bits: P's stand-alone reduce pass (one more read of y
  // and of P's raw output) is not launched.  A data gradient has no BatchNorm fold, remainder tensors or pooling, so the
  // operands travel in those fields (the struct - and with it every forward kernel's code - stays as it was):
  //   scale = P's batch mean [Cout], bias = 1/sqrt(var + eps) [Cout], res_lo = P's raw conv output [N,Ho,Wo,Cout],
  //   pool_y = ReLU bits of P's output, one per element (null: no ReLU);
  //   y_lo (optional) = ReLU bits that mask the `res` operand: res is then the output gradient of the block-closing conv
  //   whose shortcut this tensor is, and res * bit the shortcut gradient its BatchNorm backward did not write.
  //   See spk_set_bnb().
};

static inline void spk_set_bnb(ConvArgs& a, float* partials, const bf16_t* raw, const unsigned char* relu_bits,
                               const float* mean, const float* invstd, const unsigned char* res_bits = nullptr) {
  a.stats = partials;
  a.scale = mean;
  a.bias = invstd;
  a.res_lo = raw;
  a.pool_y = (bf16_t*)relu_bits;
  a.y_lo = (bf16_t*)res_bits;
}

// Forward 1x1 conv (fp16 eval) whose activation operand is x[n, h, w, c] * gate[n * gate_stride + c]: the squeeze-
// excitation scaling of an MBConv block, applied where the project conv stages its operand (fp32 product rounded to 16
// bits - the tensor the stand-alone scale pass would have written).  The stem-only pooling fields carry the operands
// (the struct - and with it every other kernel's code - stays as it was).
static inline void spk_set_gate(ConvArgs& a, const float* gate, int gate_stride) {
  a.pool_y = (bf16_t*)gate;
  a.pool_ho = gate_stride;
}

// returns 0 on success; fills *m_tiles with the number of row tiles used
// (needed to size/finalize the stats partials)
int spk_conv_launch(const ConvArgs& a, int mode, hipStream_t s, int* m_tiles_out);
int spk_conv_m_tiles(int M, int Cout, int mode);
int spk_conv_stem_launch(const ConvArgs& a, hipStream_t s, int* m_tiles_out);  // conv_stem.hip
const char* spk_conv_last_config();

// ---------------------------------------------------------------------------
// 1x1 convolution with activation fragments loaded straight into VGPRs and fragment-ordered weights (conv_pw.hip)
// ---------------------------------------------------------------------------
struct PwConvArgs {
  const bf16_t* x;      // [N,H,W,Cin]
  const bf16_t* wp;     // packed by spk_launch_pack_pw
  bf16_t* y;            // [N,Ho,Wo,Cout]
  const bf16_t* res;    // [N,Ho,Wo,Cout] or null
  const float* scale;   // [Cout] or null (= 1): applied to the fp32 accumulator
  const float* shift;   // [Cout] or null (= 0)
  int N, H, W, Ho, Wo, stride;
  int Cin, Cout, M;
  int relu;             // 0 none, 1 ReLU, 2 SiLU
  int dt;               // DT_BF16 / DT_F16
  int nb;               // 1: plain weights, 2: hi + lo images (fp16 eval only)
  unsigned int x_bytes, y_bytes;
  int ablate;           // timing experiments only: 1 drop the output stores, 2 the shortcut loads, 4 the activation loads
                        // (zero-record buffer descriptors: the instructions still issue, the range check drops them)
  // Second activation source (K-concatenated GEMM: K steps [0, Cin/32) read x, steps [Cin/32, (Cin+Cin2)/32) read x2 at
  // the same output pixel through its own stride): a block-closing 1x1 conv and the 1x1 shortcut (downsample) conv of
  // the same block as ONE kernel - the shortcut tensor is never written or re-read.  x2 == null: plain conv.
  const bf16_t* x2;     // [N,H2,W2,Cin2]
  int Cin2, H2, W2, stride2;
  unsigned int x2_bytes;
  // Chained second 1x1 conv (round 4): z = act2(BN2(W2 . y)) for the SAME pixels, computed from the output tile while it
  // is still in registers - a bottleneck's block-closing conv and the next block's first conv as one kernel: the trunk
  // tensor y is written (later shortcut adds need it) but never re-read by the conv that follows.  wpz == null: none.
  // Needs the whole cout range in one block (resident flavour, n_tiles == 1) and single (nb == 1) weight images.
  const bf16_t* wpz;    // [Cout -> Coutz] packed by spk_launch_pack_pw (nb = 1)
  bf16_t* z;            // [N,Ho,Wo,Coutz]
  const float* scalez;  // [Coutz]
  const float* shiftz;
  int Coutz, reluz;
  unsigned int z_bytes;
};
int spk_pw_num_configs();
// K-concatenation of two 1x1 convs that are added (w1 [Cout][Cin1] with eval-BN scale s1, w2 [Cout][Cin2] with s2):
// wcat[c] = [w1[c] * s1[c] | w2[c] * s2[c]] / 2^e[c], 2^e[c] = the largest power of two <= max(|s1[c]|, |s2[c]|) (so the
// 16-bit hi + lo images keep their precision whatever the BatchNorm scales are), scale_out[c] = 2^e[c] exactly,
// shift_out[c] = b1[c] + b2[c].
int spk_launch_pw_dual_prep(const float* w1, const float* w2, const float* s1, const float* s2, const float* b1,
                            const float* b2, float* wcat, float* scale_out, float* shift_out, int cout, int cin1,
                            int cin2, hipStream_t s);
// the dual-source conv by its fastest configuration (timed once per problem, "pw2 ..." lines of SPK_TUNE_CACHE);
// -3: no configuration fits
int spk_conv1x1_dual_launch(const PwConvArgs& q, hipStream_t s);
int spk_pw_launch(const PwConvArgs& a, int cfg, hipStream_t s);   // -3: this config does not fit the problem
// conv_pwr.hip: the 1x1 kernel of the deep layers (configurations 12.. of spk_pw_launch); -3: does not fit the problem
int spk_pwr_num_configs();
int spk_pwr_launch(const PwConvArgs& a, int cfg, hipStream_t s);
int spk_pw_chain_launch(const PwConvArgs& a, hipStream_t s);      // with PwConvArgs::wpz; -3: no chained kernel for this problem
int spk_launch_pack_pw(const float* w, const float* scale, bf16_t* out, int cout, int cin, int dt, int nb, hipStream_t s);
// eval path: the faster of the implicit GEMM (a) and conv_pw (q) for this problem, tuned once and cached
int spk_conv1x1_launch(const ConvArgs& a, const PwConvArgs& q, hipStream_t s);

// ---------------------------------------------------------------------------
// 3x3 stride-1 pad-1 convolution, activations as a halo window in LDS, weights straight from L2 (conv_c3.hip)
// ---------------------------------------------------------------------------
struct C3Args {
  const bf16_t* x;      // [N,H,W,Cin] fp16
  const bf16_t* wp;     // packed by spk_launch_pack_c3
  bf16_t* y;            // [N,H,W,Cout] fp16
  const float* scale;   // [Cout] or null
  const float* shift;   // [Cout] or null
  int N, H, W, Cin, Cout, M;
  int relu, dt, nb;
  unsigned int x_bytes, y_bytes, wp_bytes;
  const bf16_t* res;    // shortcut [N,H,W,Cout] added before the activation (ResNet-18/34 block-closing conv), or null
};
int spk_c3_num_configs();
int spk_c3_launch(const C3Args& a, int cfg, hipStream_t s);   // -3: this config does not fit the problem
int spk_launch_pack_c3(const float* w_ohwi, bf16_t* out, int cout, int cin, int nb, hipStream_t s);
// eval path: the fastest configuration for this problem, tuned once and cached (all give bit-identical results);
// -3 when none fits (the caller then runs the implicit GEMM)
int spk_conv3x3_launch(const C3Args& a, hipStream_t s);

// ---------------------------------------------------------------------------
// A whole identity bottleneck block (1x1 -> 3x3 -> 1x1 + shortcut, ReLU after each BatchNorm) as one kernel: the two mid
// tensors stay in LDS (conv_bneck.hip).  fp16, single weight images.
// ---------------------------------------------------------------------------
struct BneckArgs {
  const bf16_t* x;      // [N,H,W,C4] fp16: block input = shortcut
  bf16_t* y;            // [N,H,W,C4] fp16 (may not alias x: a band reads the halo rows of x its neighbours write in y)
  const bf16_t* w1;     // conv1 [CM][C4]        packed by spk_launch_pack_pw (nb = 1)
  const bf16_t* w2;     // conv2 [CM][3][3][CM]  packed by spk_launch_pack_c3 (nb = 1)
  const bf16_t* w3;     // conv3 [C4][CM]        packed by spk_launch_pack_pw (nb = 1)
  const float *s1, *b1, *s2, *b2, *s3, *b3;   // folded eval-BatchNorm scale / shift of the three convs
  int N, H, W, C4, CM;
  unsigned int x_bytes; // N*H*W*C4*2
  // spk_btail_launch only (conv2 + conv3 + shortcut of a block whose conv1 already ran, + optionally the next block's conv1):
  const bf16_t* y1;     // [N,H,W,CM] fp16: the block's conv1 output
  const bf16_t* wz;     // chained conv [Coutz][C4] packed by spk_launch_pack_pw (nb = 1), or null
  bf16_t* z;            // [N,H,W,Coutz]
  const float *sz, *bz; // its folded BatchNorm (ReLU behind it)
  int Coutz;
  unsigned int z_bytes;
  int flags;            // 4: the 7-row / 4-wave block form.  Timing experiments (SPK_BNECK_FLAGS, results then wrong): 1 no static wave
                        // priority in phase 3, 8 phase 1 reads image 0 for every block (x from L2), 16 no shortcut loads, 32 no stores
  unsigned long long* stamps;   // diagnostics (tools/bneck_bench.py): [block][2][8] s_memtime values at the phase boundaries (first wave, first wave of the second half), or null
};
int spk_bneck_launch(const BneckArgs& a, hipStream_t s);   // -3: no kernel for this shape
int spk_btail_launch(const BneckArgs& a, hipStream_t s);   // conv2 + conv3 (+ chained conv) from y1; -3: no kernel for this shape

// ---------------------------------------------------------------------------
// Zero-sum rounding of fp16 weights + activation means (zero_sum.hip)
// ---------------------------------------------------------------------------
// w, out: [rows][row_len] fp32; mu: [mu_period] (element k is weighted with mu[k % mu_period]) or null (all ones).
// out[i] is fp16(w[i]) or its neighbour on the other side of w[i], chosen so that sum_k mu_k (out_k - w_k) ~ 0 per row.
int spk_launch_zero_sum_round(const float* w, const float* mu, float* out, size_t rows, int row_len, int mu_period,
                              hipStream_t s);
int spk_chan_mean_slices(size_t rows);
int spk_launch_chan_mean(const bf16_t* x, float* part, float* mean, size_t rows, int C, int dt, hipStream_t s);

// ---------------------------------------------------------------------------
// Pointwise / pooling / packing kernels (pointwise.hip)
// ---------------------------------------------------------------------------
// image batch -> NHWC bf16 with channels padded to 4 (stem input)
// `scale`: what a pixel VALUE (a float as given, or k / 255 for uint8) is multiplied with before the 16-bit rounding.
// The model executors pass SPK_INPUT_SCALE = 255: the reference's inputs are k / 255 (ToTensor), which no 16-bit float
// holds exactly, while the integers 0..255 are exact in fp16 AND bf16 - the stem kernels take the factor back in fp32
// (eval: folded BatchNorm scale / 255; training: stem weights / 255 at packing, weight gradient / 255).  Measured on
// trained nets (tests/archive/diagnostics/input_rounding.py): against the fp32 oracle the eval path was 1.2e-3 off on the worst of
// 256 images, 7.6e-5 against the oracle fed the fp16-rounded pixels - the input rounding WAS the error.
#define SPK_INPUT_SCALE 255.0f
int spk_launch_to_nhwc4(const void* x, int layout, int dtype, int n, int c, int h, int w,
                        bf16_t* out, int dt, hipStream_t s, float scale = 1.0f);
int spk_launch_scale_inplace(float* x, float f, int n, hipStream_t s);
int spk_launch_maxpool(const bf16_t* x, bf16_t* y, int n, int h, int w, int c, int k, int stride,
                       int pad, int ho, int wo, int dt, hipStream_t s);
int spk_launch_gavgpool(const bf16_t* x, float* y, int n, int hw, int c, int dt, hipStream_t s);
// BatchNorm(eval) folding: scale = g/sqrt(var+eps), bias = b - mean*scale
int spk_launch_bn_fold(const float* g, const float* b, const float* mean, const float* var,
                       float eps, float* scale, float* bias, int c, hipStream_t s);
// master fp32 KRSC weights -> bf16 [Cout][K] (generic) or the stem image
// splitw: out = [2][Cout][K], second half = remainder w - float(first half)
// EfficientNet pieces (effnet.hip)
int spk_launch_pack_padded(const float* w, bf16_t* out, int cout, int taps, int cin, int cout_p, int cin_p, int dt,
                           int splitw, hipStream_t s);
int spk_launch_pack_tapmajor(const float* w, float* out, int c, int rows, int c_p, hipStream_t s);
int spk_launch_stem3x3(const bf16_t* x, const float* w, const float* scale, const float* bias, bf16_t* y, int n, int h,
                       int wid, int wstride, int ho, int wo, int c, int c_p, int act, int dt, hipStream_t s);
int spk_dw_chunks(int n, int hw, int c_p);
int spk_launch_dwconv(const bf16_t* x, const float* w, const float* scale, const float* bias, bf16_t* y, float* partial,
                      int n, int h, int wid, int c_p, int ho, int wo, int k, int stride, int act, int dt, hipStream_t s);
// y == nullptr: the gates only (scale[n][c_p]); the caller applies them elsewhere (fp8 mode: in the project conv)
int spk_launch_se(const bf16_t* x, bf16_t* y, const float* partial, int chunks, float* scale, const float* w1,
                  const float* b1, const float* w2t, const float* b2, int n, int hw, int c, int c_p, int sq, int dt,
                  hipStream_t s);
int spk_launch_pack_weights(const float* w_krsc, bf16_t* out, int cout, int kh, int kw, int cin,
                            int mode, int dt, int splitw, hipStream_t s, float stem_wscale = 1.0f);
// fp8 (e4m3) mode of the MBConv interior (pw_fp8.hip)
int spk_launch_pw_fp8(const void* x, int a_fp8, const unsigned char* w, void* y, int out_fp8, const bf16_t* res,
                      const float* scale, const float* bias, const float* gate, int gate_stride, int hw, int M, int Kpad,
                      int Npad, int cin_s, int cout_s, int act, float a_inv_scale, float y_inv_scale, hipStream_t s,
                      unsigned char* y8 = nullptr, int y8_stride = 0, float y8_inv_scale = 0.f);   // e4m3 copy of an fp16 output
int spk_launch_pack_fp8(const float* w, unsigned char* out, float* wscale_out, int cout, int cin, int Npad, int Kpad,
                        float col_scale, hipStream_t s);
int spk_launch_absmax_f16(const bf16_t* x, size_t n8, unsigned int* out_bits, hipStream_t s);
int spk_launch_dwconv_fp8(const unsigned char* x, const float* w, const float* scale, const float* bias, unsigned char* y,
                          float* partial, int n, int h, int wid, int c_p, int ho, int wo, int k, int stride, int act,
                          int chunks, float in_scale, float out_inv_scale, hipStream_t s);
int spk_launch_mul3(const float* a, const float* b, float c, float* out, int n, hipStream_t s);
// LDS-staged depthwise conv (dwconv_lds.hip): et 0 fp16 tensors, 1 e4m3 tensors
int spk_dwconv_lds_chunks(int et, int n, int h, int wid, int c_p, int ho, int wo, int k, int stride);
int spk_launch_dwconv_lds(int et, const void* x, const float* w, const float* scale, const float* bias, void* y,
                          float* partial, int n, int h, int wid, int c_p, int ho, int wo, int k, int stride, int act,
                          float w_scale, float out_inv_scale, hipStream_t s);

// ---------------------------------------------------------------------------
// Head (head.hip): fp32 Linear layers, softmax, cross-entropy
// ---------------------------------------------------------------------------
// C[i][j] (+)= alpha * sum_k A(i,k)*B(j,k) + bias[j], arbitrary element strides
int spk_launch_sgemm(const float* A, long sai, long sak, const float* B, long sbj, long sbk,
                     const float* bias, float* C, long sci, long scj, int M, int N, int K,
                     float alpha, int accumulate, hipStream_t s);
int spk_launch_linear_fwd(const float* x, const float* w, const float* b, float* y, int n, int in,
                          int out, hipStream_t s);
int spk_launch_softmax(const float* z, float* p, int n, int c, float scale, hipStream_t s);
// stats[0] += sum_i CE_i ; stats[1] += #(argmax == y); dlogits (may be null) = (softmax - onehot)/n
int spk_launch_ce(const float* z, const int64_t* y, int n, int c, float* stats, float* dlogits,
                  hipStream_t s);

// ---------------------------------------------------------------------------
// Training kernels (train_kernels.hip, conv_wgrad.hip)
// ---------------------------------------------------------------------------
struct OptEntry {
  unsigned long long off;  // element offset in the flat buffers
  unsigned int n;
  float lr, bc1, bc2s;
  int first;               // first step of this tensor (momentum buffers / accumulators start here)
  float c0, c1;            // per-tensor, per-step scalars of the other optimizers (see opt_generic_kernel)
};
struct OptTable {
  OptEntry e[64];
  int count;
};

struct PackEntry {
  unsigned long long src, dst;   // element offsets: master weights in the flat fp32 buffer / image in the bf16 buffer
  unsigned int cout, taps, cin;
  int kind;                      // 0 forward image (element-wise), 1 data-gradient image (transposed)
};
struct PackTable {
  PackEntry e[64];
  int count;
};
int spk_launch_pack_multi(const float* pbuf, bf16_t* wpack, const PackTable& t, hipStream_t s);

int spk_launch_bn_finalize(const float* partials, int m_tiles, int C, double M, const float* gamma,
                           const float* beta, float* rmean, float* rvar, float* mean, float* invstd,
                           float* scale, float* shift, float eps, float momentum, float* tmp,
                           hipStream_t s);
int spk_launch_bn_apply(const bf16_t* y, const float* scale, const float* shift, const bf16_t* res,
                        bf16_t* a, unsigned char* mask, size_t numel, int C, int relu, hipStream_t s);
int spk_bn_bwd_blocks(int M, int C, int* rows_per_block);
// `mask`: one ReLU bit per element ([M][C/8] bytes) written by spk_launch_bn_apply
// pre_blocks > 0: `partials` already holds pre_blocks rows of [2][C] sums (written by the dgrad epilogue that produced g,
// ConvArgs::bnb_raw): the reduce pass is skipped
int spk_launch_bn_bwd(const bf16_t* g, const unsigned char* mask, const bf16_t* y, const float* mean,
                      const float* invstd, const float* gamma, float* partials, float* coef,
                      float* dgamma, float* dbeta, bf16_t* dy, bf16_t* g_res, int res_accumulate, int M,
                      int C, int relu, float* tmp, hipStream_t s, int pre_blocks = 0);
int spk_launch_maxpool_idx(const bf16_t* x, bf16_t* y, unsigned char* idx, int n, int h, int w, int c,
                           int k, int stride, int pad, int ho, int wo, hipStream_t s);
int spk_launch_maxpool_bwd(const bf16_t* gy, const unsigned char* idx, bf16_t* gx, int n, int h, int w,
                           int c, int k, int stride, int pad, int ho, int wo, hipStream_t s);
int spk_launch_gavgpool_bwd(const float* gy, bf16_t* gx, int n, int hw, int c, hipStream_t s);
int spk_launch_colsum(const float* dy, float* db, int n, int c, hipStream_t s);
int spk_launch_dropout_fwd(const float* x, float* y, unsigned char* mask, size_t n, float p, unsigned long long seed,
                           hipStream_t s);
int spk_launch_dropout_bwd(const float* gy, const unsigned char* mask, float* gx, size_t n, float p, hipStream_t s);
int spk_launch_slab_reduce(const float* slabs, float* out, size_t n, int splits, hipStream_t s, float scale = 1.0f);
int spk_launch_stem_wgrad_unpack(const float* slabs, float* out, int cout, int kh, int kw, int cin,
                                 int splits, hipStream_t s, float scale = 1.0f);
int spk_launch_pack_dgrad(const float* w, bf16_t* out, int cout, int taps, int cin, hipStream_t s);
int spk_launch_opt_multi(int kind, float* p, const float* g, float* m, float* v, const OptTable& t,
                         float b1, float b2, float eps, float wd, float momentum, float gscale, float alpha,
                         hipStream_t s);
void spk_wgrad_plan(int M, int Cout, int Ktot, int* splits, int* pix_per_split);
int spk_wgrad_launch(const bf16_t* x, const bf16_t* dy, float* slabs, int N, int H, int W, int Cin,
                     int Ho, int Wo, int Cout, int k, int stride, int pad, int stem, int splits,
                     int pix_per_split, hipStream_t s);
