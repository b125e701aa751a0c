// Single-operator entry points of the C-ABI (include/sykepic_hip.h, "test hooks: single operators").
// Each one runs exactly the launches a training step makes for ONE layer (the shared helpers of train.hip and
// the spk_launch_* functions), on caller-provided device buffers, so that a parity test can hand a kernel
// known operands and compare its output with autograd evaluated on the same (bf16-rounded) operands:
//   conv + train-mode BatchNorm forward   torch Conv2d + BatchNorm2d(+add)(+ReLU), sykepic/train/train.py:240
//   BatchNorm / conv backward             what loss.backward() runs for that layer,  sykepic/train/train.py:242
// Scratch is allocated and freed inside the call (these are not on any hot path).
#include "model.h"

#include <cstring>
#include <vector>

static int ofail(int code, const std::string& msg) {
  spk_set_error(msg);
  return code;
}

namespace {
struct Scratch {
  std::vector<void*> p;
  ~Scratch() { for (void* q : p) (void)hipFree(q); }
  template <typename T> T* get(size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, (count ? count : 1) * sizeof(T)) != hipSuccess) return nullptr;
    p.push_back(q);
    return (T*)q;
  }
};
bool stem_shape(int cin, int cout, int k, int stride, int pad) {
  return cin <= 4 && k == 7 && stride == 2 && pad == 3 && cout == 64;
}
}  // namespace

#define O_TRY(expr, what)                                                         \
  do {                                                                            \
    if ((expr) != 0) return ofail(SPK_ERR_HIP, std::string(what) + " failed");    \
  } while (0)

extern "C" int spk_op_conv_bn_train_forward(const void* x, const float* w_ohwi, const float* gamma, const float* beta,
                                            float* running_mean, float* running_var, const void* res, void* out,
                                            void* raw, unsigned char* mask, float* mean_invstd, int n, int h, int w,
                                            int cin, int cout, int k, int stride, int pad, int relu, void* stream) {
  if (!x || !w_ohwi || !gamma || !beta || !running_mean || !running_var || !out || !raw || !mean_invstd || n < 1)
    return ofail(SPK_ERR_ARG, "op_conv_bn_train_forward: bad arguments");
  const bool stem = stem_shape(cin, cout, k, stride, pad);
  if ((!stem && (cin % 64 || cin < 64)) || cout % 64) return ofail(SPK_ERR_UNSUPPORTED, "channels must be multiples of 64 (or the 7x7/2 stem)");
  hipStream_t s = (hipStream_t)stream;
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  const int M = n * oh * ow;
  const int kpad = stem ? 256 : k * k * cin;
  Scratch sc;
  bf16_t* wp = sc.get<bf16_t>((size_t)cout * kpad);
  float* part = sc.get<float>((size_t)((M + 60) / 61) * 2 * cout);
  float* st = sc.get<float>((size_t)2 * cout);
  float* tmp = sc.get<float>((size_t)cout * 2 * 64);
  if (!wp || !part || !st || !tmp) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  O_TRY(spk_launch_pack_weights(w_ohwi, wp, cout, k, k, cin, stem ? CONV_MODE_STEM : CONV_MODE_GENERIC, DT_BF16, 0, s),
        "pack_weights");
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.cfg = a.dma = -1;
  a.cls_ph = a.cls_pw = -1;
  a.x = (const bf16_t*)x; a.w = wp; a.y = (bf16_t*)raw;
  const int wst = stem ? (w + 1) & ~1 : w;  // the stem input is stored with an even row pitch (zero pad column)
  a.N = n; a.H = h; a.W = wst; a.Cin = stem ? 4 : cin; a.Ho = oh; a.Wo = ow; a.Cout = cout;
  a.kh = a.kw = k; a.stride = stride; a.pad = pad;
  a.M = M; a.K = kpad; a.dt = DT_BF16;
  a.x_bytes = (unsigned)((size_t)n * h * wst * a.Cin * 2);
  a.w_bytes = (unsigned)((size_t)cout * kpad * 2);
  a.stats = part;
  int m_tiles = 0;
  O_TRY(spk_conv_launch(a, stem ? CONV_MODE_STEM : CONV_MODE_GENERIC, s, &m_tiles), "conv launch");
  O_TRY(spk_launch_bn_finalize(part, m_tiles, cout, (double)M, gamma, beta, running_mean, running_var, mean_invstd,
                               mean_invstd + cout, st, st + cout, 1e-5f, 0.1f, tmp, s), "bn_finalize");
  O_TRY(spk_launch_bn_apply((const bf16_t*)raw, st, st + cout, (const bf16_t*)res, (bf16_t*)out, mask,
                            (size_t)M * cout, cout, relu, s), "bn_apply");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "op_conv_bn_train_forward: kernel failed");
  return SPK_OK;
}

extern "C" int spk_op_bn_backward(const void* g, const unsigned char* mask, const void* raw, const float* mean,
                                  const float* invstd, const float* gamma, float* dgamma, float* dbeta, void* dy,
                                  void* g_res, int res_accumulate, int M, int C, int relu, void* stream) {
  if (!g || !raw || !mean || !invstd || !gamma || !dy || M < 1 || C % 8 || (relu && !mask))
    return ofail(SPK_ERR_ARG, "op_bn_backward: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  int rpb;
  const int nb = spk_bn_bwd_blocks(M, C, &rpb);
  float* part = sc.get<float>((size_t)nb * 2 * C);
  float* coef = sc.get<float>((size_t)3 * C);
  float* tmp = sc.get<float>((size_t)C * 2 * 64);
  if (!part || !coef || !tmp) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  O_TRY(spk_launch_bn_bwd((const bf16_t*)g, mask, (const bf16_t*)raw, mean, invstd, gamma, part, coef, dgamma, dbeta,
                          (bf16_t*)dy, (bf16_t*)g_res, res_accumulate, M, C, relu, tmp, s), "bn_bwd");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "op_bn_backward: kernel failed");
  return SPK_OK;
}

extern "C" int spk_op_conv_dgrad(const void* dy, const float* w_ohwi, void* dx, int accumulate, int n, int h, int w,
                                 int cin, int cout, int k, int stride, int pad, void* stream) {
  if (!dy || !w_ohwi || !dx || n < 1) return ofail(SPK_ERR_ARG, "op_conv_dgrad: bad arguments");
  if (cin % 64 || cout % 64) return ofail(SPK_ERR_UNSUPPORTED, "channels must be multiples of 64");
  hipStream_t s = (hipStream_t)stream;
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  Scratch sc;
  bf16_t* wdg = sc.get<bf16_t>((size_t)cin * k * k * cout);
  if (!wdg) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  O_TRY(spk_launch_pack_dgrad(w_ohwi, wdg, cout, k * k, cin, s), "pack_dgrad");
  const int r = spk_conv_dgrad_all((const bf16_t*)dy, wdg, (bf16_t*)dx, accumulate != 0, n, oh, ow, cout, h, w, cin, k,
                                   stride, pad, s);
  if (r != SPK_OK) return r;
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "op_conv_dgrad: kernel failed");
  return SPK_OK;
}

// Data gradient of a stride-1 conv whose epilogue also makes the BatchNorm-backward sums of the layer that PRODUCED the
// conv's input (conv_igemm.hip, spk_set_bnb), followed by that layer's finalize + apply with the reduce pass skipped:
// the launches a training step makes for (consumer conv dgrad, producer BatchNorm backward).  dx (+)= conv_transpose(dy, w)
// is the gradient g of the producer's output; raw / mask / mean / invstd / gamma are the producer's; dy_prod receives the
// gradient of the producer's raw conv output, dgamma / dbeta its parameter gradients.
extern "C" int spk_op_conv_dgrad_bn_backward(const void* dy, const float* w_ohwi, void* dx, int accumulate, const void* raw,
                                             const unsigned char* mask, const float* mean, const float* invstd,
                                             const float* gamma, float* dgamma, float* dbeta, void* dy_prod, int n, int h,
                                             int w, int cin, int cout, int k, int pad, int relu, const void* res_src,
                                             const unsigned char* res_bits, void* stream) {
  if (!dy || !w_ohwi || !dx || !raw || !mean || !invstd || !gamma || !dy_prod || n < 1 || (relu && !mask))
    return ofail(SPK_ERR_ARG, "op_conv_dgrad_bn_backward: bad arguments");
  if (cin % 64 || cout % 64) return ofail(SPK_ERR_UNSUPPORTED, "channels must be multiples of 64");
  hipStream_t s = (hipStream_t)stream;
  const int oh = h + 2 * pad - k + 1, ow = w + 2 * pad - k + 1, M = n * h * w;
  Scratch sc;
  bf16_t* wdg = sc.get<bf16_t>((size_t)cin * k * k * cout);
  float* part = sc.get<float>((size_t)((M + 60) / 61) * 2 * cin);
  float* coef = sc.get<float>((size_t)3 * cin);
  float* tmp = sc.get<float>((size_t)cin * 2 * 64);
  if (!wdg || !part || !coef || !tmp) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  O_TRY(spk_launch_pack_dgrad(w_ohwi, wdg, cout, k * k, cin, s), "pack_dgrad");
  BnbFuse fz;
  fz.raw = (const bf16_t*)raw; fz.mask = relu ? mask : nullptr; fz.mean = mean; fz.invstd = invstd; fz.partials = part; fz.tiles = 0;
  fz.res_src = (const bf16_t*)res_src; fz.res_bits = res_bits;
  if ((res_src != nullptr) != (res_bits != nullptr) || (res_src && accumulate))
    return ofail(SPK_ERR_ARG, "op_conv_dgrad_bn_backward: res_src and res_bits go together and replace accumulate");
  const int r = spk_conv_dgrad_all((const bf16_t*)dy, wdg, (bf16_t*)dx, accumulate != 0, n, oh, ow, cout, h, w, cin, k, 1, pad,
                                   s, &fz);
  if (r != SPK_OK) return r;
  if (fz.tiles < 1) return ofail(SPK_ERR_HIP, "the dgrad launch reported no partial rows");
  O_TRY(spk_launch_bn_bwd((const bf16_t*)dx, mask, (const bf16_t*)raw, mean, invstd, gamma, part, coef, dgamma, dbeta,
                          (bf16_t*)dy_prod, nullptr, 0, M, cin, relu, tmp, s, fz.tiles), "bn_bwd");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "op_conv_dgrad_bn_backward: kernel failed");
  return SPK_OK;
}

extern "C" int spk_op_conv_wgrad(const void* x, const void* dy, float* dw_ohwi, int n, int h, int w, int cin, int cout,
                                 int k, int stride, int pad, void* stream) {
  if (!x || !dy || !dw_ohwi || n < 1) return ofail(SPK_ERR_ARG, "op_conv_wgrad: bad arguments");
  const bool stem = stem_shape(cin, cout, k, stride, pad);
  if ((!stem && cin % 64) || cout % 64) return ofail(SPK_ERR_UNSUPPORTED, "channels must be multiples of 64 (or the 7x7/2 stem)");
  hipStream_t s = (hipStream_t)stream;
  const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
  const int M = n * oh * ow;
  Scratch sc;
  float* slabs = sc.get<float>(spk_conv_wgrad_slab_floats(M, cin, cout, k, stem));
  if (!slabs) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  int r = spk_conv_wgrad_slabs((const bf16_t*)x, (const bf16_t*)dy, slabs, n, h, stem ? (w + 1) & ~1 : w, cin, oh, ow,
                               cout, k, stride, pad, stem, s);
  if (r != SPK_OK) return r;
  r = spk_conv_wgrad_reduce(slabs, dw_ohwi, M, cin, cout, k, stem, s);
  if (r != SPK_OK) return r;
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "op_conv_wgrad: kernel failed");
  return SPK_OK;
}

// fp8 (e4m3) pointwise conv of the EfficientNet MBConv interior (pw_fp8.hip) on caller-provided buffers:
// packs the fp32 weights [cout][cin] to e4m3 with per-output-channel scales, folds bn_scale x weight scale x
// a_scale into the epilogue factor and runs the kernel the eval path runs.
extern "C" int spk_op_pw_fp8(const void* x, int a_fp8, const float* w, void* y, int out_fp8, const void* res,
                             const float* bn_scale, const float* bn_bias, const float* gate, int hw, int m_rows, int cin,
                             int cout, int act, float a_scale, float y_scale, void* stream) {
  if (!x || !w || !y || !bn_scale || !bn_bias || m_rows < 1 || cin < 1 || cout < 1 || a_scale <= 0.f || y_scale <= 0.f)
    return ofail(SPK_ERR_ARG, "op_pw_fp8: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int npad = (cout + 63) / 64 * 64, kpad = (cin + 63) / 64 * 64;
  Scratch sc;
  unsigned char* w8 = sc.get<unsigned char>((size_t)npad * kpad);
  float* f = sc.get<float>((size_t)4 * npad);   // ws, epilogue scale, padded bn scale, padded bias
  if (!w8 || !f) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  if (hipMemsetAsync(f, 0, (size_t)4 * npad * 4, s) != hipSuccess ||
      hipMemcpyAsync(f + 2 * npad, bn_scale, (size_t)cout * 4, hipMemcpyDeviceToDevice, s) != hipSuccess ||
      hipMemcpyAsync(f + 3 * npad, bn_bias, (size_t)cout * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return ofail(SPK_ERR_HIP, "op_pw_fp8: staging failed");
  O_TRY(spk_launch_pack_fp8(w, w8, f, cout, cin, npad, kpad, 1.f, s), "pack_fp8");
  O_TRY(spk_launch_mul3(f + 2 * npad, f, a_scale, f + npad, npad, s), "scale folding");
  const int r = spk_launch_pw_fp8(x, a_fp8, w8, y, out_fp8, (const bf16_t*)res, f + npad, f + 3 * npad, gate, cin, hw, m_rows,
                                  kpad, npad, cin, cout, act, 1.f / a_scale, 1.f / y_scale, s);
  if (r == -2) return ofail(SPK_ERR_UNSUPPORTED, "op_pw_fp8: shape not supported (cin % 8 / 16, cout % 8)");
  if (r) return ofail(SPK_ERR_HIP, "op_pw_fp8: launch failed");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "op_pw_fp8: kernel failed");
  return SPK_OK;
}

// Depthwise conv + folded BN + activation through the LDS-staged kernel (dwconv_lds.hip; lds != 0) or the gather
// kernel (effnet.hip; lds 0), fp16 NHWC tensors, w: float32 [C][k*k].  pool_out (optional): float32 [n][C] sums of the outputs
// over each image (the squeeze-excitation pool numerator).
extern "C" int spk_op_dwconv(const void* x, const float* w, const float* bn_scale, const float* bn_bias, void* y,
                             float* pool_out, int n, int h, int wid, int c, int k, int stride, int act, int lds,
                             void* stream) {
  if (!x || !w || !bn_scale || !bn_bias || !y || n < 1 || c % 8) return ofail(SPK_ERR_ARG, "op_dwconv: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int pad = (k - 1) / 2, ho = (h + 2 * pad - k) / stride + 1, wo = (wid + 2 * pad - k) / stride + 1;
  Scratch sc;
  float* wt = sc.get<float>((size_t)k * k * c);
  int chunks = lds ? spk_dwconv_lds_chunks(0, n, h, wid, c, ho, wo, k, stride) : spk_dw_chunks(n, ho * ((wo + 3) / 4), c);
  if (lds && chunks <= 0) return ofail(SPK_ERR_UNSUPPORTED, "op_dwconv: this kernel cannot run this shape");
  float* partial = sc.get<float>((size_t)n * chunks * c);
  if (!wt || !partial) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  O_TRY(spk_launch_pack_tapmajor(w, wt, c, k * k, c, s), "pack_tapmajor");
  int r;
  if (lds)
    r = spk_launch_dwconv_lds(0, x, wt, bn_scale, bn_bias, y, partial, n, h, wid, c, ho, wo, k, stride, act, 1.f, 1.f, s);
  else
    r = spk_launch_dwconv((const bf16_t*)x, wt, bn_scale, bn_bias, (bf16_t*)y, partial, n, h, wid, c, ho, wo, k, stride, act,
                          DT_F16, s);
  if (r) return ofail(SPK_ERR_HIP, "op_dwconv: launch failed");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "op_dwconv: kernel failed");
  if (pool_out) {
    std::vector<float> hp((size_t)n * chunks * c), out((size_t)n * c, 0.f);
    if (hipMemcpy(hp.data(), partial, hp.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return ofail(SPK_ERR_HIP, "copy failed");
    for (int i = 0; i < n; ++i)
      for (int q = 0; q < chunks; ++q)
        for (int ch = 0; ch < c; ++ch) out[(size_t)i * c + ch] += hp[((size_t)i * chunks + q) * c + ch];
    if (hipMemcpy(pool_out, out.data(), out.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return ofail(SPK_ERR_HIP, "copy failed");
  }
  return SPK_OK;
}

// 1x1 convolution + folded BatchNorm (+shortcut) (+ReLU) of the eval path (conv_pw.hip) on caller-provided buffers:
// packs the fp32 weights [cout][cin] into fragment order (hi + lo images when split != 0) and
// runs configuration `cfg` of the kernel; cfg < 0: the implicit-GEMM kernel on the same operands (the reference
// point of the two-kernel tuner).  SPK_ERR_UNSUPPORTED when the configuration does not fit the problem.
extern "C" int spk_op_conv1x1(const void* x, const float* w, const float* bn_scale, const float* bn_bias, const void* res,
                              void* y, int n, int h, int wd, int cin, int cout, int stride, int relu, int split, int cfg,
                              void* stream) {
  if (!x || !w || !bn_scale || !bn_bias || !y || n < 1 || h < 1 || wd < 1 || (stride != 1 && stride != 2))
    return ofail(SPK_ERR_ARG, "op_conv1x1: bad arguments");
  if (cin % 64 || cout % 64) return ofail(SPK_ERR_UNSUPPORTED, "channels must be multiples of 64");
  hipStream_t s = (hipStream_t)stream;
  const int ho = (h - 1) / stride + 1, wo = (wd - 1) / stride + 1, M = n * ho * wo;
  // 32-bit buffer offsets in both kernels: every operand below 2 GiB (the model executor micro-batches for this)
  if ((size_t)n * h * wd * cin * 2 >= 0x80000000ull || (size_t)n * ho * wo * cout * 2 >= 0x80000000ull)
    return ofail(SPK_ERR_UNSUPPORTED, "op_conv1x1: an operand of 2 GiB or more (split the batch)");
  Scratch sc;
  bf16_t* wp = sc.get<bf16_t>((size_t)2 * cout * cin);
  if (!wp) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  int r;
  if (cfg < 0) {
    O_TRY(spk_launch_pack_weights(w, wp, cout, 1, 1, cin, CONV_MODE_GENERIC, DT_F16, split != 0, s), "pack_weights");
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.cfg = a.dma = -1;
    a.cls_ph = a.cls_pw = -1;
    a.x = (const bf16_t*)x; a.w = wp; a.y = (bf16_t*)y; a.res = (const bf16_t*)res; a.scale = bn_scale; a.bias = bn_bias;
    a.N = n; a.H = h; a.W = wd; a.Cin = cin; a.Ho = ho; a.Wo = wo; a.Cout = cout;
    a.kh = a.kw = 1; a.stride = stride; a.M = M; a.K = cin; a.relu = relu; a.dt = DT_F16; a.splitw = split != 0;
    a.x_bytes = (unsigned)((size_t)n * h * wd * cin * 2);
    a.w_bytes = (unsigned)((size_t)cout * cin * 2 * (split ? 2 : 1));
    r = spk_conv_launch(a, CONV_MODE_GENERIC, s, nullptr);
  } else {
    O_TRY(spk_launch_pack_pw(w, nullptr, wp, cout, cin, DT_F16, split ? 2 : 1, s), "pack_pw");
    PwConvArgs q;
    memset(&q, 0, sizeof q);
    q.x = (const bf16_t*)x; q.wp = wp; q.y = (bf16_t*)y; q.res = (const bf16_t*)res; q.scale = bn_scale; q.shift = bn_bias;
    q.N = n; q.H = h; q.W = wd; q.Ho = ho; q.Wo = wo; q.stride = stride; q.Cin = cin; q.Cout = cout; q.M = M;
    q.relu = relu; q.dt = DT_F16; q.nb = split ? 2 : 1;
    q.x_bytes = (unsigned)((size_t)n * h * wd * cin * 2);
    q.y_bytes = (unsigned)((size_t)M * cout * 2);
    r = spk_pw_launch(q, cfg, s);
    if (r == -3) return ofail(SPK_ERR_UNSUPPORTED, "this configuration does not fit the problem");
  }
  if (r) return ofail(SPK_ERR_HIP, "conv1x1 launch failed");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "conv1x1 kernel failed");
  return SPK_OK;
}
extern "C" int spk_op_conv1x1_num_configs(void) { return spk_pw_num_configs(); }

// A block-closing 1x1 conv and the block's 1x1 shortcut (downsample) conv as ONE K-concatenated GEMM (PwConvArgs::x2):
// y = act(BN1(W1 . x) + BN2(W2 . x2[stride2])), what the eval path runs in the first block of a ResNet stage.  The two
// eval-BatchNorm scales are folded into the concatenated fp16 weight rows (spk_launch_pw_dual_prep).  cfg: a configuration
// of spk_pw_launch (SPK_ERR_UNSUPPORTED when it does not fit the problem).
extern "C" int spk_op_conv1x1_dual(const void* x, const float* w1, const float* s1, const float* b1, const void* x2,
                                   const float* w2, const float* s2, const float* b2, void* y, int n, int ho, int wo, int cin,
                                   int h2, int w2d, int cin2, int cout, int stride2, int relu, int split, int cfg,
                                   void* stream) {
  if (!x || !w1 || !s1 || !b1 || !x2 || !w2 || !s2 || !b2 || !y || n < 1 || ho < 1 || wo < 1 || stride2 < 1)
    return ofail(SPK_ERR_ARG, "op_conv1x1_dual: bad arguments");
  if (cin % 64 || cin2 % 64 || cout % 64) return ofail(SPK_ERR_UNSUPPORTED, "channels must be multiples of 64");
  if ((ho - 1) * stride2 >= h2 || (wo - 1) * stride2 >= w2d) return ofail(SPK_ERR_ARG, "op_conv1x1_dual: second source too small");
  hipStream_t s = (hipStream_t)stream;
  const int M = n * ho * wo, K = cin + cin2;
  if ((size_t)M * cin * 2 >= 0x80000000ull || (size_t)n * h2 * w2d * cin2 * 2 >= 0x80000000ull || (size_t)M * cout * 2 >= 0x80000000ull)
    return ofail(SPK_ERR_UNSUPPORTED, "op_conv1x1_dual: an operand of 2 GiB or more");
  Scratch sc;
  float* wcat = sc.get<float>((size_t)cout * K);
  float* sb = sc.get<float>((size_t)2 * cout);
  bf16_t* wp = sc.get<bf16_t>((size_t)2 * cout * K);
  if (!wcat || !sb || !wp) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  O_TRY(spk_launch_pw_dual_prep(w1, w2, s1, s2, b1, b2, wcat, sb, sb + cout, cout, cin, cin2, s), "pw_dual_prep");
  O_TRY(spk_launch_pack_pw(wcat, nullptr, wp, cout, K, DT_F16, split ? 2 : 1, s), "pack_pw");
  PwConvArgs q;
  memset(&q, 0, sizeof q);
  q.x = (const bf16_t*)x; q.wp = wp; q.y = (bf16_t*)y; q.scale = sb; q.shift = sb + cout;
  q.N = n; q.H = ho; q.W = wo; q.Ho = ho; q.Wo = wo; q.stride = 1; q.Cin = cin; q.Cout = cout; q.M = M;
  q.relu = relu; q.dt = DT_F16; q.nb = split ? 2 : 1;
  q.x_bytes = (unsigned)((size_t)M * cin * 2);
  q.y_bytes = (unsigned)((size_t)M * cout * 2);
  q.x2 = (const bf16_t*)x2; q.Cin2 = cin2; q.H2 = h2; q.W2 = w2d; q.stride2 = stride2;
  q.x2_bytes = (unsigned)((size_t)n * h2 * w2d * cin2 * 2);
  const int r = spk_pw_launch(q, cfg, s);
  if (r == -3) return ofail(SPK_ERR_UNSUPPORTED, "this configuration does not fit the problem");
  if (r) return ofail(SPK_ERR_HIP, "dual conv1x1 launch failed");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "dual conv1x1 kernel failed");
  return SPK_OK;
}

// Two chained 1x1 convs in one launch (conv_pw.hip, PwConvArgs::wpz): y = act(BN(W . x) + res) with 256 couts, then
// z = actz(BNz(Wz . y)) from the output tile in registers; single fp16 weight images.  What the eval path runs for a
// bottleneck's block-closing conv and the next block's first conv.  SPK_ERR_UNSUPPORTED: no chained kernel for the shape.
extern "C" int spk_op_conv1x1_chain(const void* x, const float* w, const float* bn_scale, const float* bn_bias, const void* res,
                                    void* y, const float* wz, const float* bnz_scale, const float* bnz_bias, void* z, int n,
                                    int h, int wd, int cin, int cout, int coutz, int relu, int reluz, void* stream) {
  if (!x || !w || !bn_scale || !bn_bias || !res || !y || !wz || !bnz_scale || !bnz_bias || !z || n < 1 || h < 1 || wd < 1)
    return ofail(SPK_ERR_ARG, "op_conv1x1_chain: bad arguments");
  if (cin % 64 || cout % 64 || coutz % 32) return ofail(SPK_ERR_UNSUPPORTED, "channels must be multiples of 64");
  hipStream_t s = (hipStream_t)stream;
  const int M = n * h * wd;
  if ((size_t)M * cin * 2 >= 0x80000000ull || (size_t)M * cout * 2 >= 0x80000000ull)
    return ofail(SPK_ERR_UNSUPPORTED, "op_conv1x1_chain: an operand of 2 GiB or more");
  Scratch sc, scz;
  bf16_t* wp = sc.get<bf16_t>((size_t)cout * cin);
  bf16_t* wpz = scz.get<bf16_t>((size_t)coutz * cout);
  if (!wp || !wpz) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  O_TRY(spk_launch_pack_pw(w, nullptr, wp, cout, cin, DT_F16, 1, s), "pack_pw");
  O_TRY(spk_launch_pack_pw(wz, nullptr, wpz, coutz, cout, DT_F16, 1, s), "pack_pw");
  PwConvArgs q;
  memset(&q, 0, sizeof q);
  q.x = (const bf16_t*)x; q.wp = wp; q.y = (bf16_t*)y; q.res = (const bf16_t*)res; q.scale = bn_scale; q.shift = bn_bias;
  q.N = n; q.H = h; q.W = wd; q.Ho = h; q.Wo = wd; q.stride = 1; q.Cin = cin; q.Cout = cout; q.M = M;
  q.relu = relu; q.dt = DT_F16; q.nb = 1;
  q.x_bytes = (unsigned)((size_t)M * cin * 2);
  q.y_bytes = (unsigned)((size_t)M * cout * 2);
  q.wpz = wpz; q.z = (bf16_t*)z; q.scalez = bnz_scale; q.shiftz = bnz_bias; q.Coutz = coutz; q.reluz = reluz;
  q.z_bytes = (unsigned)((size_t)M * coutz * 2);
  const int r = spk_pw_chain_launch(q, s);
  if (r == -3) return ofail(SPK_ERR_UNSUPPORTED, "no chained kernel for this problem");
  if (r) return ofail(SPK_ERR_HIP, "chained conv1x1 launch failed");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "chained conv1x1 kernel failed");
  return SPK_OK;
}

// 3x3 stride-1 pad-1 convolution + folded BatchNorm (+ReLU) of the eval path (conv_c3.hip) on caller-provided buffers;
// cfg >= 0: that tile configuration (SPK_ERR_UNSUPPORTED when it does not fit), cfg < 0: the implicit-GEMM kernel.
extern "C" int spk_op_conv3x3(const void* x, const float* w_ohwi, const float* bn_scale, const float* bn_bias,
                              const void* res, void* y, int n, int h, int wd, int cin, int cout, int relu, int split,
                              int cfg, void* stream) {
  if (!x || !w_ohwi || !bn_scale || !bn_bias || !y || n < 1 || h < 1 || wd < 1)
    return ofail(SPK_ERR_ARG, "op_conv3x3: bad arguments");
  if (cin % 64 || cout % 64) return ofail(SPK_ERR_UNSUPPORTED, "channels must be multiples of 64");
  hipStream_t s = (hipStream_t)stream;
  const int M = n * h * wd;
  if ((size_t)n * h * wd * cin * 2 >= 0x80000000ull || (size_t)n * h * wd * cout * 2 >= 0x80000000ull)
    return ofail(SPK_ERR_UNSUPPORTED, "op_conv3x3: an operand of 2 GiB or more (split the batch)");
  Scratch sc;
  bf16_t* wp = sc.get<bf16_t>((size_t)2 * cout * 9 * cin);
  if (!wp) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  int r;
  if (cfg < 0) {
    O_TRY(spk_launch_pack_weights(w_ohwi, wp, cout, 3, 3, cin, CONV_MODE_GENERIC, DT_F16, split != 0, s), "pack_weights");
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.cfg = a.dma = -1;
    a.cls_ph = a.cls_pw = -1;
    a.x = (const bf16_t*)x; a.w = wp; a.y = (bf16_t*)y; a.scale = bn_scale; a.bias = bn_bias;
    a.res = (const bf16_t*)res;
    a.N = n; a.H = h; a.W = wd; a.Cin = cin; a.Ho = h; a.Wo = wd; a.Cout = cout;
    a.kh = a.kw = 3; a.stride = 1; a.pad = 1; a.M = M; a.K = 9 * cin; a.relu = relu; a.dt = DT_F16; a.splitw = split != 0;
    a.x_bytes = (unsigned)((size_t)M * cin * 2);
    a.w_bytes = (unsigned)((size_t)cout * 9 * cin * 2 * (split ? 2 : 1));
    r = spk_conv_launch(a, CONV_MODE_GENERIC, s, nullptr);
  } else {
    O_TRY(spk_launch_pack_c3(w_ohwi, wp, cout, cin, split ? 2 : 1, s), "pack_c3");
    C3Args q;
    memset(&q, 0, sizeof q);
    q.x = (const bf16_t*)x; q.wp = wp; q.y = (bf16_t*)y; q.scale = bn_scale; q.shift = bn_bias;
    q.N = n; q.H = h; q.W = wd; q.Cin = cin; q.Cout = cout; q.M = M; q.relu = relu; q.dt = DT_F16; q.nb = split ? 2 : 1;
    q.x_bytes = (unsigned)((size_t)M * cin * 2);
    q.y_bytes = (unsigned)((size_t)M * cout * 2);
    q.wp_bytes = (unsigned)((size_t)cout * 9 * cin * 2 * q.nb);
    q.res = (const bf16_t*)res;
    r = spk_c3_launch(q, cfg, s);
    if (r == -3) return ofail(SPK_ERR_UNSUPPORTED, "this configuration does not fit the problem");
  }
  if (r) return ofail(SPK_ERR_HIP, "conv3x3 launch failed");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "conv3x3 kernel failed");
  return SPK_OK;
}
extern "C" int spk_op_conv3x3_num_configs(void) { return spk_c3_num_configs(); }

// A whole identity bottleneck block of the eval path on caller-provided buffers: x, y [n,h,w,4 cm] fp16; fp32 weights
// w1 [cm][4 cm], w2 [cm][3][3][cm], w3 [4 cm][cm]; folded BatchNorm scale / shift per conv.  fused != 0: the one-kernel
// form (conv_bneck.hip; SPK_ERR_UNSUPPORTED when it has no instantiation for the shape); fused == 0: the same block as three
// launches of the eval path's own kernels (conv_pw.hip, conv_c3.hip, conv_pw.hip with the shortcut), mid tensors in
// scratch - the reference point of the parity test and of the tuner.  iters > 0: the launches are repeated and *ms_out
// receives the mean time of one block (events on `stream`).
extern "C" int spk_op_bottleneck(const void* x, const float* w1, const float* w2, const float* w3, const float* s1,
                                 const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, void* y,
                                 int n, int h, int wd, int cm, int fused, int iters, float* ms_out, void* stream,
                                 unsigned long long* stamps_dev, const float* wz, const float* sz, const float* bz, void* z,
                                 int coutz) {
  if (!x || !w1 || !w2 || !w3 || !s1 || !b1 || !s2 || !b2 || !s3 || !b3 || !y || n < 1 || h < 1 || wd < 1 || x == y)
    return ofail(SPK_ERR_ARG, "op_bottleneck: bad arguments");
  if (cm % 64) return ofail(SPK_ERR_UNSUPPORTED, "mid channels must be a multiple of 64");
  hipStream_t s = (hipStream_t)stream;
  const int c4 = 4 * cm, M = n * h * wd;
  if ((size_t)M * c4 * 2 >= 0x80000000ull) return ofail(SPK_ERR_UNSUPPORTED, "op_bottleneck: an operand of 2 GiB or more");
  Scratch sc;
  bf16_t* p1 = sc.get<bf16_t>((size_t)cm * c4);
  bf16_t* p2 = sc.get<bf16_t>((size_t)9 * cm * cm);
  bf16_t* p3 = sc.get<bf16_t>((size_t)cm * c4);
  bf16_t* y1 = sc.get<bf16_t>((size_t)M * cm);
  bf16_t* y2 = sc.get<bf16_t>((size_t)M * cm);
  bf16_t* pz = sc.get<bf16_t>((size_t)(wz ? coutz : 1) * c4);
  if (!p1 || !p2 || !p3 || !y1 || !y2 || !pz) return ofail(SPK_ERR_HIP, "hipMalloc failed");
  if (wz && (!sz || !bz || !z || coutz % 32)) return ofail(SPK_ERR_ARG, "op_bottleneck: bad chained-conv arguments");
  if (wz) O_TRY(spk_launch_pack_pw(wz, nullptr, pz, coutz, c4, DT_F16, 1, s), "pack_pw");
  O_TRY(spk_launch_pack_pw(w1, nullptr, p1, cm, c4, DT_F16, 1, s), "pack_pw");
  O_TRY(spk_launch_pack_c3(w2, p2, cm, cm, 1, s), "pack_c3");
  O_TRY(spk_launch_pack_pw(w3, nullptr, p3, c4, cm, DT_F16, 1, s), "pack_pw");
  BneckArgs a;
  memset(&a, 0, sizeof a);
  a.x = (const bf16_t*)x; a.y = (bf16_t*)y; a.w1 = p1; a.w2 = p2; a.w3 = p3;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3;
  a.N = n; a.H = h; a.W = wd; a.C4 = c4; a.CM = cm; a.x_bytes = (unsigned)((size_t)M * c4 * 2);
  a.stamps = stamps_dev;
  a.y1 = y1;
  if (wz) { a.wz = pz; a.z = (bf16_t*)z; a.sz = sz; a.bz = bz; a.Coutz = coutz; a.z_bytes = (unsigned)((size_t)M * coutz * 2); }
  auto pw = [&](const bf16_t* in, const bf16_t* wp, bf16_t* out, const bf16_t* res, const float* sc_, const float* sh_, int cin,
                int cout) {
    PwConvArgs q;
    memset(&q, 0, sizeof q);
    q.x = in; q.wp = wp; q.y = out; q.res = res; q.scale = sc_; q.shift = sh_;
    q.N = n; q.H = h; q.W = wd; q.Ho = h; q.Wo = wd; q.stride = 1; q.Cin = cin; q.Cout = cout; q.M = M;
    q.relu = 1; q.dt = DT_F16; q.nb = 1;
    q.x_bytes = (unsigned)((size_t)M * cin * 2);
    q.y_bytes = (unsigned)((size_t)M * cout * 2);
    for (int cfg : {7, 9, 11, 5, 1, 3, 0, 4})    // (every configuration gives the same bits: tests/test_gpu_pw.py)
      if (const int r = spk_pw_launch(q, cfg, s); r != -3) return r;
    return -3;
  };
  auto once = [&]() -> int {
    if (fused == 1) return wz ? -3 : spk_bneck_launch(a, s);
    if (const int r = pw((const bf16_t*)x, p1, y1, nullptr, s1, b1, c4, cm)) return r;
    if (fused == 2) return spk_btail_launch(a, s);     // conv2 + conv3 + shortcut (+ the chained conv) in one kernel
    C3Args q;
    memset(&q, 0, sizeof q);
    q.x = y1; q.wp = p2; q.y = y2; q.scale = s2; q.shift = b2;
    q.N = n; q.H = h; q.W = wd; q.Cin = cm; q.Cout = cm; q.M = M; q.relu = 1; q.dt = DT_F16; q.nb = 1;
    q.x_bytes = q.y_bytes = (unsigned)((size_t)M * cm * 2);
    q.wp_bytes = (unsigned)((size_t)9 * cm * cm * 2);
    int r = -3;
    for (int cfg : {2, 0, 10, 8, 9, 3, 7})       // (likewise: tests/test_gpu_c3.py)
      if ((r = spk_c3_launch(q, cfg, s)) != -3) break;
    if (r) return r;
    if ((r = pw(y2, p3, (bf16_t*)y, (const bf16_t*)x, s3, b3, cm, c4)) != 0 || !wz) return r;
    return pw((const bf16_t*)y, pz, (bf16_t*)z, nullptr, sz, bz, c4, coutz);
  };
  int r = once();
  if (r == -3) return ofail(SPK_ERR_UNSUPPORTED, "no kernel for this bottleneck shape");
  if (r) return ofail(SPK_ERR_HIP, "bottleneck launch failed");
  if (iters > 0 && ms_out) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return ofail(SPK_ERR_HIP, "hipEventCreate failed");
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters && !r; ++i) r = once();
    (void)hipEventRecord(e1, s);
    float ms = 0.f;
    const bool ok = !r && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) return ofail(SPK_ERR_HIP, "bottleneck timing failed");
    *ms_out = ms / iters;
  }
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "bottleneck kernel failed");
  return SPK_OK;
}

// Zero-sum rounding of fp32 weight rows to fp16 values (zero_sum.hip) on caller-provided device buffers: w, out
// [rows][row_len] fp32, mu [mu_period] fp32 or null.  The kernel spk_commit runs per conv in the calibrated mode.
extern "C" int spk_op_zero_sum_round(const float* w, const float* mu, float* out, int64_t rows, int row_len, int mu_period,
                                     void* stream) {
  if (!w || !out || rows < 1 || row_len < 1 || mu_period < 1) return ofail(SPK_ERR_ARG, "op_zero_sum_round: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int r = spk_launch_zero_sum_round(w, mu, out, (size_t)rows, row_len, mu_period, s);
  if (r == -2) return ofail(SPK_ERR_UNSUPPORTED, "op_zero_sum_round: row too long (8192 elements at most)");
  if (r) return ofail(SPK_ERR_HIP, "zero-sum rounding launch failed");
  if (hipStreamSynchronize(s) != hipSuccess) return ofail(SPK_ERR_HIP, "zero-sum rounding kernel failed");
  return SPK_OK;
}
