// Model handle internals (host side).  Not part of the public ABI.
#pragma once
#include "../../include/sykepic_hip.h"
#include "spk_common.h"

#include <string>
#include <unordered_map>
#include <vector>

enum ParamKind { PK_CONV_W = 1, PK_BN_W, PK_BN_B, PK_BN_MEAN, PK_BN_VAR, PK_BN_NBT, PK_FC_W, PK_FC_B,
                 PK_SE_W, PK_SE_B };

struct Param {
  std::string key;
  int kind = 0, layer = -1, dtype = 0, ndim = 0;
  int64_t shape[4] = {1, 1, 1, 1};
  int64_t numel = 0;
  bool trainable = false;  // nn.Parameter (true) vs buffer (false)
  int requires_grad = 0;
  int group = -1;          // optimizer param group, -1: not in the optimizer
  size_t off = 0;          // element offset in the flat fp32 buffers
  long step = 0;           // per-tensor step count (torch keeps it per parameter)
  double mu_product = 1.0; // NAdam: running product of the momentum schedule
};

struct TDim {
  int h = 0, w = 0, c = 0;  // c: channels as laid out (padded to 64 for EfficientNet widths)
  bool bf16 = true;
  int c_log = 0;            // logical channels (0: same as c)
};

struct Layer {
  spk_layer_desc d;
  int mode = 0;     // CONV_MODE_*
  int kpad = 0;     // GEMM K of the packed weights
  int p_w = -1, p_g = -1, p_b = -1, p_mean = -1, p_var = -1, p_nbt = -1;
  int p_w2 = -1, p_b2 = -1;       // SE: fc2 (p_w / p_b hold fc1)
  int cin_p = 0, cout_p = 0;      // channels as laid out in HBM: padded to a multiple of 64 (zeros)
  size_t wpack_off = 0, sb_off = 0;
  size_t wpw_off = 0;             // fragment-ordered image of a 1x1 conv's weights (conv_pw.hip), when pw_ok
  bool pw_ok = false;             // 1x1, stride 1 or 2, no padding anywhere, channels multiples of 64
  bool c3_ok = false;             // 3x3 stride 1 pad 1, cin % 64 == 0, cout % 256 == 0: conv_c3.hip (image at wpw_off)
  // eval: a block-closing 1x1 conv and the 1x1 shortcut (downsample) conv of its block as ONE K-concatenated GEMM
  // (conv_pw.hip, PwConvArgs::x2): the shortcut tensor is neither written nor re-read
  int dual_src = -1;              // block-closing conv: index of the shortcut conv it absorbs, or -1
  int fused_into = -1;            // shortcut conv: index of the block-closing conv that absorbs it, or -1
  bool dual_ok = false;           // images packed for the current precision settings (spk_commit)
  size_t wdual_off = 0;           // fragment-ordered concatenated weights (elements into spk_model::wdual)
  size_t sdual_off = 0;           // [2^e per cout][summed shifts] (floats into spk_model::sdual)
  // eval, single-weight modes: a block-closing 1x1 conv (256 couts) and the 1x1 conv of the NEXT block that reads its
  // output, as one kernel (conv_pw.hip, PwConvArgs::wpz): the trunk is written but not re-read by that conv
  int chain_next = -1;            // block-closing conv: index of the conv it can also compute, or -1
  int chained_by = -1;            // that conv: index of the block-closing conv
  bool chained_now_h[2] = {false, false};   // set by the block-closing conv's launch of the current forward, per half-batch
                                  // chain (index = spk_model::half): chain_next is done for THAT half
  // eval, single-weight modes: a whole identity bottleneck block - conv1 1x1 -> conv2 3x3 -> conv3 1x1 + shortcut, ReLU behind
  // each BatchNorm - as ONE kernel (conv_bneck.hip): the two mid tensors are never written
  int bn_c2 = -1, bn_c3 = -1;     // the block's conv1: indices of its conv2 / conv3, or -1
  int bn_head = -1;               // conv2 / conv3 of such a block: index of its conv1
  bool bneck_now_h[2] = {false, false};   // conv1: the one-kernel form ran for this half-batch chain of the current forward
  bool btail_now_h[2] = {false, false};   // conv2: its launch also computed conv3 + shortcut (+ the chained conv): conv_btail_kernel
  size_t mu_off = 0;              // generic convs: offset of this layer's cin input-channel means in spk_model::act_mean
  // fp16 eval of an MBConv block: the squeeze-excitation layer computes its gates only and the project 1x1 conv that is
  // the sole reader of its output multiplies them into its activation operand (conv_igemm.hip, spk_set_gate)
  int gate_conv = -1;             // squeeze-excitation layer: index of that conv, or -1
  int gate_from = -1;             // that conv: index of the squeeze-excitation layer
  int64_t nbt = 0;  // num_batches_tracked (host copy; exact int64)
  bool trunk_writer = false;  // output is (or is added to) the residual trunk: stem, block-closing conv, downsample
  bool inner3x3 = false;      // 3x3 conv in the middle of a bottleneck block (reads a 1x1 conv's output)
  // fp8 mode of the MBConv interior (spk_model_set_fp8): role of this layer and what calibration measured
  int fp8_role = 0;           // 0 none, 1 expand conv (fp16 in, e4m3 out), 2 depthwise (e4m3 in/out), 3 squeeze-excitation
                              // (gates only), 4 project conv (e4m3 in x gate, fp16 out)
  float amax_in = 0.f, amax_out = 0.f;   // max |x| of the input / output tensor on the calibration batch
  size_t w8_off = 0, s8_off = 0;         // e4m3 weights / [ws, epilogue scale] floats of this layer
  bool side_branch = false;   // output is only ever a shortcut operand (downsample conv): what the block-closing conv can absorb
  int fuse_pool = -1;         // stem conv: index of the 3x3/2 max-pool layer its eval kernel can absorb (conv_stem.hip), or -1
  bool pooled_by_stem = false;  // that max-pool layer
};

struct TrainState;

struct spk_model {
  int device = 0;
  int in_chans = 3, num_classes = 0;
  hipStream_t stream = nullptr;
  // eval: the two halves of a batch on two streams (see spk_forward_eval_logits)
  hipStream_t half_stream = nullptr;
  hipEvent_t half_fork = nullptr, half_join = nullptr;
  std::vector<long long> half_warm;   // (n, h, w) keys whose half-batch kernels have been tuned (on a quiet GPU)
  std::vector<Layer> layers;
  std::vector<Param> params;
  std::unordered_map<std::string, int> index;
  int n_tensors = 1;

  float* pbuf = nullptr;       // flat fp32 master parameters + buffers
  size_t n_flat = 0, n_train = 0;
  bf16_t* wpack = nullptr;     // bf16 conv weights, [Cout][K] per layer
  float* scale_bias = nullptr; // eval-BN folded scale/bias per conv
  float* dwpack = nullptr;     // fp32 tap-major weights of depthwise / 3x3-stem layers (EfficientNet)
  bf16_t* wdual = nullptr;     // K-concatenated weight images of the fused (block-closing + shortcut) convs
  float* sdual = nullptr;      // their epilogue factors
  bool fuse_ds = true;         // SPK_FUSE_DS=0 turns the fusion off
  bool fuse_se = true;         // fp16 eval: squeeze-excitation scaling inside the project conv (SPK_SE_FUSE=0: its own pass)
  int chain = 1;               // conv3 -> next conv1 chaining: 1 where it is faster (timed once per problem), SPK_CHAIN=0 never, 2 always
  bool no_chain_now = false;   // the chain tuner is timing the two-kernel alternative
  int bneck = 1;               // whole-bottleneck kernel: 1 where it is faster (rule + timing: bneck_choice), SPK_BNECK=0 never, 2 always
                               // (14-row blocks), 3 always (7-row blocks)
  bool two_streams_now = false;   // the executor is enqueueing the two halves of a batch on two streams
  bool no_bneck_now = false;   // its tuner is timing the three-launch alternative
  int btail = 0;               // conv2 + conv3 (+ chained conv) kernel on the stage the whole-block kernel does not take: off - it is
                               // bit-identical but no faster than the launches it replaces; SPK_BTAIL=1
  std::vector<char> stale;     // per tensor: the last eval forward did not write it (a fused-away shortcut tensor)
  bool effnet = false;         // EfficientNet graph (widths that are not multiples of 64, depthwise / SE / SiLU ops): its
                               // TRAINING plan pads every activation tensor to a multiple of 64 channels (train_effnet.hip)
  bool plan_pad = false;       // the current activation plan is the channel-padded one
  size_t se_off = 0;           // arena offset of the squeeze-excitation scratch (partials + scales)
  bool dirty = true;
  int infer_dt = DT_F16;       // 16-bit storage type of the eval path
  int packed_dt = -1;          // dtype the packed weights currently hold
  int packed_split = -1;
  int splitw = 3;              // eval: hi+lo fp16 weights. 0 none, 1 every conv, 2 trunk writers only, 3 all but inner 3x3 (default), 4 per-op mask,
                               // 5 calibrated single pass: the stem only, every other conv zero-sum rounded (zero_sum.hip)
  // zero-sum rounding of the un-split fp16 weights against per-channel activation means (spk_model_calibrate_act_means)
  bool zero_sum = false;       // explicit switch (spk_model_set_zero_sum); splitw == 5 implies it
  bool have_means = false;
  size_t n_means = 0;          // sum of cin over the generic convs
  std::vector<float> act_mean; // host copy, graph order
  float* act_mean_dev = nullptr;
  std::vector<double> cal_sum; // calibration accumulators: sum over rows per channel ...
  std::vector<double> cal_rows;  // ... and rows seen, per generic conv (graph order)
  int packed_zs = -1;
  std::vector<unsigned char> split_mask;  // splitw == 4: one flag per graph op
  int split_epoch = 0, packed_epoch = -1;  // bumps when the mask changes
  int act_dt = DT_F16;         // dtype of the activations now in the arena

  // activations of one (n,h,w) plan
  void* arena = nullptr;
  size_t arena_bytes = 0;
  int cap_n = 0, cap_h = 0, cap_w = 0;
  int mb_limit = 1;
  std::vector<TDim> tdims;
  std::vector<size_t> toff;
  std::vector<size_t> toff_lo;  // 0 = no remainder tensor
  bool precise_res = false;     // keep the 16-bit rounding remainder of shortcut tensors
  size_t logits_off = 0;

  // fp8 (e4m3) eval mode of the EfficientNet MBConv interior — BASELINE config 5
  int fp8 = 0;                  // requested
  bool fp8_calibrated = false;  // roles assigned + activation ranges measured
  bool fp8_packed = false;
  std::vector<unsigned char> fp8_blocks;   // per qualifying MBConv block (graph order): on the e4m3 path?  empty: all
  unsigned char* w8pack = nullptr;
  float* s8 = nullptr;
  unsigned char* fp8_shadow = nullptr;  // fp8 mode: e4m3 copy of the trunk tensor the last project conv wrote (pw_fp8.hip)
  size_t fp8_shadow_bytes = 0;
  size_t fp8_shadow_img = 0;            // bytes per image of the shadow: the largest h * w * stride of any shadowed tensor; a chunk
                                        // [img0, ...) starts at img0 * fp8_shadow_img in every block (disjoint slices per half-batch)
  // state one layer leaves for the next, per half-batch chain of the two-stream forward (index = spk_model::half)
  int half = 0;                      // which of the two chains the executor is enqueueing (0: the caller's stream)
  int dw_chunks_h[2] = {0, 0};       // pool-partial rows per image the depthwise layer that ran last wrote
  const float* gate_h[2] = {nullptr, nullptr};   // gates of the squeeze-excitation op that ran last (consumed by the project conv)
  int gate_stride_h[2] = {0, 0};
  int shadow_t_h[2] = {-1, -1};      // fp8 mode: tensor id whose e4m3 copy the last project conv left (-1: none valid)
  int shadow_stride_h[2] = {0, 0};   // ... and its row stride in bytes
  size_t se_stride = 0;              // floats of squeeze-excitation scratch per image (the halves use disjoint slices)
  std::vector<float> t_fp8_scale;    // per tensor: 0 = 16-bit storage, else value = byte * scale

  // data-parallel overlap (spk_model_set_grad_ready_callback)
  spk_grad_ready_fn grad_cb = nullptr;
  void* grad_cb_user = nullptr;
  hipStream_t comm_stream = nullptr;
  int grad_buckets = 0;
  hipEvent_t grad_ev[3] = {nullptr, nullptr, nullptr};

  uint64_t seed = 0;
  float bn_eps = 1e-5f, bn_momentum = 0.1f;   // BatchNorm2d(eps, momentum) of the graph (spk_model_set_bn)
  TrainState* train = nullptr;

  int img0 = 0;                 // first image of the chunk the eval executor is working on (prefix micro-batching)
  bool fuse_stem_pool = true;   // eval: stem conv + max-pool in one kernel (SPK_FUSE_STEM_POOL=0 turns it off)
  int stale_stem_t = -1;        // tensor id of the stem output the last eval forward did NOT write (fused), or -1
  bool force_unfused = false;   // read_activation is recomputing the stem output
  int last_eval_nb = 0;         // images of the last eval micro-batch (read_activation recomputes a stale tensor)
  float* P(int pi) const { return pbuf + params[pi].off; }
  void* T(int t) const { return (char*)arena + toff[t]; }
  // tensor t from image img0 on
  void* TI(int t) const {
    const TDim& d = tdims[t];
    return (char*)arena + toff[t] + (size_t)img0 * d.h * d.w * d.c * (d.bf16 ? 2 : 4);
  }
  // ... when it holds e4m3 bytes (fp8 mode: one byte per element in a slot sized for two)
  void* TI8(int t) const {
    const TDim& d = tdims[t];
    return (char*)arena + toff[t] + (size_t)img0 * d.h * d.w * d.c;
  }
  // squeeze-excitation scratch of the chunk in work
  float* SE() const { return (float*)((char*)arena + se_off) + (size_t)img0 * se_stride; }
  void* TLo(int t) const { return toff_lo[t] ? (char*)arena + toff_lo[t] : nullptr; }
};

void spk_set_error(const std::string& s);
int spk_commit(spk_model* m);
int spk_train_join(spk_model* m);   // train.hip: the model's stream waits for the last step's deferred side-stream work
int spk_plan(spk_model* m, int n, int h, int w, bool pad = false);
int spk_run_layer_eval(spk_model* m, Layer& L, int nb);
int spk_forward_eval_logits(spk_model* m, const void* x, int n, int h, int w, int layout, int dtype,
                            float* logits_dev);
int spk_read_flat(spk_model* m, const float* flat, const Param& p, float* host);
void spk_train_free(spk_model* m);
// backward of one convolution (train.hip): shared by the training step and the single-operator test hooks
// fuse: (stride 1 only) the BatchNorm-backward reduction of the layer that produced the tensor dx is the gradient of,
// emitted by the dgrad epilogue (ConvArgs::bnb_raw); `tiles` returns the number of [2][cin] partial rows written
struct BnbFuse {
  const bf16_t* raw;
  const unsigned char* mask;   // null: no ReLU behind that BatchNorm
  const float* mean;
  const float* invstd;
  float* partials;
  int tiles;
  // optional: the shortcut gradient is taken from its source instead of from dx (dx is then overwritten, not accumulated):
  // res_src = output gradient of the block-closing conv whose shortcut this tensor is, res_bits = that conv's ReLU bits
  const bf16_t* res_src;
  const unsigned char* res_bits;
};
int spk_conv_dgrad_all(const bf16_t* dy, const bf16_t* wdg, bf16_t* dx, bool accumulate, int n, int oh, int ow,
                       int cout, int ih, int iw, int cin, int k, int stride, int pad, hipStream_t s,
                       BnbFuse* fuse = nullptr);
size_t spk_conv_wgrad_slab_floats(int M, int cin, int cout, int k, bool stem);
int spk_conv_wgrad_slabs(const bf16_t* x, const bf16_t* dy, float* slabs, int n, int ih, int iw, int cin, int oh,
                         int ow, int cout, int k, int stride, int pad, bool stem, hipStream_t s);
int spk_conv_wgrad_reduce(const float* slabs, float* gw, int M, int cin, int cout, int k, bool stem, hipStream_t s,
                          float scale = 1.0f);
void spk_train_mark_dirty(spk_model* m);
