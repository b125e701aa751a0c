/*
 * sykepic_hip.h — C-ABI of libsykepic_hip.so: the MI355X (gfx950) replacement
 * for the CNN classification hot path of sykefi/syke-pic.
 *
 * The reference has no FFI: its seam is Python duck-typing on three objects,
 * each created in exactly one place (paths relative to
 * /root/reference):
 *   - the network          sykepic/train/config.py:63-77  (TorchVisionNet(...))
 *   - the loss             sykepic/train/train.py:127     (nn.CrossEntropyLoss())
 *   - the optimizer        sykepic/train/train.py:131-138 (getattr(optim, name)([...]))
 * and everything heavy happens at two call sites:
 *   - inference  `out = net(x)` + base-1.3 softmax   sykepic/compute/probability.py:189-194
 *   - training   zero_grad/forward/loss/backward/step sykepic/train/train.py:239-243
 * Each entry point below names the reference lines it stands in for.  The
 * Python adapter (syke-pic_amd/sykepic_hip/lib.py) binds them with ctypes.
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer it
 * passes; the library owns weights, activations, gradients and workspace
 * inside the handle; no exception crosses the boundary (0 = ok, <0 = error,
 * text via spk_last_error()).  Host-facing parameter I/O always uses the
 * reference's on-disk layout (NCHW / [out,in] float32, int64 counters) so a
 * best_state.pth stays interchangeable.  Device pointers are HIP device
 * memory on the model's device; work is enqueued on the model's stream
 * (spk_model_set_stream) and is asynchronous unless stated.
 */
#ifndef SYKEPIC_HIP_H
#define SYKEPIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPK_OK 0
#define SPK_ERR_ARG (-1)
#define SPK_ERR_HIP (-2)
#define SPK_ERR_KEY (-3)
#define SPK_ERR_UNSUPPORTED (-4)
#define SPK_ERR_STATE (-5)

/* layer kinds of spk_layer_desc.kind */
#define SPK_OP_CONV 1      /* Conv2d(bias=False) + BatchNorm2d (+residual) (+ReLU) */
#define SPK_OP_MAXPOOL 2   /* MaxPool2d(k, stride, pad) */
#define SPK_OP_GAVGPOOL 3  /* AdaptiveAvgPool2d(1) + flatten */
#define SPK_OP_LINEAR 4    /* Linear with bias, no activation (head) */
#define SPK_OP_DROPOUT 5   /* Dropout(p) in the head */
/* EfficientNet (torchvision MBConv; eval and, since round 3, training):
 *   DWCONV: depthwise Conv2d(C, C, k, stride, pad=(k-1)/2, groups=C, bias=False) + BatchNorm2d + activation
 *   SE:     SqueezeExcitation(C, squeeze = `k`): avgpool -> fc1 (1x1 conv, bias) -> SiLU -> fc2 -> Sigmoid -> x * s;
 *           `name` is the module prefix (name.fc1.weight ...), cin = cout = C. */
#define SPK_OP_DWCONV 6
#define SPK_OP_SE 7

/* spk_layer_desc.relu: activation after BatchNorm (+ residual) */
#define SPK_ACT_NONE 0
#define SPK_ACT_RELU 1
#define SPK_ACT_SILU 2

/* input layouts / dtypes of the image batch */
#define SPK_LAYOUT_NCHW 0
#define SPK_LAYOUT_NHWC 1
#define SPK_DTYPE_F32 0
#define SPK_DTYPE_I64 1
#define SPK_DTYPE_U8 2

/* optimizers (reference: `getattr(optim, name)` with only `lr` given, sykepic/train/train.py:131-138; Adam is the
 * default, train.ini.example:74): the first-order torch.optim classes with their torch default hyper-parameters */
#define SPK_OPT_SGD 0
#define SPK_OPT_ADAM 1
#define SPK_OPT_ADAMW 2     /* decoupled weight decay (default 0.01) */
#define SPK_OPT_RMSPROP 3   /* alpha 0.99, eps 1e-8, optional momentum; centered = False */
#define SPK_OPT_ADAGRAD 4   /* lr_decay, initial_accumulator_value, eps 1e-10 */
#define SPK_OPT_ADAMAX 5
#define SPK_OPT_NADAM 6     /* momentum_decay 4e-3 */
#define SPK_OPT_RADAM 7
#define SPK_OPT_ADADELTA 8  /* rho 0.9, eps 1e-6 */
#define SPK_OPT_ASGD 9      /* lambd 1e-4 (field lr_decay), alpha 0.75 (field alpha), t0 1e6; the averaged copy `ax` is not kept */
#define SPK_OPT_RPROP 10    /* etas (0.5, 1.2) in beta1 / beta2, step sizes (1e-6, 50) in eps / alpha */

typedef struct spk_model spk_model;

/* One node of the layer graph that TorchVisionNet.__init__ builds
 * (sykepic/train/network.py:48-63): backbone children minus the last one,
 * then the Linear head.  `name`/`bn` are state_dict prefixes ("base.4.0.conv1",
 * "base.4.0.bn1", "head.0"). src/dst/res are activation ids (0 = input). */
typedef struct {
  int32_t kind;
  int32_t cin, cout, k, stride, pad;
  int32_t relu;  /* SPK_ACT_* */
  int32_t src, dst, res;
  int32_t child; /* index of the owning child of `base`; -1 for the head */
  float p;       /* dropout probability */
  char name[96];
  char bn[96];
} spk_layer_desc;

/* Hyper-parameters of one optimizer step; lr per param group as kept by
 * LRWarmup (sykepic/train/network.py:98-130). */
typedef struct {
  int32_t kind;     /* SPK_OPT_* */
  float lr[3];
  float beta1, beta2, eps;  /* Adam */
  float weight_decay;
  float momentum;           /* SGD, RMSprop */
  float grad_scale;         /* multiplies every gradient first (1/world for DP mean) */
  float alpha;              /* RMSprop smoothing constant / Adadelta rho */
  float momentum_decay;     /* NAdam */
  float lr_decay;           /* Adagrad */
  float initial_accumulator_value; /* Adagrad */
} spk_optim_desc;

const char* spk_last_error(void);
const char* spk_version(void);

/* TorchVisionNet(...) construction — sykepic/train/config.py:63-77,
 * sykepic/train/network.py:14-64.  Weights start at zero; load them with
 * spk_model_load_param. */
int spk_model_create(const spk_layer_desc* layers, int n_layers, int in_chans,
                     int num_classes, int device, spk_model** out);
void spk_model_destroy(spk_model* m);
/* net.to(device) has no counterpart (the handle lives on one GPU); the
 * stream is a hipStream_t (0 = default stream). */
int spk_model_set_stream(spk_model* m, void* hip_stream);

/* state_dict()/load_state_dict() — sykepic/train/train.py:300,179,
 * sykepic/compute/probability.py:129.  Enumerates tensors in torch's
 * state_dict order. dtype is SPK_DTYPE_F32 or SPK_DTYPE_I64
 * (num_batches_tracked). Synchronous. */
int spk_model_num_params(spk_model* m);
int spk_model_param_info(spk_model* m, int idx, char* key, int key_cap,
                         int64_t shape[4], int* ndim, int* dtype);
int spk_model_load_param(spk_model* m, const char* key, const void* host, int64_t numel);
int spk_model_read_param(spk_model* m, const char* key, void* host, int64_t numel);

/* param.requires_grad = flag — sykepic/train/network.py:133-172 */
int spk_model_set_requires_grad(spk_model* m, const char* key, int flag);
/* optimizer.param_groups[group]["params"] membership (group -1: not in the
 * optimizer) — sykepic/train/train.py:131-138, network.py:108-128 */
int spk_model_set_param_group(spk_model* m, const char* key, int group);
/* 16-bit storage/MFMA input type of the eval path: 0 = fp16 (default; the
 * 1e-3 probability tolerance needs its 11-bit mantissa), 1 = bf16. fp32
 * accumulation either way. Training always runs bf16. */
int spk_model_set_infer_dtype(spk_model* m, int bf16);
/* Precision knobs of the fp16 eval path (defaults: split_weights = 3,
 * precise_residual = 0).  split_weights = 2 splits only the convs that write
 * the residual trunk (stem, block-closing convs, downsample branches);
 * split_weights = 3 splits every conv except the 3x3 convs inside a residual
 * block (the lo-products that buy the least accuracy per MFMA cycle) - and, on
 * the EfficientNet graphs, no conv at all: their error is the fp16 rounding of
 * the stored activations, the weight rounding does not show beside it
 * (tests/archive/diagnostics/effnet_calibrated.py; 1 still splits every conv).
 * split_weights = 1: every conv weight is carried as
 * hi + lo fp16 halves and both products are accumulated (2x MFMA work, weight
 * rounding error ~2^-22) — weight rounding is the dominant logit error at
 * 16-bit storage.  precise_residual: shortcut tensors keep their fp16
 * rounding remainder for the residual add (+2 B/element of shortcut traffic). */
int spk_model_set_precision(spk_model* m, int split_weights, int precise_residual);
/* Per-op choice of split weights: flags[i] != 0 carries conv op i (index into the
 * spk_layer_desc array of spk_model_create) as hi+lo halves; flags of non-conv ops
 * are ignored.  Replaces the split_weights mode until spk_model_set_precision is
 * called again.  `tests/archive/diagnostics/split_search.py` derives the cheapest mask that keeps
 * the reference's 1e-3 probability tolerance. */
int spk_model_set_split_ops(spk_model* m, const unsigned char* flags, int n_ops);
/* Calibrated single-pass mode (round 4): split_weights = 5 in spk_model_set_precision.
 * An fp16 weight image loses dw = fp16(w) - w per weight and an output loses sum_k dw_k x_k; over the data that error has
 * a mean, sum_k dw_k E[x_k], the same for every pixel of an output channel - and this systematic part is most of the
 * logit error of a plain fp16 forward.  With per-channel means E[x_k] of every conv's input, each weight row (per filter
 * tap) is rounded to nearest and then the weights nearest a rounding midpoint are re-rounded until sum_k E[x_k] dw_k ~ 0
 * ("zero-sum rounding", csrc/zero_sum.hip): the mean error is gone at no run-time cost, the row's squared error grows
 * by < 1 %.  Mode 5 runs every conv as ONE fp16 product (the 7x7 stem too: its 147-weight rows are balanced as a whole
 * against the image's own channel means) and needs the means:
 *   spk_model_calibrate_act_means - one batch of representative images (as spk_forward_infer takes them) through the
 *     most accurate mode; per-channel means of every conv input, accumulated over calls (reset != 0 starts over);
 *   spk_model_get_act_means / spk_model_set_act_means - the flat vector (spk_model_act_means_size floats: cin values
 *     per conv, graph order) to store with the model and restore (`act_means.pth` next to
 *     `best_state.pth`); set with host == NULL forgets them;
 *   spk_model_set_zero_sum - apply the rounding to the un-split convs of ANY split mode (diagnostics).
 * A model's probabilities stay a function of (weights, means, image): nothing depends on the batch an image arrives in.
 * spk_forward_infer in mode 5 without means fails with SPK_ERR_STATE.  Reference call site: `net(x)`,
 * sykepic/compute/probability.py:189. */
int64_t spk_model_act_means_size(spk_model* m);
int spk_model_calibrate_act_means(spk_model* m, const void* x_dev, int n, int h, int w, int layout, int dtype, int reset);
int spk_model_get_act_means(spk_model* m, float* host, int64_t numel);
int spk_model_set_act_means(spk_model* m, const float* host, int64_t numel);
int spk_model_set_zero_sum(spk_model* m, int on);
/* fp8 (OCP e4m3) eval mode of the EfficientNet MBConv blocks — BASELINE config 5.  Inside a block (expand 1x1
 * conv -> depthwise conv -> squeeze-excitation -> project 1x1 conv) the expanded tensors are stored as e4m3
 * bytes with one scale per tensor, the 1x1 convs run on the fp8 MFMA with e4m3 weights (one scale per output
 * channel) and the squeeze-excitation gate is applied in the project conv's operand loader; the residual
 * trunk, the stem, the head and any block without an expand conv stay fp16.  spk_model_calibrate_fp8 runs one
 * fp16 forward of a representative device batch (as spk_forward_infer takes it) and records the activation
 * ranges; it must be called before the first fp8 forward and again after loading other weights.  3 mantissa
 * bits do not reach the 1e-3 probability tolerance of the reference: the fp16 path stays the parity mode. */
int spk_model_set_fp8(spk_model* m, int on);
int spk_model_calibrate_fp8(spk_model* m, const void* x_dev, int n, int h, int w, int layout, int dtype);
/* Which MBConv blocks the fp8 mode covers (round 3).  Blocks are counted in graph order over those that qualify
 * (expand conv -> depthwise -> squeeze-excitation -> project conv); flags[i] != 0 puts block i on the e4m3 path, 0 keeps
 * it fp16.  n_blocks = 0 (or never calling this) selects the default set: every qualifying block that has a
 * shortcut (a block without one replaces the trunk by its e4m3-computed output and costs most of the accuracy).  Takes
 * effect at the next spk_model_calibrate_fp8.  spk_model_num_fp8_blocks returns how many blocks qualify. */
int spk_model_set_fp8_blocks(spk_model* m, const unsigned char* flags, int n_blocks);
int spk_model_num_fp8_blocks(spk_model* m);
/* BatchNorm2d(eps, momentum) of every BatchNorm in the graph: torch's defaults 1e-5 / 0.1 unless the model family says
 * otherwise (torchvision's efficientnet_b5..b7 are built with eps 1e-3, momentum 0.01). */
int spk_model_set_bn(spk_model* m, float eps, float momentum);
/* Dropout mask seed for training steps. */
int spk_model_set_seed(spk_model* m, uint64_t seed);

/* net.eval(); out = net(x); softmax(out * ln(base)) — the body of net_pass,
 * sykepic/compute/probability.py:184-194.  x: n images h x w of the model's
 * in_chans; out: float32 [n, num_classes] device buffer.  softmax_base <= 0
 * returns raw logits (test_net / val loop, sykepic/train/train.py:265,338). */
int spk_forward_infer(spk_model* m, const void* x_dev, int n, int h, int w, int layout,
                      int dtype, float softmax_base, float* out_dev);

/* net.eval() forward + CrossEntropyLoss + arg-max accuracy — the validation
 * loop body, sykepic/train/train.py:262-270.  stats_dev: float32[2] device
 * buffer, ACCUMULATED: stats[0] += loss*n, stats[1] += #correct.  logits_dev
 * may be NULL. */
int spk_eval_step(spk_model* m, const void* x_dev, int n, int h, int w, int layout, int dtype,
                  const int64_t* y_dev, float* stats_dev, float* logits_dev);

/* net.train(); optimizer.zero_grad(); out = net(x); loss = CE(out, y);
 * loss.backward() — sykepic/train/train.py:233,239-242.  Train-mode BatchNorm
 * (batch statistics, running-stat update, num_batches_tracked += 1).
 * Gradients land in the flat buffer of spk_model_grad_buffer; stats_dev as
 * in spk_eval_step (train.py:244-247). */
int spk_train_forward_backward(spk_model* m, const void* x_dev, int n, int h, int w, int layout,
                               int dtype, const int64_t* y_dev, float* stats_dev,
                               float* logits_dev);
/* optimizer.step() — sykepic/train/train.py:243 */
int spk_optim_step(spk_model* m, const spk_optim_desc* opt);

/* Flat float32 gradient buffer covering every trainable tensor (device
 * pointer, element count): what a data-parallel caller all-reduces over RCCL
 * between spk_train_forward_backward and spk_optim_step. */
int spk_model_grad_buffer(spk_model* m, void** dev_ptr, int64_t* numel);
/* Data-parallel overlap: the flat gradient buffer is produced back to front (head and layer4 first, the stem
 * last).  With a callback installed, spk_train_forward_backward reports each contiguous slice [offset, offset +
 * numel) of the buffer as soon as every kernel that writes it has been ENQUEUED: it records an event on the
 * model's stream, makes `comm_stream` (a hipStream_t) wait for it, then calls cb(user, bucket, offset, numel)
 * on the calling host thread.  The callee starts its collective on `comm_stream` (torch: `dist.all_reduce(view,
 * async_op=True)` under `torch.cuda.stream(comm)`), which then runs beside the rest of the backward pass.
 * n_buckets 1..3 (slices: head + last stage | the stage before | everything earlier); cb == NULL removes it. */
typedef void (*spk_grad_ready_fn)(void* user, int bucket, int64_t offset, int64_t numel);
int spk_model_set_grad_ready_callback(spk_model* m, spk_grad_ready_fn cb, void* user, void* comm_stream,
                                      int n_buckets);
/* Gradient of one tensor, copied to host in state_dict layout (tests). */
int spk_model_read_grad(spk_model* m, const char* key, void* host, int64_t numel);

/* Test hook: copy activation `tensor_id` (spk_layer_desc.dst numbering) of the
 * most recent forward to the host as float32 NCHW ([n,c,h,w]; [n,c] for the
 * head).  The stem input (id 0) is not readable.  Synchronous. */
int spk_model_read_activation(spk_model* m, int tensor_id, int n, float* host, int64_t numel);

/* Test hook: gradient w.r.t. activation `tensor_id` left by the most recent
 * spk_train_forward_backward, float32 NCHW like spk_model_read_activation. */
int spk_model_read_activation_grad(spk_model* m, int tensor_id, int n, float* host, int64_t numel);

/* --- Test hooks: single operators of the training step on caller-provided device buffers ---
 * What `out = net(x)` (train mode) and `loss.backward()` run for ONE Conv2d+BatchNorm2d layer
 * (sykepic/train/train.py:240,242): the same launches spk_train_forward_backward makes, so that a parity
 * test can hand a kernel known operands and compare with torch autograd on the same bf16-rounded operands.
 * Activations / gradients: NHWC bf16 device buffers (the 7x7/2 stem input: 4 stored channels, even width);
 * weights and weight gradients: float32 [Cout][kh][kw][Cin] device buffers; h, w are the conv INPUT size.
 * Channel counts: multiples of 64 (stem: cin <= 4, cout 64).  Synchronous (the call returns after the work).
 *
 * conv -> batch statistics -> normalise (+res) (+ReLU).  raw: conv output before BatchNorm (bf16);
 * mask: one ReLU bit per element ([M][C/8] bytes; may be NULL when relu == 0); mean_invstd: float[2][C];
 * running_mean / running_var are updated in place (momentum 0.1, unbiased variance). */
int spk_op_conv_bn_train_forward(const void* x_dev, const float* w_dev, const float* gamma_dev, const float* beta_dev,
                                 float* running_mean_dev, float* running_var_dev, const void* res_dev, void* out_dev,
                                 void* raw_dev, unsigned char* mask_dev, float* mean_invstd_dev, int n, int h, int w,
                                 int cin, int cout, int k, int stride, int pad, int relu, void* hip_stream);
/* BatchNorm2d (+ReLU) backward: g = gradient w.r.t. the layer output [M][C]; dy = gradient w.r.t. the conv output;
 * g_res (optional) receives (or, res_accumulate != 0, adds) the gradient of the shortcut operand;
 * dgamma / dbeta: float[C] (either may be NULL). */
int spk_op_bn_backward(const void* g_dev, const unsigned char* mask_dev, const void* raw_dev, const float* mean_dev,
                       const float* invstd_dev, const float* gamma_dev, float* dgamma_dev, float* dbeta_dev,
                       void* dy_dev, void* g_res_dev, int res_accumulate, int m_rows, int channels, int relu,
                       void* hip_stream);
/* Conv2d data gradient: dx [n,h,w,cin] (= or, accumulate != 0, +=) conv_transpose(dy [n,ho,wo,cout], w). */
int spk_op_conv_dgrad(const void* dy_dev, const float* w_dev, void* dx_dev, int accumulate, int n, int h, int w,
                      int cin, int cout, int k, int stride, int pad, void* hip_stream);
/* Data gradient of a STRIDE-1 conv + the BatchNorm backward of the layer that produced the conv's input, as a training step
 * runs them since round 4: the dgrad epilogue makes that BatchNorm's per-channel sums (sum dz, sum dz * xhat) from the fp32
 * gradient values it is about to store, so the producer's reduce pass is skipped (csrc/conv_igemm.hip, spk_set_bnb).
 * dx (+)= conv_transpose(dy, w) [n,h,w,cin] bf16; raw / mask / mean / invstd / gamma describe the producer ([n,h,w,cin] bf16
 * raw output, one ReLU bit per element, [cin] floats); dy_prod = gradient of the producer's raw output, dgamma / dbeta [cin].
 * res_src / res_bits (both or neither, instead of accumulate): the shortcut gradient picked up at its source - dx =
 * conv_transpose(dy, w) + res_src * bit, res_src [n,h,w,cin] bf16 = the output gradient of the block-closing conv whose
 * shortcut this tensor is, res_bits its ReLU bits - so that its BatchNorm backward need not write it into dx first.
 * Reference: loss.backward(), sykepic/train/train.py:242. */
int spk_op_conv_dgrad_bn_backward(const void* dy, const float* w_ohwi, void* dx, int accumulate, const void* raw,
                                  const unsigned char* mask, const float* mean, const float* invstd, const float* gamma,
                                  float* dgamma, float* dbeta, void* dy_prod, int n, int h, int w, int cin, int cout, int k,
                                  int pad, int relu, const void* res_src, const unsigned char* res_bits, void* stream);
/* Conv2d weight gradient: dw [Cout][kh][kw][Cin] float32 from x [n,h,w,cin] and dy [n,ho,wo,cout]. */
int spk_op_conv_wgrad(const void* x_dev, const void* dy_dev, float* dw_dev, int n, int h, int w, int cin, int cout,
                      int k, int stride, int pad, void* hip_stream);
/* fp8 pointwise conv (the 1x1 convs of the fp8 EfficientNet mode): y = act((e4m3(A) . e4m3(W)^T) * factor + bias)
 * (+ res).  x: [m_rows][cin] fp16 (a_fp8 = 0: converted as x / a_scale) or e4m3 bytes (a_fp8 = 1: value = byte *
 * a_scale; gate, optional: fp32 [m_rows / hw][cin] multiplied in and re-rounded); w: float32 [cout][cin];
 * y: [m_rows][cout] fp16 or e4m3 bytes (out_fp8: byte = e4m3(value / y_scale)); bn_scale / bn_bias: float[cout]. */
int spk_op_pw_fp8(const void* x_dev, int a_fp8, const float* w_dev, void* y_dev, int out_fp8, const void* res_dev,
                  const float* bn_scale_dev, const float* bn_bias_dev, const float* gate_dev, int hw, int m_rows, int cin,
                  int cout, int act, float a_scale, float y_scale, void* hip_stream);
/* Eval-path 1x1 convolution (Conv2d(k=1, bias=False) + eval BatchNorm2d (+ shortcut) (+ ReLU); the 36 pointwise convs
 * `net(x)` runs in a ResNet-50 forward, sykepic/compute/probability.py:189).  x [n,h,w,cin] fp16 NHWC, w float32
 * [cout][cin], bn_scale / bn_bias float[cout] (folded statistics), res (optional) and y [n,ho,wo,cout] fp16;
 * stride 1 or 2; relu 0 / 1; split != 0: weights as fp16 hi + lo.  cfg >= 0: that tile configuration of the
 * direct-operand kernel (0 .. spk_op_conv1x1_num_configs() - 1; SPK_ERR_UNSUPPORTED when it does not fit the
 * problem), cfg < 0: the implicit-GEMM kernel.  Channel counts: multiples of 64.  Synchronous. */
int spk_op_conv1x1(const void* x_dev, const float* w_dev, const float* bn_scale_dev, const float* bn_bias_dev,
                   const void* res_dev, void* y_dev, int n, int h, int w, int cin, int cout, int stride, int relu,
                   int split, int cfg, void* hip_stream);
int spk_op_conv1x1_num_configs(void);
/* Eval-path 3x3 stride-1 pad-1 convolution (Conv2d(k=3, padding=1, bias=False) + eval BatchNorm2d (+ ReLU): the
 * middle conv of the bottleneck blocks `net(x)` runs, sykepic/compute/probability.py:189).  x [n,h,w,cin] fp16 NHWC,
 * w float32 [cout][3][3][cin], y [n,h,w,cout] fp16.  cfg >= 0: that tile configuration of the LDS-window kernel
 * (conv_c3.hip; SPK_ERR_UNSUPPORTED when it does not fit), cfg < 0: the implicit-GEMM kernel.  res_dev (may be null):
 * shortcut [n,h,w,cout] fp16 added before the activation (the block-closing conv of a ResNet-18/34 basic block).
 * Synchronous. */
int spk_op_conv3x3(const void* x_dev, const float* w_dev, const float* bn_scale_dev, const float* bn_bias_dev,
                   const void* res_dev, void* y_dev, int n, int h, int w, int cin, int cout, int relu, int split, int cfg,
                   void* hip_stream);
int spk_op_conv3x3_num_configs(void);
/* A whole identity bottleneck block of the eval path (torchvision Bottleneck.forward without downsample, reached through
 * `net(x)`: /root/reference/sykepic/compute/probability.py:189): y = ReLU(BN3(W3 . ReLU(BN2(W2 * ReLU(BN1(W1 . x))))) + x).
 * x, y: [n,h,w,4 cm] fp16 device tensors (y != x); fp32 device weights w1 [cm][4 cm], w2 [cm][3][3][cm], w3 [4 cm][cm];
 * folded eval-BatchNorm scale / shift vectors.  fused != 0: one kernel (csrc/conv_bneck.hip), SPK_ERR_UNSUPPORTED when
 * the shape has no instantiation; fused == 0: the three launches of the eval path.  iters > 0: *ms_out = mean time of one
 * block over `iters` repetitions. */
int spk_op_bottleneck(const void* x_dev, const float* w1_dev, const float* w2_dev, const float* w3_dev, const float* s1_dev,
                      const float* b1_dev, const float* s2_dev, const float* b2_dev, const float* s3_dev, const float* b3_dev,
                      void* y_dev, int n, int h, int w, int cm, int fused, int iters, float* ms_out, void* stream,
                      unsigned long long* stamps_dev /* diagnostics: [blocks][8] shader-clock stamps of the fused kernel, or null */,
                      /* optional: the 1x1 conv + BatchNorm + ReLU that reads the block's output (the next block's conv1):
                       * wz [coutz][4 cm] fp32, z [n,h,w,coutz] fp16.  fused == 2: conv1 on its own, then conv2 + conv3 + shortcut
                       * (+ that conv) as one kernel (conv_btail_kernel) */
                      const float* wz_dev, const float* sz_dev, const float* bz_dev, void* z_dev, int coutz);
/* Two chained 1x1 convs in ONE launch (round 4, csrc/conv_pw.hip): y = act(BN(W . x) + res) with cout = 256, then
 * z = actz(BNz(Wz . y)) computed from the output tile while it is still in registers - what the eval path runs for a
 * bottleneck's block-closing conv and the next block's first conv in the single-weight-image modes (the trunk y is
 * written for later shortcut adds, but never re-read by the conv that follows).  Tensors NHWC fp16 on the device,
 * weights fp32 [cout][cin] / [coutz][cout].  y and z are bit-identical to two spk_op_conv1x1(split = 0) calls. */
int spk_op_conv1x1_chain(const void* x, const float* w, const float* bn_scale, const float* bn_bias, const void* res,
                         void* y, const float* wz, const float* bnz_scale, const float* bnz_bias, void* z, int n, int h,
                         int wd, int cin, int cout, int coutz, int relu, int reluz, void* stream);
/* A block-closing 1x1 conv and the block's 1x1 shortcut (downsample) conv as ONE K-concatenated GEMM (round 3,
 * csrc/conv_pw.hip / conv_pwr.hip, PwConvArgs::x2): y = act(BN1(W1 . x) + BN2(W2 . x2[stride2])) - what the eval path
 * runs in the first block of every ResNet stage (torchvision Bottleneck.forward with `downsample`, reached through
 * `net(x)`: sykepic/compute/probability.py:189).  x [n,ho,wo,cin], x2 [n,h2,w2,cin2], y [n,ho,wo,cout] NHWC fp16 on the
 * device; weights fp32 [cout][cin] / [cout][cin2]; s / b: the folded eval-BatchNorm scale and shift of each conv.
 * cfg: a configuration of the 1x1 kernels, 0 .. spk_op_conv1x1_num_configs() - 1 (SPK_ERR_UNSUPPORTED when it does not
 * fit the problem); every configuration gives the same bits. */
int spk_op_conv1x1_dual(const void* x, const float* w1, const float* s1, const float* b1, const void* x2, const float* w2,
                        const float* s2, const float* b2, void* y, int n, int ho, int wo, int cin, int h2, int w2d, int cin2,
                        int cout, int stride2, int relu, int split, int cfg, void* stream);
/* Zero-sum rounding (csrc/zero_sum.hip) of fp32 weight rows on caller-provided DEVICE buffers: w, out [rows][row_len],
 * mu [mu_period] (element k weighted with mu[k % mu_period]) or NULL (all ones).  out[i] is fp16(w[i]) or the fp16
 * neighbour on the other side of w[i]; per row sum_k mu_k (out_k - w_k) is driven to ~0.  Synchronises the stream. */
int spk_op_zero_sum_round(const float* w_dev, const float* mu_dev, float* out_dev, int64_t rows, int row_len,
                          int mu_period, void* stream);
/* Depthwise Conv2d(C, C, k, stride, pad (k-1)/2, groups=C) + folded BatchNorm + activation (EfficientNet MBConv):
 * x [n,h,w,C] fp16 NHWC, w float32 [C][k*k], y [n,ho,wo,C] fp16; pool (optional) float32 [n][C] = per-image sums of
 * the outputs (squeeze-excitation numerator).  lds != 0: the LDS row-ring kernel, else the gather kernel. */
int spk_op_dwconv(const void* x_dev, const float* w_dev, const float* bn_scale_dev, const float* bn_bias_dev, void* y_dev,
                  float* pool_dev, int n, int h, int w, int channels, int k, int stride, int act, int lds, void* hip_stream);

/* --- ROI preprocessing straight from the .roi blob ---
 * One ROI of an IFCB sample: byte offset into the .roi blob, width, height
 * (columns 17/15/16 of the .adc line, sykepic/utils/ifcb.py:100-110). */
typedef struct {
  int64_t offset;
  int32_t width, height;
} spk_roi;
/* blob -> [n, out_h, out_w, 3] uint8 (mode/black/white border, aspect-preserving
 * OpenCV-style fixed-point bilinear resize), replacing the PNG round trip +
 * ImageDataset + Compose of the reference (sykepic/utils/ifcb.py:76-118,
 * sykepic/train/data.py:210-231, sykepic/train/image.py:25-56).  border: -1 =
 * the ROI's modal grey level, else a grey value 0..255.  The result feeds
 * spk_forward_infer(layout NHWC, dtype U8).  All pointers are device memory. */
int spk_preprocess_rois(const unsigned char* blob_dev, int64_t blob_bytes, const spk_roi* rois_dev,
                        int n, int out_h, int out_w, int border, unsigned char* out_dev, void* hip_stream);

/* Training augmentations on a batch of resized+bordered uint8 NHWC images:
 * replaces the cv2 calls of sykepic/train/image.py:80-180 that Compose.__call__ (image.py:25-56) makes per
 * image in the DataLoader workers.  The random draws stay on the host (Python `random`, reference call order);
 * ops_dev is [n_ops][n]: op j of every sample, applied in order, each rounding to uint8 as the host does.
 *   FLIP_H / FLIP_V: i0 = apply flag            TRANSLATE: i0 = x shift, i1 = y shift (constant border)
 *   ZOOM: d[0] = zoom factor f, i0 = side of the resized square (cvRound(w * f)); cv2.resize(fx = fy = f), then
 *         centred pad (f < 1) or crop
 *   ROTATE: d[0..5] = the 2x3 matrix as cv::warpAffine inverts it (row major); OpenCV's fixed-point bilinear
 *           warp (1/32-pixel coordinates, 15-bit weight table), constant border
 *   BRIGHT: d[0] = factor, truncating
 * border_dev: [n][4] bytes (one value per channel).  tmp_dev: scratch of the batch size (needed when n_ops > 1).
 * The result is written to out_dev. */
#define SPK_AUG_FLIP_H 1
#define SPK_AUG_FLIP_V 2
#define SPK_AUG_TRANSLATE 3
#define SPK_AUG_ZOOM 4
#define SPK_AUG_ROTATE 5
#define SPK_AUG_BRIGHT 6
typedef struct {
  int32_t kind;
  int32_t i0, i1;
  int32_t pad_;
  double d[6];
} spk_aug_op;
int spk_augment_batch(const unsigned char* in_dev, unsigned char* out_dev, unsigned char* tmp_dev, int n, int h,
                      int w, int c, const spk_aug_op* ops_dev, int n_ops, const unsigned char* border_dev,
                      void* stream);

/* --- prediction from probabilities and thresholds ---
 * row_prediction of sykepic/compute/prediction.py:49-71 for n rows at once:
 * thresholds_dev = float[num_classes] (+inf for a class without a threshold):
 * the highest-probability class with p >= its own threshold, classified = 1;
 * if none, arg-max with classified = 0.  thresholds_dev == NULL: arg-max and
 * classified = p > scalar_threshold.  pred_dev int32[n], classified_dev u8[n]. */
int spk_predict_rows(const float* probs_dev, int n, int num_classes, const float* thresholds_dev,
                     float scalar_threshold, int32_t* pred_dev, unsigned char* classified_dev,
                     void* hip_stream);

/* Per-layer timing of the last spk_forward_infer call (HIP events on the
 * model's stream); used by bench.py for the roofline line. Returns the
 * number of records written. */
typedef struct {
  char name[96];
  float ms;
  double flops;
  double bytes;
} spk_layer_time;
int spk_model_profile_infer(spk_model* m, const void* x_dev, int n, int h, int w, int layout,
                            int dtype, int iters, spk_layer_time* out, int cap);

/* Per-phase timing of spk_train_forward_backward (same event method): one
 * record per kernel group (conv fwd / dgrad / wgrad, BN, pooling, head...);
 * `flops` = algorithmic FLOPs of the phase, `bytes` = launches per step. */
int spk_model_profile_train(spk_model* m, const void* x_dev, int n, int h, int w, int layout,
                            int dtype, const int64_t* y_dev, float* stats_dev, int iters,
                            spk_layer_time* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
