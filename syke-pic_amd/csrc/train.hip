// Training step executor: the library side of
//   net.train(); optimizer.zero_grad(); out = net(x); loss = CE(out, y);
//   loss.backward(); optimizer.step()         (sykepic/train/train.py:233,239-243)
// bf16 activations / gradients, fp32 master weights, statistics and optimizer
// state.  Train-mode BatchNorm everywhere (train.py:233 overrides the .eval()
// of frozen modules); data gradients flow through every layer down to the
// stem's BatchNorm, weight gradients only where requires_grad is set
// (sykepic/train/network.py:133-172).
#include "model.h"
#include "train_effnet.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

static int tfail(int code, const std::string& msg) {
  spk_set_error(msg);
  return code;
}
#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) return tfail(SPK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define SPK_TRY(expr)            \
  do {                           \
    int r_ = (expr);             \
    if (r_ != SPK_OK) return r_; \
  } while (0)
#define K_TRY(expr, what)                                            \
  do {                                                               \
    if ((expr) != 0) return tfail(SPK_ERR_HIP, std::string(what) + " launch failed"); \
  } while (0)

struct ConvTrain {
  size_t raw_off = 0;       // bf16 raw conv output (pre-BN), same shape as dst
  size_t mask_off = 0;      // ReLU mask, 1 bit per element
  size_t stat_off = 0;      // floats: mean[C], invstd[C], scale[C], shift[C]
  size_t wfwd_off = 0;      // bf16 [Cout][K] forward image
  size_t wdg_off = 0;       // bf16 [Cin][taps][Cout] dgrad image
  size_t dwt_off = 0;       // depthwise conv: fp32 tap-major weights [taps][C] (floats into TrainState::dwt)
  size_t se_off = 0;        // squeeze-excitation: pooled [n][C], u1 [n][S], h1 [n][S], gate [n][C] floats kept for backward,
                            // then du2 [n][C], du1 [n][S]: per layer, the side stream's weight-gradient kernel reads them
  size_t rs_off = 0;        // stochastic depth: per-image factor [n] floats (0: none)
  size_t dy_off = 0;        // this layer's own pre-BatchNorm gradient tensor dy (side-stream weight gradients read it)
};

struct PhaseProf {
  bool on = false;
  std::vector<hipEvent_t> pool;
  std::vector<std::pair<int, int>> marks;  // (phase id, event index)
  int used = 0;
};
enum { PH_INPUT, PH_CONV_FWD, PH_BN_FWD, PH_POOL_FWD, PH_HEAD_FWD, PH_LOSS, PH_HEAD_BWD, PH_POOL_BWD,
       PH_BN_BWD, PH_CONV_DGRAD, PH_CONV_WGRAD, PH_WGRAD_REDUCE, PH_COUNT };
static const char* kPhaseName[PH_COUNT] = {
    "input.to_nhwc4", "conv_fwd (conv_igemm_kernel)", "bn_fwd (finalize+apply)", "pool_fwd", "head_fwd",
    "cross_entropy", "head_bwd", "pool_bwd", "bn_bwd (reduce+finalize+apply)",
    "conv_dgrad (conv_igemm_kernel)", "conv_wgrad (conv_wgrad_kernel)", "wgrad_slab_reduce"};

struct TrainState {
  PhaseProf prof;
  // persistent across plans
  float* gbuf = nullptr;   // flat gradients   [n_train]
  float* m1 = nullptr;     // Adam exp_avg / SGD momentum [n_train]
  float* m2 = nullptr;     // Adam exp_avg_sq [n_train]
  bf16_t* wpack = nullptr; // bf16 forward + dgrad weight images
  float* stats = nullptr;  // per-conv mean/invstd/scale/shift
  float* dwt = nullptr;    // depthwise weights, tap-major (EfficientNet)
  float* unit = nullptr;   // [unit_c] ones, [unit_c] zeros
  size_t unit_c = 0;
  std::vector<ConvTrain> conv;  // indexed by layer
  bool weights_dirty = true;
  bool sgd_started = false;
  unsigned long long steps = 0;   // training steps so far (dropout masks differ from step to step)

  // per (n,h,w) plan
  void* arena = nullptr;
  int cap_n = 0, cap_h = 0, cap_w = 0;
  std::vector<size_t> goff;   // gradient tensor per activation id
  size_t dy_off = 0, idx_off = 0, part_off = 0, coef_off = 0, slab_off = 0, tmp_off = 0;
  size_t fpart_off = 0;    // BatchNorm-backward partial sums written by the dgrad epilogue of the consumer layer
  size_t se_tmp_off = 0;   // squeeze-excitation scratch shared by the layers (pool partials, gate / hidden gradients)
  // Weight gradients on a second stream (ResNets): wgrad(i) needs only the layer's input activation and dy(i), so it runs
  // beside dgrad(i) and the HBM-bound BatchNorm backward of the next layer instead of in front of them.  Every conv layer
  // has its own dy tensor (ConvTrain::dy_off, round 4); with SPK_DY_PER_LAYER=0 dy is double buffered as before: bn_bwd of
  // layer i-2 may then overwrite a buffer only after the wgrad that read it has finished (ev_dy_free).
  size_t dy2_off = 0;
  // The step's LAST weight gradient (the first layer's: its dy is the last tensor the backward makes) runs on the side
  // stream while the main stream has nothing left to do.  Unless somebody reads the gradients in between, the step
  // returns WITHOUT waiting for it: spk_optim_step updates every other parameter first and joins the side stream in
  // front of that one tensor (ResNet-50, batch 256: 0.36 ms of idle main queue in front of the optimizer otherwise).
  // Every other consumer of side-stream results goes through spk_train_join.
  bool tail_pending = false;     // ev_side_done is recorded, the main stream has not waited for it
  int tail_param = -1;           // the parameter whose gradient is still being made
  bool grads_exported = false;   // spk_model_grad_buffer handed the flat buffer out: always join at the end of the step
  hipEvent_t ev_side_pre = nullptr;   // side stream, in front of the tail weight gradient
  hipStream_t side = nullptr;
  hipEvent_t ev_dy_ready[2] = {nullptr, nullptr};   // main: dy buffer written
  hipEvent_t ev_dy_free[2] = {nullptr, nullptr};    // side: wgrad has read the dy buffer
  hipEvent_t ev_side_done = nullptr;
  hipEvent_t ev_se = nullptr;                       // main: gate gradients of a squeeze-excitation layer written
  bool dy_busy[2] = {false, false};
  int dy_slot = 0;
  size_t part_floats = 0, slab_floats = 0;

  void* G(int t) const { return (char*)arena + goff[t]; }
  bf16_t* RAW(int layer) const { return (bf16_t*)((char*)arena + conv[layer].raw_off); }
  unsigned char* MASK(int layer) const { return (unsigned char*)arena + conv[layer].mask_off; }
};

void spk_train_free(spk_model* m) {
  TrainState* t = m->train;
  if (!t) return;
  if (t->arena) hipFree(t->arena);
  if (t->gbuf) hipFree(t->gbuf);
  if (t->m1) hipFree(t->m1);
  if (t->m2) hipFree(t->m2);
  if (t->wpack) hipFree(t->wpack);
  if (t->stats) hipFree(t->stats);
  if (t->dwt) hipFree(t->dwt);
  if (t->unit) hipFree(t->unit);
  if (t->side) hipStreamDestroy(t->side);
  for (int i = 0; i < 2; ++i) {
    if (t->ev_dy_ready[i]) hipEventDestroy(t->ev_dy_ready[i]);
    if (t->ev_dy_free[i]) hipEventDestroy(t->ev_dy_free[i]);
  }
  if (t->ev_side_done) hipEventDestroy(t->ev_side_done);
  if (t->ev_side_pre) hipEventDestroy(t->ev_side_pre);
  if (t->ev_se) hipEventDestroy(t->ev_se);
  delete t;
  m->train = nullptr;
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// records an event AFTER the work of `phase` was enqueued (profiling runs only)
static void mark(spk_model* m, int phase) {
  PhaseProf& p = m->train->prof;
  if (!p.on) return;
  if (p.used == (int)p.pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    p.pool.push_back(e);
  }
  hipEventRecord(p.pool[p.used], m->stream);
  p.marks.push_back({phase, p.used});
  ++p.used;
}

static int ensure_state(spk_model* m) {
  if (m->train) return SPK_OK;
  TrainState* t = new TrainState();
  m->train = t;
  const size_t nb = std::max<size_t>(m->n_train, 64) * sizeof(float);
  HIP_TRY(hipMalloc((void**)&t->gbuf, nb));
  HIP_TRY(hipMalloc((void**)&t->m1, nb));
  HIP_TRY(hipMalloc((void**)&t->m2, nb));
  HIP_TRY(hipMemset(t->gbuf, 0, nb));
  HIP_TRY(hipMemset(t->m1, 0, nb));
  HIP_TRY(hipMemset(t->m2, 0, nb));
  t->conv.resize(m->layers.size());
  // channel counts as the training plan lays them out: cout_p / cin_p (= the layer's own for the ResNets, whose widths
  // are multiples of 64; rounded up to 64 for the EfficientNets)
  size_t w = 0, st = 0, dw = 0, unit_c = 0;
  for (size_t i = 0; i < m->layers.size(); ++i) {
    const Layer& L = m->layers[i];
    ConvTrain& c = t->conv[i];
    if (L.d.kind == SPK_OP_DWCONV) {
      c.stat_off = st;
      st += (size_t)4 * L.cout_p;
      c.dwt_off = dw;
      dw += (size_t)2 * L.d.k * L.d.k * L.cout_p;   // + the flipped window (pack_padded_multi_kernel)
      unit_c = std::max(unit_c, (size_t)L.cout_p);
      continue;
    }
    if (L.d.kind != SPK_OP_CONV) continue;
    c.wfwd_off = w;
    w += (size_t)L.cout_p * L.kpad;
    c.wdg_off = w;
    if (L.mode == CONV_MODE_GENERIC) w += (size_t)L.cin_p * L.d.k * L.d.k * L.cout_p;
    c.stat_off = st;
    st += (size_t)4 * L.cout_p;
  }
  HIP_TRY(hipMalloc((void**)&t->wpack, std::max<size_t>(w, 8) * 2));
  HIP_TRY(hipMalloc((void**)&t->stats, std::max<size_t>(st, 8) * 4));
  HIP_TRY(hipMalloc((void**)&t->dwt, std::max<size_t>(dw, 8) * 4));
  if (unit_c) {   // [C] ones, [C] zeros: the depthwise kernel shared with the eval path applies a per-channel a*x + b
    std::vector<float> unit(2 * unit_c, 0.f);
    std::fill(unit.begin(), unit.begin() + unit_c, 1.f);
    HIP_TRY(hipMalloc((void**)&t->unit, unit.size() * 4));
    HIP_TRY(hipMemcpy(t->unit, unit.data(), unit.size() * 4, hipMemcpyHostToDevice));
    t->unit_c = unit_c;
  }
  static const bool side_wgrad = !getenv("SPK_WGRAD_STREAM") || atoi(getenv("SPK_WGRAD_STREAM")) != 0;
  if (side_wgrad) {
    {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo: numerically greatest = lowest priority
      // lowest priority: the critical path (BatchNorm backward -> dgrad chain) gets the CUs first, the weight gradients
      // fill what is left (measured 24.00 -> 23.83 ms against the default priority; highest: no change)
      const char* pe = getenv("SPK_WGRAD_PRIO");
      const int prio = pe ? (atoi(pe) > 0 ? hi : (atoi(pe) < 0 ? lo : 0)) : lo;
      HIP_TRY(hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, prio));
    }
    // the events order two streams of ONE device: no system-scope fence (SPK_EVENT_SYSFENCE=1 keeps it, for A/B runs)
    static const bool sysfence = getenv("SPK_EVENT_SYSFENCE") && atoi(getenv("SPK_EVENT_SYSFENCE")) != 0;
    const unsigned evf = hipEventDisableTiming | (sysfence ? 0u : (unsigned)hipEventDisableSystemFence);
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipEventCreateWithFlags(&t->ev_dy_ready[i], evf));
      HIP_TRY(hipEventCreateWithFlags(&t->ev_dy_free[i], evf));
    }
    HIP_TRY(hipEventCreateWithFlags(&t->ev_side_done, evf));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_side_pre, evf));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_se, evf));
  }
  return SPK_OK;
}

// activations live in the model's arena (spk_plan); everything backward needs
// lives in the training arena
static int plan_train(spk_model* m, int n, int h, int w) {
  TrainState* t = m->train;
  SPK_TRY(spk_plan(m, n, h, w, m->effnet));
  if (n <= t->cap_n && h == t->cap_h && w == t->cap_w) return SPK_OK;
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (t->arena) hipFree(t->arena);
  t->arena = nullptr;
  size_t total = 0, max_conv = 0, max_slab = 0, max_part = 0, max_c = 0, max_se = 0;
  t->goff.assign(m->n_tensors, 0);
  for (int id = 0; id < m->n_tensors; ++id) {
    const TDim& d = m->tdims[id];
    const size_t bytes = (size_t)n * d.h * d.w * d.c * (d.bf16 ? 2 : 4);
    if (bytes >= ((size_t)1 << 31))
      return tfail(SPK_ERR_UNSUPPORTED, "training batch too large: an activation exceeds 2 GiB");
    t->goff[id] = total;
    total += al256(bytes);
  }
  for (size_t i = 0; i < m->layers.size(); ++i) {
    const Layer& L = m->layers[i];
    const TDim& in = m->tdims[L.d.src];
    const TDim& o = m->tdims[L.d.dst];
    if (L.d.kind == SPK_OP_CONV || L.d.kind == SPK_OP_DWCONV) {
      const int C = o.c;   // channels as laid out
      const size_t bytes = (size_t)n * o.h * o.w * C * 2;
      t->conv[i].raw_off = total;
      total += al256(bytes);
      t->conv[i].mask_off = total;
      total += al256(bytes / 16);
      max_conv = std::max(max_conv, bytes);
      const int M = n * o.h * o.w;
      if (L.d.kind == SPK_OP_DWCONV) {
        max_slab = std::max(max_slab, (size_t)spk_dw_wgrad_rows(M, C) * L.d.cout * L.d.k * L.d.k);
      } else if (L.mode == CONV_MODE_STEM3) {
        int ppb;
        max_slab = std::max(max_slab, (size_t)spk_stem3_wgrad_blocks(M, &ppb) * L.d.cout * 9 * L.d.cin);
      } else {
        int sp, pps;
        const int ktot = L.mode == CONV_MODE_STEM ? 256 : L.d.k * L.d.k * in.c;
        spk_wgrad_plan(M, C, ktot, &sp, &pps);
        max_slab = std::max(max_slab, (size_t)sp * C * ktot);
      }
      int rpb;
      const int nb = spk_bn_bwd_blocks(M, C, &rpb);
      const size_t fwd_part = (size_t)((M + 60) / 61) * 2 * C;  // sized for M tiles of 61 rows (the smallest tile any forward kernel has used; today 64)
      max_part = std::max(max_part, std::max((size_t)nb * 2 * C, fwd_part));
      max_c = std::max(max_c, (size_t)C);
      if (L.d.kind == SPK_OP_CONV && L.d.res >= 0 && L.d.p > 0.f) {   // stochastic depth on the residual branch
        t->conv[i].rs_off = total;
        total += al256((size_t)n * 4);
      }
    } else if (L.d.kind == SPK_OP_SE) {
      const size_t fl = (size_t)n * (3 * o.c + 3 * L.d.k);
      t->conv[i].se_off = total;
      total += al256(fl * 4);
      // shared scratch: pool partials [n][chunks][C], dpool [n][C], hidden-gradient partials [n][gate tiles][S]
      max_se = std::max(max_se, (size_t)n * ((size_t)(spk_se_chunks(o.h * o.w) + 1) * o.c +
                                             (size_t)spk_se_gate_tiles(L.d.cout, L.d.k) * L.d.k));
    } else if (L.d.kind == SPK_OP_MAXPOOL) {
      t->idx_off = total;
      total += al256((size_t)n * o.h * o.w * o.c);
    } else if (L.d.kind == SPK_OP_DROPOUT) {
      t->conv[i].mask_off = total;   // one byte per element of the head tensor
      total += al256((size_t)n * o.c);
    }
  }
  t->dy_off = total;      total += al256(max_conv);
  t->dy2_off = total;     total += al256(max_conv);
  // One dy tensor per conv layer (SPK_DY_PER_LAYER=0: the two shared buffers of rounds 3-4): the main stream then never waits
  // for the weight-gradient stream to release a buffer - one barrier packet less in front of every BatchNorm backward,
  // ResNet-50 22.97 -> 22.49 ms, EfficientNet-B4 27.93 -> 27.55 - and the weight gradients may lag as far as their own queue
  // allows (5.6 GB at ResNet-50 batch 256 against 288 GB of HBM).  (Releasing the weight gradients of 2-6 layers behind ONE
  // event record instead of one each was measured as well: 22.8-23.1 ms - the later start costs more than the packets.)
  static const bool dy_per_layer = !getenv("SPK_DY_PER_LAYER") || atoi(getenv("SPK_DY_PER_LAYER")) != 0;
  for (size_t i = 0; i < m->layers.size(); ++i) {
    const Layer& L = m->layers[i];
    t->conv[i].dy_off = 0;
    if (!dy_per_layer || (L.d.kind != SPK_OP_CONV && L.d.kind != SPK_OP_DWCONV)) continue;
    const TDim& o = m->tdims[L.d.dst];
    t->conv[i].dy_off = total;
    total += al256((size_t)n * o.h * o.w * o.c * 2);
  }
  t->part_off = total;    total += al256(max_part * 4);
  t->fpart_off = total;   total += al256(max_part * 4);
  t->coef_off = total;    total += al256(max_c * 3 * 4);
  t->tmp_off = total;     total += al256(max_c * 2 * 64 * 4);
  t->slab_off = total;    total += al256(max_slab * 4);
  t->se_tmp_off = total;  total += al256(max_se * 4);
  t->part_floats = max_part;
  t->slab_floats = max_slab;
  HIP_TRY(hipMalloc(&t->arena, total));
  if (max_se) HIP_TRY(hipMemsetAsync((char*)t->arena + t->se_tmp_off, 0, al256(max_se * 4), m->stream));
  for (size_t i = 0; i < m->layers.size(); ++i)   // pad columns of the gate pre-activations stay zero for good
    if (m->layers[i].d.kind == SPK_OP_SE) {
      const TDim& o = m->tdims[m->layers[i].d.dst];
      HIP_TRY(hipMemsetAsync((char*)t->arena + t->conv[i].se_off, 0, (size_t)n * (3 * o.c + 3 * m->layers[i].d.k) * 4,
                             m->stream));
    }
  t->cap_n = n; t->cap_h = h; t->cap_w = w;
  return SPK_OK;
}

static int repack_weights(spk_model* m) {
  TrainState* t = m->train;
  if (!t->weights_dirty) return SPK_OK;
  // every conv's forward and data-gradient image through table-driven launches (64 images each); the stem's
  // forward image has its own layout and kernel
  PackTable tab;
  tab.count = 0;
  auto flush = [&]() -> int {
    const int r = spk_launch_pack_multi(m->pbuf, t->wpack, tab, m->stream);
    tab.count = 0;
    return r;
  };
  PadPackTable ptab;   // EfficientNet: channel-padded GEMM images and depthwise windows, 48 tensors per launch
  ptab.count = 0;
  auto pflush = [&]() -> int {
    const int r = spk_launch_pack_padded_multi(m->pbuf, t->wpack, t->dwt, ptab, m->stream);
    ptab.count = 0;
    return r;
  };
  auto padd = [&](size_t src, size_t dst, int cout, int taps, int cin, int cout_p, int cin_p, int kind) -> int {
    PadPackEntry& e = ptab.e[ptab.count++];
    e.src = src; e.dst = dst;
    e.cout = (unsigned)cout; e.taps = (unsigned)taps; e.cin = (unsigned)cin;
    e.cout_p = (unsigned)cout_p; e.cin_p = (unsigned)cin_p; e.kind = (unsigned)kind;
    return ptab.count == 48 ? pflush() : 0;
  };
  for (size_t i = 0; i < m->layers.size(); ++i) {
    const Layer& L = m->layers[i];
    if (L.d.kind == SPK_OP_DWCONV) {
      K_TRY(padd(m->params[L.p_w].off, t->conv[i].dwt_off, L.d.cout, L.d.k * L.d.k, 0, L.cout_p, 0, 2), "dw_pack");
      continue;
    }
    if (L.d.kind != SPK_OP_CONV || L.mode == CONV_MODE_STEM3) continue;   // the 3x3 stem reads the master weights
    if (m->effnet) {   // channel-padded GEMM images, zeros outside the layer's own widths
      for (int kind = 0; kind < 2; ++kind)
        K_TRY(padd(m->params[L.p_w].off, kind ? t->conv[i].wdg_off : t->conv[i].wfwd_off, L.d.cout, L.d.k * L.d.k, L.d.cin,
                   L.cout_p, L.cin_p, kind), "pack_train_padded");
      continue;
    }
    if (L.mode == CONV_MODE_STEM) {
      K_TRY(spk_launch_pack_weights(m->P(L.p_w), t->wpack + t->conv[i].wfwd_off, L.d.cout, L.d.k, L.d.k, L.d.cin,
                                    L.mode, DT_BF16, 0, m->stream, 1.0f / SPK_INPUT_SCALE), "pack_weights");
      continue;
    }
    for (int kind = 0; kind < 2; ++kind) {
      PackEntry& e = tab.e[tab.count++];
      e.src = m->params[L.p_w].off;
      e.dst = kind ? t->conv[i].wdg_off : t->conv[i].wfwd_off;
      e.cout = (unsigned)L.d.cout; e.taps = (unsigned)(L.d.k * L.d.k); e.cin = (unsigned)L.d.cin;
      e.kind = kind;
      if (tab.count == 64) K_TRY(flush(), "pack_multi");
    }
  }
  K_TRY(flush(), "pack_multi");
  K_TRY(pflush(), "pack_padded_multi");
  t->weights_dirty = false;
  return SPK_OK;
}

void spk_train_mark_dirty(spk_model* m) {
  if (m->train) m->train->weights_dirty = true;
}

static void fill_conv(ConvArgs& a, const bf16_t* x, const bf16_t* w, bf16_t* y, int N, int H, int W,
                      int Cin, int Ho, int Wo, int Cout, int k, int stride, int pad, int K) {
  memset(&a, 0, sizeof a);
  a.cfg = a.dma = -1;
  a.cls_ph = a.cls_pw = -1;
  a.x = x; a.w = w; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.kh = a.kw = k; a.stride = stride; a.pad = pad;
  a.M = N * Ho * Wo;
  a.K = K;
  a.dt = DT_BF16;
  a.x_bytes = (unsigned)((size_t)N * H * W * Cin * 2);
  a.w_bytes = (unsigned)((size_t)Cout * K * 2);
}


// ---------------------------------------------------------------------------
// Backward of one convolution, shared by the training step below and by the single-operator test hooks
// (ops_abi.hip), so that the isolation tests exercise exactly the launches a training step makes.
// ---------------------------------------------------------------------------
// dx (+)= conv_transpose(dy, w): dy [n,oh,ow,cout], wdg = the [Cin][kh][kw][Cout] bf16 image, dx [n,ih,iw,cin]
int spk_conv_dgrad_all(const bf16_t* dy, const bf16_t* wdg, bf16_t* dx, bool accumulate, int n, int oh, int ow,
                       int cout, int ih, int iw, int cin, int k, int stride, int pad, hipStream_t s, BnbFuse* fuse) {
  ConvArgs a;
  fill_conv(a, dy, wdg, dx, n, oh, ow, cout, ih, iw, cin, k, stride, pad, k * k * cout);
  a.res = accumulate ? (const bf16_t*)dx : nullptr;
  if (stride == 1) {
    if (fuse) {
      spk_set_bnb(a, fuse->partials, fuse->raw, fuse->mask, fuse->mean, fuse->invstd, fuse->res_src ? fuse->res_bits : nullptr);
      if (fuse->res_src) a.res = fuse->res_src;
    }
    K_TRY(spk_conv_launch(a, CONV_MODE_DGRAD, s, fuse ? &fuse->tiles : nullptr), "conv dgrad");
    return SPK_OK;
  }
  if (fuse) return tfail(SPK_ERR_ARG, "the fused BatchNorm reduction needs a single-launch (stride 1) data gradient");
  if (stride != 2) return tfail(SPK_ERR_UNSUPPORTED, "conv stride must be 1 or 2 on the training path");
  // one launch per output parity class; a class that no tap can reach (1x1 stride 2: three of four) is all zeros
  bool need_zero = false;
  for (int cl = 0; cl < 4; ++cl) {
    const int ph = cl >> 1, pw = cl & 1;
    const int r0 = (ph + pad) & 1, s0 = (pw + pad) & 1;
    const int nr = r0 < k ? (k - r0 + 1) / 2 : 0, ns = s0 < k ? (k - s0 + 1) / 2 : 0;
    if (nr * ns == 0 && (ih - ph + 1) / 2 > 0 && (iw - pw + 1) / 2 > 0) need_zero = true;
  }
  if (need_zero && !accumulate) HIP_TRY(hipMemsetAsync(dx, 0, (size_t)n * ih * iw * cin * 2, s));
  for (int cl = 0; cl < 4; ++cl) {
    const int ph = cl >> 1, pw = cl & 1;
    const int r0 = (ph + pad) & 1, s0 = (pw + pad) & 1;
    const int nr = r0 < k ? (k - r0 + 1) / 2 : 0, ns = s0 < k ? (k - s0 + 1) / 2 : 0;
    const int h2 = (ih - ph + 1) / 2, w2 = (iw - pw + 1) / 2;
    if (nr * ns == 0 || h2 <= 0 || w2 <= 0) continue;
    ConvArgs c = a;
    c.cls_ph = ph; c.cls_pw = pw;
    c.oH = ih; c.oW = iw;
    c.Ho = h2; c.Wo = w2;
    c.M = n * h2 * w2;
    c.kt_count = nr * ns * (cout / 64);
    K_TRY(spk_conv_launch(c, CONV_MODE_DGRAD, s, nullptr), "conv dgrad (parity class)");
  }
  return SPK_OK;
}

size_t spk_conv_wgrad_slab_floats(int M, int cin, int cout, int k, bool stem) {
  const int ktot = stem ? 256 : k * k * cin;
  int sp, pps;
  spk_wgrad_plan(M, cout, ktot, &sp, &pps);
  return (size_t)sp * cout * ktot;
}

// split-K partial weight gradients into `slabs` ...
int spk_conv_wgrad_slabs(const bf16_t* x, const bf16_t* dy, float* slabs, int n, int ih, int iw, int cin, int oh,
                         int ow, int cout, int k, int stride, int pad, bool stem, hipStream_t s) {
  const int ktot = stem ? 256 : k * k * cin;
  int sp, pps;
  spk_wgrad_plan(n * oh * ow, cout, ktot, &sp, &pps);
  K_TRY(spk_wgrad_launch(x, dy, slabs, n, ih, iw, cin, oh, ow, cout, k, stride, pad, stem ? 1 : 0, sp, pps, s),
        "conv wgrad");
  return SPK_OK;
}

// ... and their fixed-order sum into gw ([Cout][kh][kw][Cin] fp32)
int spk_conv_wgrad_reduce(const float* slabs, float* gw, int M, int cin, int cout, int k, bool stem, hipStream_t s,
                          float scale) {
  const int ktot = stem ? 256 : k * k * cin;
  int sp, pps;
  spk_wgrad_plan(M, cout, ktot, &sp, &pps);
  if (stem)
    K_TRY(spk_launch_stem_wgrad_unpack(slabs, gw, cout, k, k, cin, sp, s, scale), "stem wgrad unpack");
  else
    K_TRY(spk_launch_slab_reduce(slabs, gw, (size_t)cout * ktot, sp, s), "wgrad reduce");
  return SPK_OK;
}

static int grad_bucket_of(const spk_model* m, const Layer& L);
static int grad_bucket_done(spk_model* m, int b);

extern "C" int spk_train_forward_backward(spk_model* m, const void* x, int n, int h, int w, int layout,
                                          int dtype, const int64_t* y, float* stats, float* logits_out) {
  if (!m || !x || !y || !stats || n < 1) return tfail(SPK_ERR_ARG, "train step: bad arguments");
  if (dtype != SPK_DTYPE_F32 && dtype != SPK_DTYPE_U8) return tfail(SPK_ERR_ARG, "train step: dtype must be f32 or u8");
  HIP_TRY(hipSetDevice(m->device));
  SPK_TRY(ensure_state(m));
  SPK_TRY(plan_train(m, n, h, w));
  SPK_TRY(repack_weights(m));
  TrainState* t = m->train;
  hipStream_t s = m->stream;
  // train-mode BatchNorm2d needs more than one value per channel (torch raises the same ValueError, e.g. for a
  // last batch of ONE image whose feature map has shrunk to 1x1); a batch of one larger image trains, as in torch
  for (const Layer& L : m->layers) {
    const TDim& o = m->tdims[L.d.dst];
    if ((L.d.kind == SPK_OP_CONV || L.d.kind == SPK_OP_DWCONV) && (long)n * o.h * o.w < 2)
      return tfail(SPK_ERR_ARG, std::string("Expected more than 1 value per channel when training, got input size [") +
                                    std::to_string(n) + ", " + std::to_string(L.d.cout) + ", " + std::to_string(o.h) +
                                    ", " + std::to_string(o.w) + "] at " + L.d.bn);
  }
  m->act_dt = DT_BF16;
  // every activation is really written by this pass: spk_model_read_activation must not "recompute" tensors that an earlier
  // EVAL forward on this handle left to a fused kernel (stem + pool, shortcut conv, squeeze-excitation scaling)
  m->last_eval_nb = 0;
  float* part = (float*)((char*)t->arena + t->part_off);
  float* coef = (float*)((char*)t->arena + t->coef_off);
  float* tmp = (float*)((char*)t->arena + t->tmp_off);
  float* slabs = (float*)((char*)t->arena + t->slab_off);
  bf16_t* dy = (bf16_t*)((char*)t->arena + t->dy_off);
  bf16_t* const dy_bufs[2] = {dy, (bf16_t*)((char*)t->arena + t->dy2_off)};
  const bool side_on = t->side != nullptr && !t->prof.on;   // (the per-phase profile times a single-stream step)
  t->dy_busy[0] = t->dy_busy[1] = false;
  t->dy_slot = 0;
  unsigned char* pool_idx = (unsigned char*)((char*)t->arena + t->idx_off);
  const int nl = (int)m->layers.size();
  int squeezed = -1;   // the squeeze-excitation layer whose pooled means the depthwise layer before it has written

  // ------------------------------ forward ------------------------------
  mark(m, -1);
  // pixel values x 255: exact in bf16 (k / 255 is not: 8 mantissa bits put the input 0.2-0.4 % off); the stem's packed
  // weights carry 1 / 255 and its weight gradient is scaled back (spk_common.h SPK_INPUT_SCALE)
  K_TRY(spk_launch_to_nhwc4(x, layout, dtype, n, m->in_chans, h, w, (bf16_t*)m->T(0), DT_BF16, s, SPK_INPUT_SCALE), "to_nhwc4");
  mark(m, PH_INPUT);
  for (int i = 0; i < nl; ++i) {
    Layer& L = m->layers[i];
    const TDim& in = m->tdims[L.d.src];
    const TDim& o = m->tdims[L.d.dst];
    switch (L.d.kind) {
      case SPK_OP_CONV: {
        const int C = o.c;   // channels as laid out (= cout for the ResNets)
        const int M = n * o.h * o.w;
        float* st = t->stats + t->conv[i].stat_off;
        int m_tiles = 0;
        if (L.mode == CONV_MODE_STEM3) {
          K_TRY(spk_launch_stem3_train_fwd((const bf16_t*)m->T(0), m->P(L.p_w), t->RAW(i), n, in.h, w, in.w, L.d.cin,
                                           L.d.cout, C, o.h, o.w, s, 1.0f / SPK_INPUT_SCALE), "stem3 fwd");
          K_TRY(spk_launch_col_stats(t->RAW(i), part, M, C, &m_tiles, s), "col_stats");
        } else {
          ConvArgs a;
          fill_conv(a, (const bf16_t*)m->T(L.d.src), t->wpack + t->conv[i].wfwd_off, t->RAW(i), n, in.h, in.w,
                    in.c, o.h, o.w, C, L.d.k, L.d.stride, L.d.pad, L.kpad);
          a.stats = part;
          K_TRY(spk_conv_launch(a, L.mode, s, &m_tiles), "conv");
        }
        mark(m, PH_CONV_FWD);
        L.nbt += 1;
        if (!m->effnet) {
          K_TRY(spk_launch_bn_finalize(part, m_tiles, C, (double)M, m->P(L.p_g), m->P(L.p_b),
                                       m->P(L.p_mean), m->P(L.p_var), st, st + C, st + 2 * C, st + 3 * C,
                                       m->bn_eps, m->bn_momentum, tmp, s), "bn_finalize");
          K_TRY(spk_launch_bn_apply(t->RAW(i), st + 2 * C, st + 3 * C,
                                    L.d.res >= 0 ? (const bf16_t*)m->T(L.d.res) : nullptr,
                                    (bf16_t*)m->T(L.d.dst), t->MASK(i), (size_t)M * C, C, L.d.relu, s), "bn_apply");
        } else {
          K_TRY(spk_launch_bna_finalize(part, m_tiles, C, L.d.cout, (double)M, m->P(L.p_g), m->P(L.p_b), m->P(L.p_mean),
                                        m->P(L.p_var), st, m->bn_eps, m->bn_momentum, tmp, s), "bn_finalize");
          float* rs = nullptr;
          if (t->conv[i].rs_off) {   // StochasticDepth(p, "row") on the residual branch, train mode
            rs = (float*)((char*)t->arena + t->conv[i].rs_off);
            K_TRY(spk_launch_sd_rowscale(rs, n, L.d.p,
                                         (m->seed * 0x100000001B3ull) ^ (t->steps << 12) ^ (unsigned long long)i, s),
                  "stochastic depth");
          }
          K_TRY(spk_launch_bna_apply(t->RAW(i), st + 2 * C, st + 3 * C,
                                     L.d.res >= 0 ? (const bf16_t*)m->T(L.d.res) : nullptr, rs, (bf16_t*)m->T(L.d.dst), M,
                                     C, o.h * o.w, L.d.relu, s), "bn_apply");
        }
        mark(m, PH_BN_FWD);
        break;
      }
      case SPK_OP_DWCONV: {
        const int C = o.c, M = n * o.h * o.w;
        float* st = t->stats + t->conv[i].stat_off;
        int nbk = 0;
        // the eval path's kernel (window weights in LDS, four adjacent outputs per thread) with a = 1, b = 0, no
        // activation; -2: a window it does not have
        int r = L.d.pad == (L.d.k - 1) / 2
                    ? spk_launch_dwconv((const bf16_t*)m->T(L.d.src), t->dwt + t->conv[i].dwt_off, t->unit,
                                        t->unit + t->unit_c, t->RAW(i), nullptr, n, in.h, in.w, C, o.h, o.w, L.d.k,
                                        L.d.stride, 0, DT_BF16, s)
                    : -2;
        if (r == -2)
          r = spk_launch_dw_train_fwd((const bf16_t*)m->T(L.d.src), t->dwt + t->conv[i].dwt_off, t->RAW(i), n, in.h, in.w,
                                      C, L.d.k, L.d.stride, L.d.pad, o.h, o.w, s);
        K_TRY(r, "depthwise fwd");
        K_TRY(spk_launch_col_stats(t->RAW(i), part, M, C, &nbk, s), "col_stats");
        mark(m, PH_CONV_FWD);
        L.nbt += 1;
        K_TRY(spk_launch_bna_finalize(part, nbk, C, L.d.cout, (double)M, m->P(L.p_g), m->P(L.p_b), m->P(L.p_mean),
                                      m->P(L.p_var), st, m->bn_eps, m->bn_momentum, tmp, s), "bn_finalize");
        if (i + 1 < nl && m->layers[i + 1].d.kind == SPK_OP_SE && m->layers[i + 1].d.src == L.d.dst) {
          // the squeeze of the layer behind rides on this pass (per-chunk channel sums into the shared scratch; that
          // layer's first gate kernel turns them into its pooled means)
          K_TRY(spk_launch_bna_apply_pool(t->RAW(i), st + 2 * C, st + 3 * C, (bf16_t*)m->T(L.d.dst),
                                          (float*)((char*)t->arena + t->se_tmp_off), n, o.h * o.w, C, L.d.relu, s),
                "bn_apply + squeeze");
          squeezed = i + 1;
        } else {
          K_TRY(spk_launch_bna_apply(t->RAW(i), st + 2 * C, st + 3 * C, nullptr, nullptr, (bf16_t*)m->T(L.d.dst), M, C,
                                     o.h * o.w, L.d.relu, s), "bn_apply");
        }
        mark(m, PH_BN_FWD);
        break;
      }
      case SPK_OP_SE: {
        // s = sigmoid(fc2(silu(fc1(mean_hw(a))))), out = a * s; fp32 on the [n][C] vectors
        const int C = o.c, Cl = L.d.cout, S = L.d.k, HW = o.h * o.w;
        float* pooled = (float*)((char*)t->arena + t->conv[i].se_off);
        float* u1 = pooled + (size_t)n * C;
        float* h1 = u1 + (size_t)n * S;
        float* gate = h1 + (size_t)n * S;
        float* scratch = (float*)((char*)t->arena + t->se_tmp_off);
        const bf16_t* a = (const bf16_t*)m->T(L.d.src);
        if (squeezed != i)
          K_TRY(spk_launch_pool_rows(a, nullptr, scratch, n, HW, C, s), "se pool");
        K_TRY(spk_launch_se_gate_fwd(scratch, spk_se_chunks(HW), 1.f / (float)HW, pooled, m->P(L.p_w), m->P(L.p_b), m->P(L.p_w2), m->P(L.p_b2), u1, h1, gate, n, C, Cl, S,
                                     s), "se gates");
        K_TRY(spk_launch_se_scale(a, gate, (bf16_t*)m->T(L.d.dst), n, HW, C, s), "se scale");
        mark(m, PH_BN_FWD);
        break;
      }
      case SPK_OP_MAXPOOL:
        K_TRY(spk_launch_maxpool_idx((const bf16_t*)m->T(L.d.src), (bf16_t*)m->T(L.d.dst), pool_idx, n, in.h,
                                     in.w, in.c, L.d.k, L.d.stride, L.d.pad, o.h, o.w, s), "maxpool");
        mark(m, PH_POOL_FWD);
        break;
      case SPK_OP_GAVGPOOL:
        K_TRY(spk_launch_gavgpool((const bf16_t*)m->T(L.d.src), (float*)m->T(L.d.dst), n, in.h * in.w, in.c,
                                  DT_BF16, s), "avgpool");
        mark(m, PH_POOL_FWD);
        break;
      case SPK_OP_LINEAR:
        K_TRY(spk_launch_linear_fwd((const float*)m->T(L.d.src), m->P(L.p_w), m->P(L.p_b),
                                    (float*)m->T(L.d.dst), n, L.d.cin, L.d.cout, s), "linear");
        mark(m, PH_HEAD_FWD);
        break;
      case SPK_OP_DROPOUT:  // nn.Dropout(p) between head layers, train mode (reference network.py:59-61)
        K_TRY(spk_launch_dropout_fwd((const float*)m->T(L.d.src), (float*)m->T(L.d.dst), t->MASK(i), (size_t)n * in.c,
                                     L.d.p, (m->seed * 0x100000001B3ull) ^ (t->steps << 8) ^ (unsigned long long)i, s),
              "dropout");
        mark(m, PH_HEAD_FWD);
        break;
      default:
        HIP_TRY(hipMemcpyAsync(m->T(L.d.dst), m->T(L.d.src), (size_t)n * in.c * 4, hipMemcpyDeviceToDevice, s));
        break;
    }
  }
  t->steps += 1;
  m->dirty = true;  // running statistics moved: the eval-BN fold is stale

  // ------------------------- loss + dlogits -------------------------
  const int last = m->layers.back().d.dst;
  const float* logits = (const float*)m->T(last);
  K_TRY(spk_launch_ce(logits, y, n, m->num_classes, stats, (float*)t->G(last), s), "cross-entropy");
  mark(m, PH_LOSS);
  if (logits_out)
    HIP_TRY(hipMemcpyAsync(logits_out, logits, (size_t)n * m->num_classes * 4, hipMemcpyDeviceToDevice, s));

  // ------------------------------ backward ------------------------------
  std::vector<char> has_grad(m->n_tensors, 0);
  has_grad[last] = 1;
  // needs[t]: some parameter that the optimizer updates sits at or upstream of the layer that produces tensor t, i.e.
  // dL/dt changes something.  Gradients nobody needs are not computed, as with autograd.  NOTE the reference's own
  // schedule never gets there: its freeze() (sykepic/train/network.py:149-172) leaves every BatchNorm of the "frozen"
  // base trainable and train.py:131 puts them into param group 0, so the data-gradient chain runs down to the stem's
  // BatchNorm from the first epoch on (head-only epochs: 20.3 ms per ResNet-50 step at batch 256 against 26.9 with
  // every conv trainable - only the weight-gradient kernels drop out).  The cut applies when a caller freezes the
  // BatchNorm layers as well, or keeps them out of the optimizer.  A parameter counts when it requires grad AND an
  // optimizer group holds it (without any optimizer every trainable tensor is in group 0).
  std::vector<char> needs(m->n_tensors, 0);
  bool any_group = false;
  for (const Param& p : m->params) any_group |= p.trainable && p.group >= 0;
  auto trainable = [&](const Layer& Q) {
    for (int pi : {Q.p_w, Q.p_g, Q.p_b, Q.p_w2, Q.p_b2})
      if (pi >= 0 && m->params[pi].requires_grad && (!any_group || m->params[pi].group >= 0)) return true;
    return false;
  };
  for (const Layer& Q : m->layers)
    needs[Q.d.dst] = trainable(Q) || needs[Q.d.src] || (Q.d.kind == SPK_OP_CONV && Q.d.res >= 0 && needs[Q.d.res]);
  // BatchNorm-backward reductions that ride on the data-gradient kernel of the consumer (conv_igemm.hip, bnb_raw): when the
  // stride-1 dgrad of layer i is the LAST writer of dL/dt (no consumer of t comes earlier in the graph) and t is the output
  // of conv layer P, its epilogue holds the complete fp32 gradient and emits P's per-channel sums; P's own reduce pass -
  // one more read of the gradient and of P's raw output - is skipped.  fused_tiles[P] = partial rows waiting in `fpart`.
  // SPK_BNB_FUSE: bit 0 = 1x1 consumers that accumulate into a trunk gradient, bit 1 = other 1x1 consumers, bit 2 = 3x3
  // consumers; 0 = never (every BatchNorm backward runs its own reduce pass).  Default 3, measured per category on one box
  // (ResNet-50 step, ms): none 23.54 | trunk 23.11 | other 1x1 23.48 | 3x3 23.77 | trunk + other 1x1 23.07 | all 23.11 -
  // the trunk tensors are where the dropped pass is large (4 of a block's 6 tensor units) and the 1x1 data gradient short;
  // a 3x3 data gradient is MFMA-bound with one block per CU, and the longer epilogue costs more than the pass it replaces
  static const int bnb_mask = getenv("SPK_BNB_FUSE") ? atoi(getenv("SPK_BNB_FUSE")) : 3;
  const bool bnb_on = bnb_mask != 0;
  std::vector<int> fused_tiles(nl, 0);
  // deferred[t] = i: the block-closing conv i did NOT write its shortcut gradient dz into dL/dt; the fused data gradient of
  // t's other consumer (the block's first 1x1 conv) reads dz from its source - conv i's output gradient and ReLU bits - and
  // writes the complete tensor once (one write and one read of every identity-block trunk gradient less)
  static const bool defer_on = !getenv("SPK_BNB_DEFER") || atoi(getenv("SPK_BNB_DEFER")) != 0;
  std::vector<int> deferred(m->n_tensors, -1);
  int fpart_owner = -1;
  float* fpart = (float*)((char*)t->arena + t->fpart_off);
  auto fuse_target = [&](int i) -> int {   // producer layer whose reduction dgrad(i) can carry, or -1
    const Layer& L = m->layers[i];
    if (!bnb_on || m->effnet || L.d.stride != 1 || fpart_owner >= 0) return -1;
    const int cat = L.d.k == 1 ? (has_grad[L.d.src] || deferred[L.d.src] >= 0 ? 1 : 2) : 4;
    if (!(bnb_mask & cat)) return -1;
    int prod = -1;
    for (int q = 0; q < i; ++q) {
      const Layer& Q = m->layers[q];
      if (Q.d.src == L.d.src || (Q.d.kind == SPK_OP_CONV && Q.d.res == L.d.src)) return -1;   // an earlier consumer writes later
      if (Q.d.dst == L.d.src) prod = q;
    }
    if (prod < 0 || m->layers[prod].d.kind != SPK_OP_CONV || m->layers[prod].d.relu > 1) return -1;
    return prod;
  };
  // may block-closing conv i leave its shortcut gradient to the data gradient of the tensor's other consumer?
  auto can_defer = [&](int i) -> bool {
    const Layer& L = m->layers[i];
    if (!defer_on || !(bnb_mask & 1) || m->effnet || L.d.res < 0 || !L.d.relu || has_grad[L.d.res] || !needs[L.d.res]) return false;
    const int tr = L.d.res;
    int other = -1, count = 0, prod = -1;
    for (int q = 0; q < nl; ++q) {
      const Layer& Q = m->layers[q];
      if (Q.d.dst == tr) prod = q;
      if (Q.d.src == tr || (Q.d.kind == SPK_OP_CONV && Q.d.res == tr)) { ++count; if (q != i) other = q; }
    }
    if (count != 2 || other < 0 || other > i || prod < 0 || prod > other) return false;
    const Layer& Q = m->layers[other];
    const Layer& P = m->layers[prod];
    return Q.d.kind == SPK_OP_CONV && Q.mode == CONV_MODE_GENERIC && Q.d.src == tr && Q.d.k == 1 && Q.d.stride == 1 &&
           P.d.kind == SPK_OP_CONV && P.d.relu <= 1;
  };
  // deferred join of the side stream (TrainState::tail_pending): single-process steps whose gradients nobody reads in place
  static const bool tail_env = !getenv("SPK_TAIL_DEFER") || atoi(getenv("SPK_TAIL_DEFER")) != 0;
  const bool tail_ok = tail_env && side_on && !m->grad_cb && !t->grads_exported;
  int first_conv = -1, tail_layer = -1;
  for (int q = 0; q < nl && first_conv < 0; ++q)
    if (m->layers[q].d.kind == SPK_OP_CONV || m->layers[q].d.kind == SPK_OP_DWCONV) first_conv = q;
  int cur_bucket = 0;
  for (int i = nl - 1; i >= 0; --i) {
    Layer& L = m->layers[i];
    const TDim& in = m->tdims[L.d.src];
    const TDim& o = m->tdims[L.d.dst];
    if (m->grad_cb && (L.d.kind == SPK_OP_CONV || L.d.kind == SPK_OP_LINEAR)) {
      const int b = grad_bucket_of(m, L);
      while (cur_bucket < b) SPK_TRY(grad_bucket_done(m, cur_bucket++));   // backward has left that stage
    }
    if (!has_grad[L.d.dst] || !needs[L.d.dst]) continue;
    switch (L.d.kind) {
      case SPK_OP_LINEAR: {
        const float* gy = (const float*)t->G(L.d.dst);
        const float* xin = (const float*)m->T(L.d.src);
        const int fin = L.d.cin, fout = L.d.cout;
        if (m->params[L.p_w].requires_grad)  // dW[o][i] = sum_n gy[n][o] * x[n][i]
          K_TRY(spk_launch_sgemm(gy, 1, fout, xin, 1, fin, nullptr, t->gbuf + m->params[L.p_w].off, fin, 1,
                                 fout, fin, n, 1.f, 0, s), "linear wgrad");
        if (m->params[L.p_b].requires_grad)
          K_TRY(spk_launch_colsum(gy, t->gbuf + m->params[L.p_b].off, n, fout, s), "bias grad");
        if (needs[L.d.src]) {   // dX[n][i] = sum_o gy[n][o] * W[o][i]
          K_TRY(spk_launch_sgemm(gy, fout, 1, m->P(L.p_w), 1, fin, nullptr, (float*)t->G(L.d.src), fin, 1, n,
                                 fin, fout, 1.f, 0, s), "linear dgrad");
          has_grad[L.d.src] = 1;
        }
        mark(m, PH_HEAD_BWD);
        break;
      }
      case SPK_OP_DROPOUT:
        K_TRY(spk_launch_dropout_bwd((const float*)t->G(L.d.dst), t->MASK(i), (float*)t->G(L.d.src), (size_t)n * in.c,
                                     L.d.p, s), "dropout bwd");
        mark(m, PH_HEAD_BWD);
        has_grad[L.d.src] = 1;
        break;
      case SPK_OP_GAVGPOOL:
        K_TRY(spk_launch_gavgpool_bwd((const float*)t->G(L.d.dst), (bf16_t*)t->G(L.d.src), n, in.h * in.w,
                                      in.c, s), "avgpool bwd");
        mark(m, PH_POOL_BWD);
        has_grad[L.d.src] = 1;
        break;
      case SPK_OP_MAXPOOL:
        K_TRY(spk_launch_maxpool_bwd((const bf16_t*)t->G(L.d.dst), pool_idx, (bf16_t*)t->G(L.d.src), n, in.h,
                                     in.w, in.c, L.d.k, L.d.stride, L.d.pad, o.h, o.w, s), "maxpool bwd");
        mark(m, PH_POOL_BWD);
        has_grad[L.d.src] = 1;
        break;
      case SPK_OP_CONV:
      case SPK_OP_DWCONV: {
        const int C = o.c, M = n * o.h * o.w;   // C: channels as laid out
        float* st = t->stats + t->conv[i].stat_off;
        const Param& pg = m->params[L.p_g];
        const Param& pb = m->params[L.p_b];
        bf16_t* g_res = L.d.res >= 0 && needs[L.d.res] ? (bf16_t*)t->G(L.d.res) : nullptr;
        const bool defer = L.d.kind == SPK_OP_CONV && g_res && can_defer(i);
        if (defer) {   // the shortcut gradient is picked up at its source by the data gradient that completes dL/d(res)
          g_res = nullptr;
          deferred[L.d.res] = i;
        }
        float* dgam = pg.requires_grad ? t->gbuf + pg.off : nullptr;
        float* dbet = pb.requires_grad ? t->gbuf + pb.off : nullptr;
        int slot = 0;
        if (side_on) {   // this layer's dy buffer: free once the side-stream wgrad of two layers ago has read it
          slot = t->dy_slot;
          t->dy_slot ^= 1;
          if (t->conv[i].dy_off) {   // its own tensor: nothing to wait for (the event ring below only orders dy -> wgrad)
            dy = (bf16_t*)((char*)t->arena + t->conv[i].dy_off);
          } else {
            dy = dy_bufs[slot];
            if (t->dy_busy[slot]) HIP_TRY(hipStreamWaitEvent(s, t->ev_dy_free[slot], 0));
          }
        }
        if (!m->effnet) {
          const int pre = fused_tiles[i];   // sums already made by the dgrad that completed this layer's output gradient
          K_TRY(spk_launch_bn_bwd((const bf16_t*)t->G(L.d.dst), t->MASK(i), t->RAW(i), st, st + C, m->P(L.p_g),
                                  pre ? fpart : part, coef, dgam, dbet, dy, g_res, g_res ? has_grad[L.d.res] : 0, M, C,
                                  L.d.relu, tmp, s, pre), "bn bwd");
          if (pre) { fused_tiles[i] = 0; fpart_owner = -1; }
        } else {
          const float* rs = t->conv[i].rs_off ? (const float*)((char*)t->arena + t->conv[i].rs_off) : nullptr;
          const bf16_t* g = (const bf16_t*)t->G(L.d.dst);
          int nbk = 0;
          K_TRY(spk_launch_bna_bwd_reduce(g, t->RAW(i), st + 2 * C, st + 3 * C, st, st + C, rs, part, M, C, o.h * o.w,
                                          L.d.relu, &nbk, s), "bn bwd reduce");
          K_TRY(spk_launch_bna_bwd_finalize(part, nbk, C, L.d.cout, (double)M, m->P(L.p_g), st + C, dgam, dbet, coef, tmp, s),
                "bn bwd finalize");
          K_TRY(spk_launch_bna_bwd_apply(g, t->RAW(i), st + 2 * C, st + 3 * C, st, st + C, coef, rs, dy, g_res,
                                         g_res ? has_grad[L.d.res] : 0, M, C, o.h * o.w, L.d.relu, s), "bn bwd apply");
        }
        mark(m, PH_BN_BWD);
        // (6-9 us of idle main queue behind every one of these records; attaching the event to the apply kernel's own
        // completion - hipExtLaunchKernelGGL's stopEvent - leaves the same gap: measured in round 5, not kept)
        if (side_on) HIP_TRY(hipEventRecord(t->ev_dy_ready[slot], s));
        if (g_res) has_grad[L.d.res] = 1;
        const Param& pw = m->params[L.p_w];
        if (L.d.kind == SPK_OP_DWCONV) {
          const float* wt = t->dwt + t->conv[i].dwt_off;
          if (needs[L.d.src]) {
            // stride 1: dx = depthwise conv of dy with the flipped window (the forward kernel); else the gather kernel
            int r = L.d.stride == 1 && !has_grad[L.d.src] && L.d.pad == (L.d.k - 1) / 2
                        ? spk_launch_dwconv(dy, wt + (size_t)L.d.k * L.d.k * C, t->unit, t->unit + t->unit_c,
                                            (bf16_t*)t->G(L.d.src), nullptr, n, o.h, o.w, C, in.h, in.w, L.d.k, 1, 0,
                                            DT_BF16, s)
                        : -2;
            if (r == -2)
              r = spk_launch_dw_dgrad(dy, wt, (bf16_t*)t->G(L.d.src), has_grad[L.d.src] != 0, n, in.h, in.w, C, L.d.k,
                                      L.d.stride, L.d.pad, o.h, o.w, s);
            K_TRY(r, "depthwise dgrad");
            mark(m, PH_CONV_DGRAD);
            has_grad[L.d.src] = 1;
          }
          if (pw.requires_grad) {
            int rows = 0;
            const hipStream_t ws = side_on ? t->side : s;   // weight gradients: second stream (see the conv case below)
            if (side_on) HIP_TRY(hipStreamWaitEvent(ws, t->ev_dy_ready[slot], 0));
            K_TRY(spk_launch_dw_wgrad((const bf16_t*)m->T(L.d.src), dy, slabs, n, in.h, in.w, C, L.d.cout, L.d.k,
                                      L.d.stride, L.d.pad, o.h, o.w, &rows, ws), "depthwise wgrad");
            mark(m, PH_CONV_WGRAD);
            K_TRY(spk_launch_slab_reduce(slabs, t->gbuf + pw.off, (size_t)L.d.cout * L.d.k * L.d.k, rows, ws),
                  "depthwise wgrad reduce");
            if (side_on) {
              if (!t->conv[i].dy_off) {
                HIP_TRY(hipEventRecord(t->ev_dy_free[slot], ws));
                t->dy_busy[slot] = true;
              }
            }
            mark(m, PH_WGRAD_REDUCE);
          }
          break;
        }
        if (L.d.src != 0 && needs[L.d.src]) {
          // data gradient: implicit GEMM over the dgrad weight image (stride 2: one launch per parity class)
          const int pi = fuse_target(i);
          const int dsrc = deferred[L.d.src];
          if (dsrc >= 0 && pi < 0)
            return tfail(SPK_ERR_STATE, std::string("deferred shortcut gradient of ") + m->layers[dsrc].d.name + " has no fused data gradient to land in");
          BnbFuse fz;
          fz.res_src = nullptr;
          fz.res_bits = nullptr;
          if (pi >= 0) {
            const Layer& P = m->layers[pi];
            float* stp = t->stats + t->conv[pi].stat_off;
            fz.raw = t->RAW(pi);
            fz.mask = P.d.relu ? t->MASK(pi) : nullptr;
            fz.mean = stp;
            fz.invstd = stp + in.c;
            fz.partials = fpart;
            fz.tiles = 0;
            if (dsrc >= 0) {
              fz.res_src = (const bf16_t*)t->G(m->layers[dsrc].d.dst);
              fz.res_bits = t->MASK(dsrc);
              deferred[L.d.src] = -1;
            }
          }
          SPK_TRY(spk_conv_dgrad_all(dy, t->wpack + t->conv[i].wdg_off, (bf16_t*)t->G(L.d.src), has_grad[L.d.src] != 0, n,
                                     o.h, o.w, C, in.h, in.w, in.c, L.d.k, L.d.stride, L.d.pad, s, pi >= 0 ? &fz : nullptr));
          if (pi >= 0 && fz.tiles > 0) { fused_tiles[pi] = fz.tiles; fpart_owner = pi; }
          mark(m, PH_CONV_DGRAD);
          has_grad[L.d.src] = 1;
        }
        if (pw.requires_grad) {
          float* gw = t->gbuf + pw.off;
          if (L.mode == CONV_MODE_STEM3) {
            int nbk = 0;
            const hipStream_t ws3 = side_on ? t->side : s;
            if (side_on) HIP_TRY(hipStreamWaitEvent(ws3, t->ev_dy_ready[slot], 0));
            K_TRY(spk_launch_stem3_wgrad((const bf16_t*)m->T(0), dy, slabs, n, in.h, w, in.w, L.d.cin, L.d.cout, C, o.h,
                                         o.w, &nbk, ws3), "stem3 wgrad");
            mark(m, PH_CONV_WGRAD);
            K_TRY(spk_launch_slab_reduce(slabs, gw, (size_t)L.d.cout * 9 * L.d.cin, nbk, ws3, 1.0f / SPK_INPUT_SCALE),
                  "stem3 wgrad reduce");
            if (side_on) {
              if (!t->conv[i].dy_off) {
                HIP_TRY(hipEventRecord(t->ev_dy_free[slot], ws3));
                t->dy_busy[slot] = true;
              }
            }
            mark(m, PH_WGRAD_REDUCE);
            break;
          }
          const bool stem = L.mode == CONV_MODE_STEM;
          const int cin_t = stem ? L.d.cin : in.c;   // channels of the stored input tensor (the 7x7 stem reads NHWC4 itself)
          // ResNets: on the second stream, beside this layer's dgrad and the next layer's BatchNorm backward
          const hipStream_t ws = side_on ? t->side : s;
          if (side_on && tail_ok && L.d.src == 0 && i == first_conv) {   // the step's last weight gradient: see tail_pending
            HIP_TRY(hipEventRecord(t->ev_side_pre, ws));
            tail_layer = i;
          }
          if (side_on) HIP_TRY(hipStreamWaitEvent(ws, t->ev_dy_ready[slot], 0));
          SPK_TRY(spk_conv_wgrad_slabs((const bf16_t*)m->T(L.d.src), dy, slabs, n, in.h, in.w, cin_t, o.h, o.w, C, L.d.k,
                                       L.d.stride, L.d.pad, stem, ws));
          mark(m, PH_CONV_WGRAD);
          if (!stem && (C != L.d.cout || cin_t != L.d.cin)) {   // padded GEMM: keep the layer's own rows / columns
            int sp, pps;
            spk_wgrad_plan(M, C, L.d.k * L.d.k * cin_t, &sp, &pps);
            K_TRY(spk_launch_slab_reduce_sub(slabs, gw, L.d.cout, L.d.k * L.d.k, L.d.cin, C, cin_t, sp, ws),
                  "wgrad reduce (padded)");
          } else {
            SPK_TRY(spk_conv_wgrad_reduce(slabs, gw, M, cin_t, C, L.d.k, stem, ws, stem ? 1.0f / SPK_INPUT_SCALE : 1.0f));
          }
          if (side_on) {
            if (!t->conv[i].dy_off) {
              HIP_TRY(hipEventRecord(t->ev_dy_free[slot], ws));
              t->dy_busy[slot] = true;
            }
          }
          mark(m, PH_WGRAD_REDUCE);
        }
        break;
      }
      case SPK_OP_SE: {
        // out = a * s(pool(a)):  da = g * s + W1^T[ silu'(u1) * W2^T[ s(1-s) * sum_hw(g * a) ] ] / HW
        const int C = o.c, Cl = L.d.cout, S = L.d.k, HW = o.h * o.w;
        float* pooled = (float*)((char*)t->arena + t->conv[i].se_off);
        float* u1 = pooled + (size_t)n * C;
        float* h1 = u1 + (size_t)n * S;
        float* gate = h1 + (size_t)n * S;
        float* scratch = (float*)((char*)t->arena + t->se_tmp_off);
        float* dgate = gate + (size_t)n * C;   // becomes du2 in place
        float* du1 = dgate + (size_t)n * C;
        float* dpool = scratch + (size_t)n * spk_se_chunks(HW) * C;
        const bf16_t* g = (const bf16_t*)t->G(L.d.dst);
        const bf16_t* a = (const bf16_t*)m->T(L.d.src);
        K_TRY(spk_launch_pool_rows(g, a, scratch, n, HW, C, s), "se dgate");
        const Param &w1 = m->params[L.p_w], &b1 = m->params[L.p_b], &w2 = m->params[L.p_w2], &b2 = m->params[L.p_b2];
        K_TRY(spk_launch_se_gate_bwd(scratch, spk_se_chunks(HW), dgate, gate, u1, m->P(L.p_w), m->P(L.p_w2), du1, dpool, dpool + (size_t)n * C, n, C, Cl,
                                     S, s), "se gates bwd");
        hipStream_t ws = s;
        if (side_on && (w1.requires_grad || b1.requires_grad || w2.requires_grad || b2.requires_grad)) {
          ws = t->side;   // du2 / du1 are this layer's own: nothing on the main stream waits for the kernel
          HIP_TRY(hipEventRecord(t->ev_se, s));
          HIP_TRY(hipStreamWaitEvent(ws, t->ev_se, 0));
        }
        K_TRY(spk_launch_se_wgrad(dgate, h1, du1, pooled, w1.requires_grad ? t->gbuf + w1.off : nullptr,
                                  b1.requires_grad ? t->gbuf + b1.off : nullptr,
                                  w2.requires_grad ? t->gbuf + w2.off : nullptr,
                                  b2.requires_grad ? t->gbuf + b2.off : nullptr, n, C, Cl, S, ws), "se wgrad");
        K_TRY(spk_launch_se_bwd_apply(g, gate, dpool, (bf16_t*)t->G(L.d.src), n, HW, C, s), "se bwd apply");
        mark(m, PH_BN_BWD);
        has_grad[L.d.src] = 1;
        break;
      }
    }
  }
  if (m->grad_cb)
    while (cur_bucket < m->grad_buckets) SPK_TRY(grad_bucket_done(m, cur_bucket++));
  if (side_on) {   // the step is complete on the caller's stream only when the side-stream weight gradients are
    HIP_TRY(hipEventRecord(t->ev_side_done, t->side));
    if (tail_layer >= 0) {   // ... all but the last one: whoever needs it joins (spk_train_join, spk_optim_step)
      HIP_TRY(hipStreamWaitEvent(s, t->ev_side_pre, 0));
      t->tail_pending = true;
      t->tail_param = m->layers[tail_layer].p_w;
    } else {
      HIP_TRY(hipStreamWaitEvent(s, t->ev_side_done, 0));
    }
  }
  return SPK_OK;
}

// Makes the model's stream wait for whatever the last training step left running on the side stream (the deferred tail
// weight gradient).  Called by everything that reads gradients or overwrites what that kernel reads: the next forward
// of any kind (spk_plan), the gradient readers, a stream change.
int spk_train_join(spk_model* m) {
  TrainState* t = m ? m->train : nullptr;
  if (!t || !t->tail_pending) return SPK_OK;
  HIP_TRY(hipStreamWaitEvent(m->stream, t->ev_side_done, 0));
  t->tail_pending = false;
  return SPK_OK;
}

extern "C" int spk_model_set_grad_ready_callback(spk_model* m, spk_grad_ready_fn cb, void* user, void* comm_stream,
                                                 int n_buckets) {
  if (!m || (cb && (n_buckets < 1 || n_buckets > 3))) return tfail(SPK_ERR_ARG, "set_grad_ready_callback: bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  m->grad_cb = cb;
  m->grad_cb_user = user;
  m->comm_stream = (hipStream_t)comm_stream;
  m->grad_buckets = cb ? n_buckets : 0;
  for (int i = 0; i < 3; ++i)
    if (cb && i < n_buckets && !m->grad_ev[i]) HIP_TRY(hipEventCreateWithFlags(&m->grad_ev[i], hipEventDisableTiming));
  return SPK_OK;
}

// Bucket of the flat gradient buffer a layer's parameters belong to (0 = produced first by backward): with B
// buckets the last B-1 conv stages (by `child` index of the owning module) get one each, front to back.
static int grad_bucket_of(const spk_model* m, const Layer& L) {
  if (m->grad_buckets <= 1) return 0;
  int top = -1;
  for (const Layer& Q : m->layers) if (Q.d.kind == SPK_OP_CONV) top = std::max(top, Q.d.child);
  if (L.d.child < 0) return 0;                       // head
  const int b = top - L.d.child;                     // 0 for the last conv stage
  return std::min(b, m->grad_buckets - 1);
}

// every kernel writing bucket `b` is enqueued: fence it for the communication stream and tell the caller
static int grad_bucket_done(spk_model* m, int b) {
  // slice of the flat buffer = offsets of the trainable tensors of the bucket's layers (contiguous: parameters are
  // laid out in state_dict order, stage by stage)
  size_t lo = (size_t)-1, hi = 0;
  for (const Param& p : m->params) {
    if (!p.trainable || p.layer < 0) continue;
    if (grad_bucket_of(m, m->layers[p.layer]) != b) continue;
    lo = std::min(lo, p.off);
    hi = std::max(hi, p.off + (size_t)((p.numel + 63) / 64 * 64));
  }
  if (lo == (size_t)-1) return SPK_OK;
  if (b == m->grad_buckets - 1) lo = 0;              // the last bucket reaches to the front of the buffer
  if (b == 0) hi = m->n_train;
  if (m->train && m->train->side) {   // weight gradients of the bucket's layers may still be running on the side stream
    HIP_TRY(hipEventRecord(m->train->ev_side_done, m->train->side));
    HIP_TRY(hipStreamWaitEvent(m->stream, m->train->ev_side_done, 0));
  }
  HIP_TRY(hipEventRecord(m->grad_ev[b], m->stream));
  if (m->comm_stream != m->stream) HIP_TRY(hipStreamWaitEvent(m->comm_stream, m->grad_ev[b], 0));
  m->grad_cb(m->grad_cb_user, b, (int64_t)lo, (int64_t)(hi - lo));
  return SPK_OK;
}

extern "C" int spk_optim_step(spk_model* m, const spk_optim_desc* opt) {
  if (!m || !opt) return tfail(SPK_ERR_ARG, "optim_step: bad arguments");
  if (opt->kind < SPK_OPT_SGD || opt->kind > SPK_OPT_RPROP)
    return tfail(SPK_ERR_UNSUPPORTED,
                 "optimizer kind not supported (SGD, Adam, AdamW, RMSprop, Adagrad, Adamax, NAdam, RAdam, Adadelta, ASGD, Rprop)");
  if (!m->train) return tfail(SPK_ERR_STATE, "optim_step before any training step");
  HIP_TRY(hipSetDevice(m->device));
  TrainState* t = m->train;
  const float gscale = opt->grad_scale == 0.f ? 1.f : opt->grad_scale;
  const double b1 = opt->beta1, b2 = opt->beta2;
  OptTable tab;
  tab.count = 0;
  auto flush = [&]() -> int {
    if (tab.count == 0) return 0;
    const int r = spk_launch_opt_multi(opt->kind, m->pbuf, t->gbuf, t->m1, t->m2, tab, opt->beta1, opt->beta2,
                                       opt->eps, opt->weight_decay, opt->momentum, gscale, opt->alpha, m->stream);
    tab.count = 0;
    return r;
  };
  // parameters whose gradients are complete first; the one the side stream may still be writing last, behind the join
  const int tail_param = t->tail_pending ? t->tail_param : -1;
  for (int pass = 0; pass < 2; ++pass)
  for (size_t pidx = 0; pidx < m->params.size(); ++pidx) {
    Param& p = m->params[pidx];
    if (((int)pidx == tail_param) != (pass == 1)) continue;
    if (pass == 1) {
      K_TRY(flush(), "optimizer");
      SPK_TRY(spk_train_join(m));
    }
    if (!p.trainable || !p.requires_grad || p.group < 0) continue;
    p.step += 1;
    OptEntry& e = tab.e[tab.count++];
    e.off = p.off;
    e.n = (unsigned)p.numel;
    e.lr = opt->lr[p.group];
    const double bc1 = 1.0 - std::pow(b1, (double)p.step), bc2 = 1.0 - std::pow(b2, (double)p.step);
    e.bc1 = (float)bc1;
    e.bc2s = (float)std::sqrt(bc2);
    e.first = p.step == 1;
    e.c0 = e.c1 = 0.f;
    if (opt->kind == SPK_OPT_ADAGRAD) {
      e.c0 = (float)((double)e.lr / (1.0 + (double)(p.step - 1) * opt->lr_decay));
      e.c1 = opt->initial_accumulator_value;
    } else if (opt->kind == SPK_OPT_NADAM) {
      // torch.optim.NAdam: mu_t = beta1 (1 - 0.5 * 0.96^(t * momentum_decay))
      const double md = opt->momentum_decay;
      const double mu = b1 * (1.0 - 0.5 * std::pow(0.96, (double)p.step * md));
      const double mu_next = b1 * (1.0 - 0.5 * std::pow(0.96, (double)(p.step + 1) * md));
      p.mu_product *= mu;
      e.c0 = (float)((double)e.lr * (1.0 - mu) / (1.0 - p.mu_product));
      e.c1 = (float)((double)e.lr * mu_next / (1.0 - p.mu_product * mu_next));
    } else if (opt->kind == SPK_OPT_ASGD) {
      // torch.optim.ASGD: the step uses eta of the PREVIOUS step, eta_0 = lr, eta_t = lr / (1 + lambd lr t)^alpha
      const double lam = opt->lr_decay, t1 = (double)(p.step - 1);
      e.c0 = (float)((double)e.lr / std::pow(1.0 + lam * (double)e.lr * t1, (double)opt->alpha));
      e.c1 = (float)lam;
    } else if (opt->kind == SPK_OPT_RADAM) {
      const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
      const double rho_t = rho_inf - 2.0 * (double)p.step * std::pow(b2, (double)p.step) / bc2;
      if (rho_t > 5.0) {
        const double rect = std::sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
        e.c0 = (float)((double)e.lr * rect * std::sqrt(bc2));
        e.c1 = 1.f;
      }
    }
    if (tab.count == 64) K_TRY(flush(), "optimizer");
  }
  K_TRY(flush(), "optimizer");
  SPK_TRY(spk_train_join(m));
  t->weights_dirty = true;
  m->dirty = true;
  return SPK_OK;
}

extern "C" int spk_model_grad_buffer(spk_model* m, void** dev_ptr, int64_t* numel) {
  if (!m || !dev_ptr || !numel) return tfail(SPK_ERR_ARG, "grad_buffer: bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  SPK_TRY(ensure_state(m));
  SPK_TRY(spk_train_join(m));
  m->train->grads_exported = true;   // the caller reads the buffer on its own stream from now on: steps end joined
  *dev_ptr = m->train->gbuf;
  *numel = (int64_t)m->n_train;
  return SPK_OK;
}

extern "C" int spk_model_read_grad(spk_model* m, const char* key, void* host, int64_t numel) {
  if (!m || !key || !host) return tfail(SPK_ERR_ARG, "read_grad: bad arguments");
  auto it = m->index.find(key);
  if (it == m->index.end()) return tfail(SPK_ERR_KEY, std::string("unknown state_dict key: ") + key);
  const Param& p = m->params[it->second];
  if (!p.trainable || p.numel != numel) return tfail(SPK_ERR_ARG, std::string("no gradient / size mismatch for ") + key);
  if (!m->train) return tfail(SPK_ERR_STATE, "read_grad before any training step");
  HIP_TRY(hipSetDevice(m->device));
  SPK_TRY(spk_train_join(m));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return spk_read_flat(m, m->train->gbuf, p, (float*)host);
}

// Test hook: gradient w.r.t. activation `t` of the last training step, as
// float32 NCHW ([n,c,h,w]; [n,c] in the head).
extern "C" int spk_model_read_activation_grad(spk_model* m, int t, int n, float* host, int64_t numel) {
  if (!m || !host || !m->train || !m->train->arena || t <= 0 || t >= m->n_tensors || n > m->train->cap_n)
    return tfail(SPK_ERR_ARG, "read_activation_grad: bad arguments or no training step has run");
  const TDim& d = m->tdims[t];
  const int cl = d.c_log > 0 ? d.c_log : d.c;   // the caller sees the layer's own channels, not the padding
  const size_t cnt = (size_t)n * d.h * d.w * d.c;
  if ((int64_t)((size_t)n * d.h * d.w * cl) != numel) return tfail(SPK_ERR_ARG, "read_activation_grad: size mismatch");
  HIP_TRY(hipSetDevice(m->device));
  SPK_TRY(spk_train_join(m));
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (!d.bf16) {
    HIP_TRY(hipMemcpy(host, m->train->G(t), cnt * 4, hipMemcpyDeviceToHost));
    return SPK_OK;
  }
  std::vector<bf16_t> tmp(cnt);
  HIP_TRY(hipMemcpy(tmp.data(), m->train->G(t), cnt * 2, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i)
    for (int y = 0; y < d.h; ++y)
      for (int x = 0; x < d.w; ++x)
        for (int c = 0; c < cl; ++c) {
          const unsigned u = (unsigned)tmp[(((size_t)i * d.h + y) * d.w + x) * d.c + c] << 16;
          float f;
          memcpy(&f, &u, 4);
          host[(((size_t)i * cl + c) * d.h + y) * d.w + x] = f;
        }
  return SPK_OK;
}

// Per-phase timing of the training step (HIP events on the model's stream
// between kernel groups), averaged over `iters` steps after one warm-up.
extern "C" int spk_model_profile_train(spk_model* m, const void* x, int n, int h, int w, int layout,
                                       int dtype, const int64_t* y, float* stats, int iters,
                                       spk_layer_time* out, int cap) {
  if (!m || !out || cap < PH_COUNT || iters <= 0) return tfail(SPK_ERR_ARG, "profile_train: bad arguments");
  SPK_TRY(spk_train_forward_backward(m, x, n, h, w, layout, dtype, y, stats, nullptr));  // warm-up + plan
  TrainState* t = m->train;
  std::vector<double> ms(PH_COUNT, 0.0);
  std::vector<int> launches(PH_COUNT, 0);
  for (int it = 0; it < iters; ++it) {
    t->prof.on = true;
    t->prof.marks.clear();
    t->prof.used = 0;
    const int rc = spk_train_forward_backward(m, x, n, h, w, layout, dtype, y, stats, nullptr);
    t->prof.on = false;
    if (rc != SPK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (size_t k = 1; k < t->prof.marks.size(); ++k) {
      float dt = 0.f;
      HIP_TRY(hipEventElapsedTime(&dt, t->prof.pool[t->prof.marks[k - 1].second], t->prof.pool[t->prof.marks[k].second]));
      const int ph = t->prof.marks[k].first;
      if (ph >= 0) { ms[ph] += dt; launches[ph] += 1; }
    }
  }
  // algorithmic work of the conv phases
  double f_fwd = 0, f_dg = 0, f_wg = 0;
  for (const Layer& L : m->layers) {
    if (L.d.kind != SPK_OP_CONV) continue;
    const TDim& o = m->tdims[L.d.dst];
    const double f = 2.0 * n * o.h * o.w * (double)L.d.cout * L.d.cin * L.d.k * L.d.k;
    f_fwd += f;
    if (L.d.src != 0) f_dg += f;
    if (m->params[L.p_w].requires_grad) f_wg += f;
  }
  for (int ph = 0; ph < PH_COUNT; ++ph) {
    memset(&out[ph], 0, sizeof out[ph]);
    strncpy(out[ph].name, kPhaseName[ph], sizeof out[ph].name - 1);
    out[ph].ms = (float)(ms[ph] / iters);
    out[ph].flops = ph == PH_CONV_FWD ? f_fwd : ph == PH_CONV_DGRAD ? f_dg : ph == PH_CONV_WGRAD ? f_wg : 0.0;
    out[ph].bytes = (double)launches[ph] / iters;  // launches per step
  }
  return PH_COUNT;
}
