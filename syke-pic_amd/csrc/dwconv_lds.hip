// Depthwise KxK convolution + folded BatchNorm + activation with the input staged through LDS (gfx950).
// Reference op: the depthwise Conv2d(C, C, k, stride, groups=C) + BatchNorm2d + SiLU of torchvision's MBConv block,
// reached through `net(x)` (sykepic/compute/probability.py:189; model built at sykepic/train/network.py:48).
//
// Why: a depthwise layer moves ~2 bytes and does ~2 k^2 flops per element — both floors are ~25 us for the 28^2
// k5 layers of EfficientNet-B4 at batch 256 — but the first kernel (effnet.hip: every thread gathers its taps from
// global memory) re-fetched each input value ~10 x through L1/L2 and ran at 300-450 us: half of the network's time.
// Here a block owns a band of output rows of one image x a slab of channels and slides a ring of input rows through
// LDS: every input element is fetched from HBM/L2 once per band (+ the K-S halo rows at the band's start), with
// fully coalesced 16-B accesses (channels innermost), zero-filled borders (no bounds checks in the tap loop), and
// the taps are read from LDS (pixel stride padded by 16 B: conflict-free ds_read_b128 for 8/16/32 channel groups).
// A thread owns one 16-B channel group (8 fp16 / 16 e4m3 channels) x PX adjacent outputs; the K weights of a filter
// row sit in registers while the row's columns stream by.  The squeeze-excitation pool partials (fp32 sums of the
// outputs before rounding) are reduced over the block in fixed order, as before.
//
// ET = 0: fp16 in / out (the parity path).  ET = 1: e4m3 in / out (fp8 mode, pw_fp8.hip): the caller folds the
// input scale into the weights; the output is v * out_inv_scale.
#include "dw_util.h"

#include <algorithm>

namespace {

using namespace dwu;

struct DwArgs {
  const unsigned char* x;
  const float* w;       // [K*K][c_p] tap-major fp32
  const float* scale;   // [c_p]
  const float* bias;
  unsigned char* y;
  float* partial;       // [n][bands][c_p] or null
  int h, wid, c_p, ho, wo, act;
  int bands, rows_per_band, tcg, R;
  float w_scale, out_inv_scale;
};

template <int ET, int K, int S>
__global__ __launch_bounds__(256) void dwconv_lds_kernel(DwArgs a) {
  constexpr int PAD = (K - 1) / 2, PX = DwT<ET>::PX, CPT = DwT<ET>::CPT, ELEM = DwT<ET>::ELEM;
  constexpr int COLS = (PX - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tcg = a.tcg, lanes = 256 / tcg, tcs = tcg * CPT;    // channel groups / pixel lanes / channels of a full tile
  float* const sw = (float*)smem;                               // [K*K + 2][tcs]
  unsigned char* const ring = smem + (size_t)(K * K + 2) * tcs * 4;
  const int bands = a.bands;
  const int img = blockIdx.y / bands, band = blockIdx.y % bands;
  const int groups_total = a.c_p / CPT;
  const int cg0 = blockIdx.x * tcg;
  const int ncg = min(tcg, groups_total - cg0);
  const int tid = threadIdx.x;
  const int cg = tid % tcg, lane = tid / tcg;
  const bool ch_ok = cg < ncg;

  stage_dw_weights<K, CPT>(sw, a.w, a.scale, a.bias, a.c_p, cg0 * CPT, ncg, tcg, a.w_scale, tid);   // layout: dw_util.h

  const int R = a.R;
  const int NR = (R - 1) * S + K;             // input rows an iteration needs
  const int WP = a.wid + 2 * PAD;             // ring row: the image row + zero columns either side
  const int PSTRIDE = tcg * 16 + 16;          // bytes per ring pixel (+16: bank spread)
  const size_t RSTRIDE = (size_t)WP * PSTRIDE;
  const int gpr = (a.wo + PX - 1) / PX;       // groups of PX outputs per output row
  const int oy_begin = band * a.rows_per_band, oy_end = min(a.ho, oy_begin + a.rows_per_band);
  const unsigned char* xi = a.x + (size_t)img * a.h * a.wid * a.c_p * ELEM + (size_t)cg0 * 16;
  unsigned char* yi = a.y + (size_t)img * a.ho * a.wo * a.c_p * ELEM + (size_t)(cg0 + cg) * 16;

  float pool[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) pool[j] = 0.f;

  // Ring fill.  The first window of a band is copied straight into LDS; after each iteration the R*S new rows replace
  // the rows that left the window.  (A variant that fetched the next rows into 8 registers per thread before the
  // compute, to hide their latency under the FMAs, was measured: 32 more VGPRs, one wave per SIMD fewer, slower than
  // this one or than the gather kernel on every EfficientNet-B4 layer - removed.)
  const int per_row = WP * tcg;
  // copies `count` 16-byte chunks (rows new_lo.. of the padded window) into their ring slots, four per thread in
  // flight: the loads are unconditional (address clamped into the image, value zeroed by a select for the padding),
  // so the compiler issues all four before the first LDS store instead of a load-wait-store chain per chunk
  auto fill = [&](int new_lo, int count) {
    for (int base = 0; base < count; base += 4 * 256) {
      u32x4_t v[4];
      int off[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = min(base + j * 256 + tid, count - 1);
        const int row = idx / per_row, rem = idx - row * per_row;
        const int px = rem / tcg, g = rem - px * tcg;
        const int iy = new_lo + row, ix = px - PAD;
        off[j] = ((iy + PAD) % NR) * (int)RSTRIDE + px * PSTRIDE + g * 16;   // iy + PAD >= 0
        const bool ok = g < ncg && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.wid;
        const int iyc = min(max(iy, 0), a.h - 1), ixc = min(max(ix, 0), a.wid - 1), gc = min(g, ncg - 1);
        v[j] = *(const u32x4_t*)(xi + ((size_t)iyc * a.wid + ixc) * a.c_p * ELEM + (size_t)gc * 16);
        if (!ok) v[j] = u32x4_t{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (base + j * 256 + tid < count) *(u32x4_t*)(ring + off[j]) = v[j];
    }
  };
  fill(oy_begin * S - PAD, NR * per_row);
  int loaded_hi = oy_begin * S - PAD + NR;    // input rows [loaded_hi - NR, loaded_hi) are in the ring
  __syncthreads();                            // ring and sw are written
  for (int oyb = oy_begin; oyb < oy_end; oyb += R) {
    const bool have_next = oyb + R < oy_end;
    const int nlo = loaded_hi, nhi = (oyb + R) * S - PAD + NR;
    const int ncount = have_next ? (nhi - nlo) * per_row : 0;
    const int items = R * gpr;
    for (int item = lane; item < items; item += lanes) {
      const int r_sub = item / gpr, pg = item - r_sub * gpr;
      const int oy = oyb + r_sub;
      if (oy >= oy_end || !ch_ok) continue;
      const int ox0 = pg * PX;
      float acc[PX][CPT];
#pragma unroll
      for (int u = 0; u < PX; ++u)
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[u][j] = 0.f;
      // One filter row at a time (a REAL loop: unrolled, hipcc 7.2 hoists every LDS read of all K rows to the top and
      // spills ~1000 VGPRs); inside a row the columns are unrolled (the tap index of each output is a compile-time
      // constant) with the next column's 16 B fetched before the current column's FMAs and a scheduling fence per
      // column, so at most two columns are live.
#pragma unroll 1
      for (int r = 0; r < K; ++r) {
        const int slot = (oy * S + r) % NR;   // (iy + PAD) with iy = oy*S - PAD + r
        const unsigned char* rp = ring + slot * RSTRIDE + (size_t)(ox0 * S) * PSTRIDE + cg * 16;
        float wrow[K][CPT];
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const float* wp = sw + ((r * K + q) * (CPT / 4) * tcg + cg) * 4;
#pragma unroll
          for (int j4 = 0; j4 < CPT / 4; ++j4) {
            const f32x4_t t = *(const f32x4_t*)(wp + j4 * tcg * 4);
            wrow[q][4 * j4] = t[0]; wrow[q][4 * j4 + 1] = t[1]; wrow[q][4 * j4 + 2] = t[2]; wrow[q][4 * j4 + 3] = t[3];
          }
        }
        // columns past the padded row belong to outputs past wo (never stored): clamp the read inside the ring row
        const int cmax = WP - 1 - ox0 * S;
        u32x4_t raw = *(const u32x4_t*)(rp);
#pragma unroll
        for (int col = 0; col < COLS; ++col) {
          u32x4_t nxt = raw;
          if (col + 1 < COLS) nxt = *(const u32x4_t*)(rp + (size_t)min(col + 1, cmax) * PSTRIDE);
          float xv[CPT];
          unpack16B<ET>(raw, xv);
#pragma unroll
          for (int u = 0; u < PX; ++u) {
            const int q = col - u * S;
            if (q >= 0 && q < K) {
#pragma unroll
              for (int j = 0; j < CPT; ++j) acc[u][j] += xv[j] * wrow[q][j];
            }
          }
          raw = nxt;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int j4 = 0; j4 < CPT / 4; ++j4) {   // folded BN: 4 channels' scale and shift at a time (8 live registers)
        const f32x4_t t0 = *(const f32x4_t*)(sw + (((K * K) * (CPT / 4) + j4) * tcg + cg) * 4);
        const f32x4_t t1 = *(const f32x4_t*)(sw + (((K * K + 1) * (CPT / 4) + j4) * tcg + cg) * 4);
#pragma unroll
        for (int u = 0; u < PX; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[u][4 * j4 + j] = acc[u][4 * j4 + j] * t0[j] + t1[j];
      }
#pragma unroll
      for (int u = 0; u < PX; ++u) {
        if (ox0 + u >= a.wo) continue;
        if (a.act == 2) {   // uniform branch: one activation's instructions, not both + selects
#pragma unroll
          for (int j = 0; j < CPT; ++j) acc[u][j] = silu_f(acc[u][j]);
        } else if (a.act == 1) {
#pragma unroll
          for (int j = 0; j < CPT; ++j) acc[u][j] = fmaxf(acc[u][j], 0.f);
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) pool[j] += acc[u][j];
        *(u32x4_t*)(yi + ((size_t)oy * a.wo + ox0 + u) * a.c_p * ELEM) = pack16B<ET>(acc[u], a.out_inv_scale);
      }
    }
    __syncthreads();                          // every wave is done with the rows that leave the window
    if (have_next) {
      fill(nlo, ncount);
      loaded_hi = nhi;
    }
    __syncthreads();
  }
  if (!a.partial) return;
  __syncthreads();                            // ring is dead: it becomes [lanes][tcs] pool partials
  float* red = (float*)ring;
#pragma unroll
  for (int j = 0; j < CPT; ++j) red[lane * tcs + cg * CPT + j] = ch_ok ? pool[j] : 0.f;
  __syncthreads();
  for (int c = tid; c < ncg * CPT; c += 256) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * tcs + c];
    a.partial[((size_t)img * bands + band) * a.c_p + cg0 * CPT + c] = t;
  }
}

struct DwPlan {
  int tcg, R, bands, rows_per_band, ctiles;
  size_t lds;
};

// channel groups per block (a power of two dividing 256), output rows per iteration, bands per image
bool dw_plan(int et, int n, int wid, int c_p, int ho, int wo, int k, int s, DwPlan* p) {
  const int cpt = et ? 16 : 8, px = et ? 2 : 4, pad = (k - 1) / 2;
  if (c_p % cpt) return false;
  const int groups = c_p / cpt, gpr = (wo + px - 1) / px;
  double best_score = -1.0;
  DwPlan best{};
  for (int tcg = 32; tcg >= 4; tcg >>= 1) {
    if (et && tcg > 16) continue;             // 16 e4m3 groups = 256 channels
    const int lanes = 256 / tcg;
    int R = std::max(1, lanes / gpr);         // about one item per thread and iteration
    R = std::min(R, ho);
    const size_t red = (size_t)256 * cpt * 4;
    size_t lds = 0;
    bool ok = false;
    for (; R >= 1; --R) {                     // fewer output rows per iteration until the ring fits
      const int NR = (R - 1) * s + k;
      const size_t ring = (size_t)NR * (wid + 2 * pad) * (tcg * 16 + 16);
      lds = (size_t)(k * k + 2) * tcg * cpt * 4 + std::max(ring, red);
      if (lds <= 64 * 1024) { ok = true; break; }   // <= 64 KB: two to three blocks per CU
    }
    if (!ok) continue;
    const int ctiles = (groups + tcg - 1) / tcg;
    const double ch_util = (double)groups / (ctiles * tcg);
    const int items = R * gpr;
    const double lane_util = (double)items / (((items + lanes - 1) / lanes) * lanes);
    const double score = ch_util * lane_util + 0.01 * tcg / 32.0;   // ties: wider channel slabs (longer contiguous segments)
    if (score > best_score) {
      best_score = score;
      best.tcg = tcg; best.R = R; best.ctiles = ctiles; best.lds = lds;
    }
  }
  if (best_score < 0) return false;
  const int iters = (ho + best.R - 1) / best.R;
  int bands = (1024 + n * best.ctiles - 1) / (n * best.ctiles);   // ~4 blocks per CU; fewer bands = fewer halo rows
  bands = std::max(1, std::min(std::min(bands, 64), iters));
  const int ipb = (iters + bands - 1) / bands;
  best.rows_per_band = ipb * best.R;
  best.bands = (ho + best.rows_per_band - 1) / best.rows_per_band;
  *p = best;
  return true;
}

template <int ET, int K, int S>
int launch_dw(const DwArgs& a, const DwPlan& p, int n, hipStream_t s) {
  auto k = dwconv_lds_kernel<ET, K, S>;
  static std::atomic<unsigned long long> attr;
  (void)spk_lds_limit_once(attr, (const void*)k, 64 * 1024);
  hipLaunchKernelGGL(k, dim3(p.ctiles, n * p.bands), dim3(256), p.lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// Number of pool-partial rows per image ("chunks" of the squeeze-excitation kernels) the LDS kernel writes for this
// problem, or 0 when it cannot run it (the caller then uses the gather kernels).
int spk_dwconv_lds_chunks(int et, int n, int h, int wid, int c_p, int ho, int wo, int k, int stride) {
  DwPlan p;
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2) || !dw_plan(et, n, wid, c_p, ho, wo, k, stride, &p)) return 0;
  return p.bands;
}

// et 0: fp16 tensors, 1: e4m3 tensors.  w_scale multiplies the tap weights (fp8: the input tensor's scale).
int spk_launch_dwconv_lds(int et, const void* x, const float* w, const float* scale, const float* bias, void* y,
                          float* partial, int n, int h, int wid, int c_p, int ho, int wo, int k, int stride, int act,
                          float w_scale, float out_inv_scale, hipStream_t s) {
  DwPlan p;
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2) || !dw_plan(et, n, wid, c_p, ho, wo, k, stride, &p)) return -2;
  DwArgs a;
  a.x = (const unsigned char*)x; a.w = w; a.scale = scale; a.bias = bias; a.y = (unsigned char*)y; a.partial = partial;
  a.h = h; a.wid = wid; a.c_p = c_p; a.ho = ho; a.wo = wo; a.act = act;
  a.bands = p.bands; a.rows_per_band = p.rows_per_band; a.tcg = p.tcg; a.R = p.R;
  a.w_scale = w_scale; a.out_inv_scale = out_inv_scale;
#define SPK_DWL(ET, K, S) return launch_dw<ET, K, S>(a, p, n, s)
  if (et == 0) {
    if (k == 3 && stride == 1) SPK_DWL(0, 3, 1);
    if (k == 3) SPK_DWL(0, 3, 2);
    if (stride == 1) SPK_DWL(0, 5, 1);
    SPK_DWL(0, 5, 2);
  }
  if (k == 3 && stride == 1) SPK_DWL(1, 3, 1);
  if (k == 3) SPK_DWL(1, 3, 2);
  if (stride == 1) SPK_DWL(1, 5, 1);
  SPK_DWL(1, 5, 2);
#undef SPK_DWL
}
