// Weight gradient of a 2-D convolution on gfx950 (bf16 in, fp32 out).
//
// Part of `loss.backward()` at sykepic/train/train.py:242 for every conv whose
// weight has requires_grad (the unfreeze schedule of
// sykepic/train/network.py:98-130 decides which).
//
//   dW[co][r][s][ci] = sum over output pixels m=(n,ho,wo) of
//                      dy[m][co] * x[n, ho*stride-pad+r, wo*stride-pad+s, ci]
//
// GEMM view: rows = Cout, cols = (tap, Cin), contraction = pixels.  Both
// operands are pixel-major in memory (NHWC), i.e. k-strided for the MFMA, so
// the [pixel][channel] tiles are staged in LDS exactly as they are loaded
// (coalesced 16-B rows) and read back TRANSPOSED with ds_read_b64_tr_b16 — the
// hardware transpose gives each lane its 4 consecutive k for one channel.  The
// LDS images are XOR-swizzled so that the transposed reads are conflict-free.
// The pixel range is split over blockIdx.y; every split writes its own fp32
// slab and a second kernel adds the slabs in a fixed order (reproducible, no
// atomics).
#include "spk_common.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

struct WgradArgs {
  const bf16_t* x;
  const bf16_t* dy;
  float* slabs;  // [splits][Cout][Ktot]
  int N, H, W, Cin, Ho, Wo, Cout;
  int kh, kw, stride, pad;
  int M, Ktot;
  int pix_per_split;
  unsigned int x_bytes, dy_bytes;
  int xcd_map;   // 1: XCD-aware block order (all tiles of a pixel split on one XCD)
};

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <int WIDTH>
__device__ __forceinline__ int tile_off(int row, int ch) {
  if (WIDTH == 128)  // 256-B rows: T10 image (b)
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  // 128-B rows: two rows per 256-B bank period; spread row bit 1 and bit 3
  return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1));
}

// transposed fragment: 8 consecutive k (rows k0 + 8*(lane>>4) + 0..7) of column
// col0 + (lane&15)
template <int WIDTH>
__device__ __forceinline__ u32x4_t tr_frag(const unsigned char* tile, int k0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int ch = (col0 >> 3) + (p >> 1);
  const int r0 = k0 + 8 * g + q;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4_t __attribute__((address_space(3)))*)(tile + tile_off<WIDTH>(r0, ch) + 8 * (p & 1)));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4_t __attribute__((address_space(3)))*)(tile + tile_off<WIDTH>(r0 + 4, ch) + 8 * (p & 1)));
  u32x2_t a = __builtin_bit_cast(u32x2_t, lo), b = __builtin_bit_cast(u32x2_t, hi);
  return u32x4_t{a[0], a[1], b[0], b[1]};
}

// NBUF: LDS stages.  2: the next pixel block is stored while the current one is multiplied (one barrier per
// step).  1: half the LDS - twice the resident blocks, whose phases overlap instead (two barriers per step);
// which one is faster depends on the layer, spk_wgrad_launch times both once per problem.
template <int BCO, int BCI, bool STEM, int NBUF>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a, int co_tiles) {
  constexpr int BK = 64;  // pixels per LDS stage
  constexpr int WCO = BCO / 2, WCI = BCI / 2;
  constexpr int MT = WCO / 16, NT = WCI / 16;
  constexpr int DY_BYTES = BK * BCO * 2, X_BYTES = BK * BCI * 2;
  constexpr int DY_CPR = BCO / 8, X_CPR = BCI / 8;          // 16-B chunks per row
  constexpr int DY_RPP = 256 / DY_CPR, X_RPP = 256 / X_CPR; // rows per pass
  constexpr int DY_IT = BK / DY_RPP, X_IT = BK / X_RPP;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sD = smem;                     // [NBUF][BK][BCO]
  unsigned char* const sX = smem + NBUF * DY_BYTES;   // [NBUF][BK][BCI]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // Block -> (tile, pixel split).  Blocks are dispatched in linear order (x fastest) round-robin over the 8 XCDs, each
  // with its own L2.  All tiles of ONE split read the same pixel rows of x and dy, so they are placed on ONE XCD,
  // back to back: that XCD's L2 fetches the rows once for all of them (with the plain (x = tile, y = split) order the
  // tiles of a split were spread over all 8 XCDs and every L2 fetched the same rows).  SPK_WGRAD_XCD=0: plain order.
  int tile_id = blockIdx.x, split = blockIdx.y;
  if (a.xcd_map) {
    const int T = gridDim.x, S = gridDim.y;
    const int lin = blockIdx.y * T + blockIdx.x;
    const int full = (S >> 3) << 3;                 // splits covered by whole groups of 8
    if (lin < full * T) {
      const int xcd = lin & 7, j = lin >> 3;
      tile_id = j % T;
      split = (j / T) * 8 + xcd;
    } else {
      const int r = lin - full * T;
      tile_id = r % T;
      split = full + r / T;
    }
  }
  const int co0 = (tile_id % co_tiles) * BCO;
  const int kcol0 = (tile_id / co_tiles) * BCI;
  const int p_begin = split * a.pix_per_split;
  const int p_end = min(a.M, p_begin + a.pix_per_split);


  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);

  const int d_ch = tid % DY_CPR, d_row = tid / DY_CPR;
  const int x_ch = tid % X_CPR, x_row = tid / X_CPR;
  // filter tap and input channel of THIS thread's 8-channel column chunk: a column tile may span several taps (Cin = 64
  // with 128-column tiles: half the dy re-reads of the 64-column tiles, whose 16 KB of loads per 8 MFMAs were bound by
  // the L2 -> LDS rate) and may hang over Ktot (those columns load zeros and are not stored)
  int tap_r = 0, tap_s = 0, ci0 = 0;
  bool col_ok = true;
  if (!STEM) {
    const int kc = kcol0 + x_ch * 8;
    col_ok = kc < a.Ktot;
    const int tap = kc / a.Cin;
    ci0 = kc - tap * a.Cin;
    tap_r = tap / a.kw;
    tap_s = tap - tap_r * a.kw;
  }
  const int HoWo = a.Ho * a.Wo;

  u32x4_t rdv[DY_IT], rxv[X_IT];

  // (image, output row, output column) of the X_IT pixels this thread fetches, advanced by BK pixels per step with adds
  // and compares: the two run-time divisions per load (m / HoWo, rem / Wo: ~30 VALU instructions each) used to cost
  // more issue slots per step than the step's 32 MFMAs (SQ counters: 30 % of the wave cycles issuing VALU, 22.8 M VALU
  // against 3.1 M LDS instructions per launch)
  int c_img[X_IT], c_ho[X_IT], c_wo[X_IT];
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int m = p_begin + x_row + i * X_RPP;
    c_img[i] = m / HoWo;
    const int rem = m - c_img[i] * HoWo;
    c_ho[i] = rem / a.Wo;
    c_wo[i] = rem - c_ho[i] * a.Wo;
  }
  const int adv_rows = BK / a.Wo, adv_cols = BK - adv_rows * a.Wo;   // BK pixels = adv_rows rows + adv_cols columns

  auto issue = [&](int p0) {
#pragma unroll
    for (int i = 0; i < DY_IT; ++i) {
      const int m = p0 + d_row + i * DY_RPP;
      const unsigned off = m < p_end ? (unsigned)((m * a.Cout + co0 + d_ch * 8) * 2) : 0x80000000u;
      rdv[i] = __builtin_amdgcn_raw_buffer_load_b128(rd, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int m = p0 + x_row + i * X_RPP;
      const int img = c_img[i], ho = c_ho[i], wo = c_wo[i];
      {   // the next step's pixel: m + BK
        int nw = wo + adv_cols, nh = ho + adv_rows;
        if (nw >= a.Wo) { nw -= a.Wo; nh += 1; }
        int ni = img;
        while (nh >= a.Ho) { nh -= a.Ho; ni += 1; }
        c_img[i] = ni; c_ho[i] = nh; c_wo[i] = nw;
      }
      unsigned off = 0x80000000u;
      if (STEM) {
        // column chunk -> (filter row, pixel pair) of the [8 rows][8 taps][4 ch] image
        const int fr = (kcol0 >> 5) + (x_ch >> 2), qq = x_ch & 3;
        const int hi = ho * 2 - 3 + fr, px = wo * 2 - 4 + 2 * qq;
        if (m < p_end && fr < 7 && (unsigned)hi < (unsigned)a.H && (unsigned)px < (unsigned)a.W)
          off = (unsigned)((((img * a.H + hi) * a.W + px) * 4) * 2);
      } else {
        const int hi = ho * a.stride - a.pad + tap_r, wi = wo * a.stride - a.pad + tap_s;
        if (col_ok && m < p_end && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
          off = (unsigned)((((img * a.H + hi) * a.W + wi) * a.Cin + ci0) * 2);
      }
      rxv[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < DY_IT; ++i)
      *(u32x4_t*)(sD + buf * DY_BYTES + tile_off<BCO>(d_row + i * DY_RPP, d_ch)) = rdv[i];
#pragma unroll
    for (int i = 0; i < X_IT; ++i)
      *(u32x4_t*)(sX + buf * X_BYTES + tile_off<BCI>(x_row + i * X_RPP, x_ch)) = rxv[i];
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nsteps = (p_end - p_begin + BK - 1) / BK;
  if (nsteps > 0) {
    issue(p_begin);
    stash(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = NBUF == 2 ? (st & 1) : 0;
    if (st + 1 < nsteps) issue(p_begin + (st + 1) * BK);
    const unsigned char* pd = sD + buf * DY_BYTES;
    const unsigned char* px = sX + buf * X_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = tr_frag<BCO>(pd, ks * 32, wr * WCO + i * 16, lane);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = tr_frag<BCI>(px, ks * 32, wc * WCI + j * 16, lane);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<DT_BF16>(fa[i], fb[j], acc[i][j]);
    }
    if (NBUF == 1) __syncthreads();  // every wave is done reading the only stage
    if (st + 1 < nsteps) stash(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // C layout: col = lane&15 (ci), row = (lane>>4)*4 + reg (co)
  float* slab = a.slabs + (size_t)split * a.Cout * a.Ktot;
  const int fq = lane >> 4, fr = lane & 15;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wr * WCO + i * 16 + fq * 4 + r;
        const int kc = kcol0 + wc * WCI + j * 16 + fr;
        if (kc < a.Ktot) slab[(size_t)co * a.Ktot + kc] = acc[i][j][r];
      }
}

template <int BCO, int BCI, bool STEM, int NBUF>
int launch_nb(const WgradArgs& a, int splits, hipStream_t s) {
  const int co_tiles = a.Cout / BCO;
  const int k_tiles = (a.Ktot + BCI - 1) / BCI;
  const size_t lds = NBUF * (size_t)64 * (BCO + BCI) * 2;
  auto k = conv_wgrad_kernel<BCO, BCI, STEM, NBUF>;
  static std::atomic<unsigned long long> attr;
  (void)spk_lds_limit_once(attr, (const void*)k, (int)lds);
  hipLaunchKernelGGL(k, dim3(co_tiles * k_tiles, splits), dim3(256), lds, s, a, co_tiles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// one- vs two-stage: timed once per problem (both give bit-identical slabs); process-wide table behind a mutex,
// persisted in SPK_TUNE_CACHE like the conv tile choices (conv_igemm.hip)
typedef std::tuple<int, int, int, int, int, int, int, int> WgKey;
std::map<WgKey, int> g_wg_tuned;
std::mutex g_wg_mu;
bool g_wg_loaded = false;

void wg_cache_load_locked() {
  if (g_wg_loaded) return;
  g_wg_loaded = true;
  const char* path = getenv("SPK_TUNE_CACHE");
  if (!path || !*path) return;
  FILE* f = fopen(path, "r");
  if (!f) return;
  char line[512];
  while (fgets(line, sizeof line, f)) {
    int v[9];
    if (sscanf(line, "wgrad %d %d %d %d %d %d %d %d %d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7], &v[8]) == 9)
      g_wg_tuned[WgKey(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])] = v[8];
  }
  fclose(f);
}

template <int BCO, int BCI, bool STEM>
int launch(const WgradArgs& a, int splits, hipStream_t s) {
  static const int forced = getenv("SPK_WGRAD_NBUF") ? atoi(getenv("SPK_WGRAD_NBUF")) : 0;
  int nbuf = forced;
  if (nbuf != 1 && nbuf != 2) {
    const WgKey key(a.M, a.Cin, a.Cout, a.kh, a.stride, (int)STEM, splits, BCO * 1000 + BCI);
    bool have = false;
    {
      std::lock_guard<std::mutex> lk(g_wg_mu);
      wg_cache_load_locked();
      auto it = g_wg_tuned.find(key);
      if (it != g_wg_tuned.end()) { nbuf = it->second; have = true; }
    }
    if (!have) {
      hipEvent_t e0, e1;
      int best = 2;
      if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        float tbest = 1e30f;
        for (int nb = 2; nb >= 1; --nb) {
          if (nb == 2 ? launch_nb<BCO, BCI, STEM, 2>(a, splits, s) : launch_nb<BCO, BCI, STEM, 1>(a, splits, s)) continue;
          (void)hipEventRecord(e0, s);
          for (int r = 0; r < 2; ++r)
            nb == 2 ? launch_nb<BCO, BCI, STEM, 2>(a, splits, s) : launch_nb<BCO, BCI, STEM, 1>(a, splits, s);
          (void)hipEventRecord(e1, s);
          float ms = 1e30f;
          if (hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
          if (ms < tbest) { tbest = ms; best = nb; }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (getenv("SPK_TUNE_LOG"))
          fprintf(stderr, "[spk tune] wgrad M%d C%d->%d k%d s%d: %d stage(s) (%.1f us)\n", a.M, a.Cin, a.Cout, a.kh,
                  a.stride, best, tbest * 500.f);
      }
      {
        std::lock_guard<std::mutex> lk(g_wg_mu);
        g_wg_tuned[key] = best;
        const char* path = getenv("SPK_TUNE_CACHE");
        if (path && *path) {
          if (FILE* f = fopen(path, "a")) {
            fprintf(f, "wgrad %d %d %d %d %d %d %d %d %d\n", a.M, a.Cin, a.Cout, a.kh, a.stride, (int)STEM, splits,
                    BCO * 1000 + BCI, best);
            fclose(f);
          }
        }
      }
      nbuf = best;
    }
  }
  return nbuf == 1 ? launch_nb<BCO, BCI, STEM, 1>(a, splits, s) : launch_nb<BCO, BCI, STEM, 2>(a, splits, s);
}

}  // namespace

// Chooses the split count (fills *splits, *pix_per_split); slabs must hold
// splits*Cout*Ktot floats.
void spk_wgrad_plan(int M, int Cout, int Ktot, int* splits, int* pix_per_split) {
  const int bco = Cout % 128 == 0 ? 128 : 64;
  const int bci = 128;
  const int tiles = (Cout / bco) * ((Ktot + bci - 1) / bci);
  // blocks per launch the split count aims at.  384 since round 4 (1024 before): every split writes a full [Cout][Ktot] fp32
  // slab that the ordered reduce reads back - 3.0 GB + 3.0 GB per ResNet-50 step at 1024 - and the kernels run on their
  // own low-priority stream beside the HBM-bound BatchNorm passes; measured step 23.10 (1024) / 22.93 (512) / 22.91 (384) /
  // 22.90 ms (256)
  static const int want = getenv("SPK_WGRAD_BLOCKS") ? atoi(getenv("SPK_WGRAD_BLOCKS")) : 384;
  int sp = (want + tiles - 1) / tiles;
  const int max_sp = (M + 511) / 512;
  if (sp > max_sp) sp = max_sp;
  // slab traffic + the ordered reduce grow with the split count.  384 since the weight gradients run on the second
  // stream: the few-tile problems (the stem: 2 tiles) are the last kernels of the step, alone on the chip, and 96
  // splits left them on 192 blocks (ResNet-50 step 23.91 -> 23.68 ms)
  static const int cap = getenv("SPK_WGRAD_SPLIT_CAP") ? atoi(getenv("SPK_WGRAD_SPLIT_CAP")) : 384;
  if (sp > cap) sp = cap;
  if (sp < 1) sp = 1;
  int pps = ((M + sp - 1) / sp + 63) / 64 * 64;
  sp = (M + pps - 1) / pps;
  *splits = sp;
  *pix_per_split = pps;
}

int spk_wgrad_launch(const bf16_t* x, const bf16_t* dy, float* slabs, int N, int H, int W, int Cin,
                     int Ho, int Wo, int Cout, int k, int stride, int pad, int stem, int splits,
                     int pix_per_split, hipStream_t s) {
  WgradArgs a;
  a.x = x; a.dy = dy; a.slabs = slabs;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.kh = a.kw = k; a.stride = stride; a.pad = pad;
  a.M = N * Ho * Wo;
  a.Ktot = stem ? 256 : k * k * Cin;
  a.pix_per_split = pix_per_split;
  static const int xcd_map = getenv("SPK_WGRAD_XCD") ? atoi(getenv("SPK_WGRAD_XCD")) : 1;
  a.xcd_map = xcd_map;
  a.x_bytes = (unsigned)((size_t)N * H * W * (stem ? 4 : Cin) * 2);
  a.dy_bytes = (unsigned)((size_t)a.M * Cout * 2);
  if (stem) {
    if (Cout % 64) return -2;
    return launch<64, 128, true>(a, splits, s);
  }
  if (Cin % 128 == 0) {
    if (Cout % 128 == 0) return launch<128, 128, false>(a, splits, s);
    return launch<64, 128, false>(a, splits, s);
  }
  if (a.Ktot >= 256) {   // several taps per 128-column tile (the last tile may hang over Ktot)
    if (Cout % 128 == 0) return launch<128, 128, false>(a, splits, s);
    return launch<64, 128, false>(a, splits, s);
  }
  if (Cout % 128 == 0) return launch<128, 64, false>(a, splits, s);
  return launch<64, 64, false>(a, splits, s);
}
