// EfficientNet (torchvision MBConv) pieces of the eval path that are not GEMM-shaped:
// the 3x3/2 RGB stem, depthwise k3/k5 convolutions, squeeze-excitation, and the
// zero-padded packing that lets the implicit-GEMM kernel (64-channel granularity) run
// the 1x1 expand / project / head convolutions of a network whose widths are
// multiples of 8 only (expanded C in {48,144,192,336,672,960,1632}).
// Reference: the torchvision backbone reached through sykepic/train/network.py:48;
// topology restated in oracle/backbones.py (parameter counts of B0-B4 match the
// published ones).  Every activation tensor carries its channels padded to a multiple
// of 64; padded channels are exactly zero everywhere (zero weights, zero BN scale/shift,
// SiLU(0) = 0, SE scale 0), so they never contribute.
// All of these are HBM-bound passes: 16-B per-lane accesses, channels innermost.
#include "dw_util.h"

#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int DT>
__device__ __forceinline__ void unpack8f(const u32x4_t v, float* f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = lo_f32<DT>(v[j]);
    f[2 * j + 1] = hi_f32<DT>(v[j]);
  }
}
template <int DT>
__device__ __forceinline__ u32x4_t pack8f(const float* f) {
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack2<DT>(f[2 * j], f[2 * j + 1]);
  return o;
}
__device__ __forceinline__ float act_f(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return silu_f(v);
  return v;
}

// [cout][taps][cin] fp32 -> [cout_p][taps][cin_p] 16-bit (hi, then lo halves), zero padded
template <int DT>
__global__ void pack_padded_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int taps,
                                   int cin, int cout_p, int cin_p, int splitw) {
  const size_t n = (size_t)cout_p * taps * cin_p;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin_p);
    const size_t r = i / cin_p;
    const int t = (int)(r % taps), co = (int)(r / taps);
    const float v = (co < cout && ci < cin) ? w[((size_t)co * taps + t) * cin + ci] : 0.f;
    const unsigned short hi = to_h16<DT>(v);
    out[i] = hi;
    if (splitw) out[n + i] = to_h16<DT>(v - lo_f32<DT>((unsigned int)hi));
  }
}

// depthwise / stem weights: [c][taps*cin1] fp32 -> [taps*cin1][c_p] fp32 (channel innermost), zero padded
__global__ void pack_tapmajor_kernel(const float* __restrict__ w, float* __restrict__ out, int c, int rows,
                                     int c_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c_p) return;
  const int ch = i % c_p, r = i / c_p;
  out[i] = ch < c ? w[(size_t)ch * rows + r] : 0.f;
}

// 3x3 stride-2 pad-1 stem on the NHWC4 input image (RGB + zero channel): one thread = 8 output channels of
// STEM_PX adjacent output pixels; the 27 x C weights, scales and shifts sit in LDS, staged once per block (STEM_ROWS
// output rows).  Per item: the 3 x 9 8-byte image loads are issued together and unconditionally (coordinates clamped
// into the image, the value zeroed by a select) - with a bounds branch per tap every load waited for the one before
// it -, then 4 x 216 FMAs tap by tap (packed, v_pk_fma_f32) and four 16-B stores.  C (32..48 for B0-B4, a multiple
// of 8) is the stored channel count.  Exact fp32 products of the 16-bit image.
constexpr int STEM_ROWS = 2;                            // output rows per block
constexpr int STEM_PX = 4;                              // adjacent outputs per thread
template <int DT, int GT>                               // GT = C / 8 (4..6: B0-B4) or 0: any C, run-time divisions
__global__ __launch_bounds__(256) void stem3x3_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ bias, bf16_t* __restrict__ y, int n,
                                                      int h, int wid, int wstride, int ho, int wo, int c, int c_p,
                                                      int act) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [27 (tap, channel)][c], scale[c], bias[c]
  {
    // 29 rows of c floats as 16-byte loads, all issued before the first LDS store (c <= 256: at most 8 per thread); as a
    // load-store loop this was a chain of L2 round trips at the start of each of the 14 k blocks of a batch of 256
    constexpr int WV = (29 * 256 / 4 + 255) / 256;
    const int total4 = 29 * c / 4;
    f32x4_t v[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int j = min((int)threadIdx.x + i * 256, total4 - 1) * 4;
      const int k = j / c, col = j - k * c;                  // k = tap*3 + channel, then scale, then shift
      const float* src = k < 27 ? w + (size_t)((k / 3) * 4 + k % 3) * c_p : (k == 27 ? scale : bias);   // [tap][4 ch] x c_p
      v[i] = *(const f32x4_t*)(src + col);
    }
#pragma unroll
    for (int i = 0; i < WV; ++i)
      if ((int)threadIdx.x + i * 256 < total4) *(f32x4_t*)(sw + ((int)threadIdx.x + i * 256) * 4) = v[i];
  }
  __syncthreads();
  // a block owns STEM_ROWS output rows of one image: image and row come from the block index (scalar), the only
  // per-item division is by the compile-time group count - with a flat item index the three run-time divisions per
  // item cost more VALU instructions than the 108 packed FMAs
  const int G = GT ? GT : (c >> 3);
  const int row_blocks = (ho + STEM_ROWS - 1) / STEM_ROWS;
  const int img = blockIdx.x / row_blocks, oy0 = (blockIdx.x - img * row_blocks) * STEM_ROWS;
  // an item = STEM_PX horizontally adjacent outputs x 8 channels: every weight vector read from LDS serves STEM_PX
  // pixels (one pixel per item, the 864 B of weights per item made the kernel LDS-bandwidth-bound at 1.7 TB/s)
  const int gpr = (wo + STEM_PX - 1) / STEM_PX;
  const int row_items = gpr * G, items = min(STEM_ROWS, ho - oy0) * row_items;
#pragma unroll 1
  for (int item = threadIdx.x; item < items; item += 256) {
    const int rsel = item >= row_items ? 1 : 0;          // STEM_ROWS == 2
    const int in_row = item - rsel * row_items;
    const int og = in_row / G, c0 = (in_row - og * G) * 8;
    const int oy = oy0 + rsel, ox0 = og * STEM_PX;
    constexpr int NCOL = 2 * STEM_PX + 1;                // input columns 2*ox0 - 1 ... 2*ox0 + 2*STEM_PX - 1
    uint2 px[3][NCOL];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = 2 * oy - 1 + r;
      const int iyc = min(max(iy, 0), h - 1);
      const bf16_t* rowp = x + ((size_t)img * h + iyc) * wstride * 4;
#pragma unroll
      for (int q = 0; q < NCOL; ++q) {
        const int ix = 2 * ox0 - 1 + q;
        const int ixc = min(max(ix, 0), wid - 1);
        px[r][q] = *(const uint2*)(rowp + (size_t)ixc * 4);
        if ((unsigned)iy >= (unsigned)h || (unsigned)ix >= (unsigned)wid) px[r][q] = uint2{0u, 0u};
      }
    }
    f32x2_t acc2[STEM_PX][4];                            // packed pairs: v_pk_fma_f32, 2 FMAs per lane and instruction
#pragma unroll
    for (int u = 0; u < STEM_PX; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[u][j] = f32x2_t{0.f, 0.f};
    int woff = c0;                                     // LDS float offset of this tap's weights
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // ties the offset to the last tap's results: this tap's LDS reads cannot move above the last tap's FMAs, nor those
      // FMAs below this point (unfenced, hipcc 7.2 front-loads all 54 LDS reads: 216 VGPRs of weights)
      static_assert(STEM_PX == 4, "the fence lists the accumulators");
      asm volatile("" : "+v"(woff), "+v"(acc2[0][0]), "+v"(acc2[0][1]), "+v"(acc2[0][2]), "+v"(acc2[0][3]), "+v"(acc2[1][0]),
                   "+v"(acc2[1][1]), "+v"(acc2[1][2]), "+v"(acc2[1][3]), "+v"(acc2[2][0]), "+v"(acc2[2][1]), "+v"(acc2[2][2]),
                   "+v"(acc2[2][3]), "+v"(acc2[3][0]), "+v"(acc2[3][1]), "+v"(acc2[3][2]), "+v"(acc2[3][3]));
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float* wp = sw + woff + (t * 3 + ch) * c;
        const f32x4_t w0 = *(const f32x4_t*)wp, w1 = *(const f32x4_t*)(wp + 4);
#pragma unroll
        for (int u = 0; u < STEM_PX; ++u) {
          const uint2 p = px[t / 3][2 * u + t % 3];
          const float xv = ch == 0 ? lo_f32<DT>(p.x) : (ch == 1 ? hi_f32<DT>(p.x) : lo_f32<DT>(p.y));
          const f32x2_t xx = {xv, xv};
          acc2[u][0] = __builtin_elementwise_fma(xx, f32x2_t{w0[0], w0[1]}, acc2[u][0]);
          acc2[u][1] = __builtin_elementwise_fma(xx, f32x2_t{w0[2], w0[3]}, acc2[u][1]);
          acc2[u][2] = __builtin_elementwise_fma(xx, f32x2_t{w1[0], w1[1]}, acc2[u][2]);
          acc2[u][3] = __builtin_elementwise_fma(xx, f32x2_t{w1[2], w1[3]}, acc2[u][3]);
        }
      }
    }
    const f32x4_t s0 = *(const f32x4_t*)(sw + 27 * c + c0), s1 = *(const f32x4_t*)(sw + 27 * c + c0 + 4);
    const f32x4_t b0 = *(const f32x4_t*)(sw + 28 * c + c0), b1 = *(const f32x4_t*)(sw + 28 * c + c0 + 4);
#pragma unroll
    for (int u = 0; u < STEM_PX; ++u) {
      if (ox0 + u >= wo) continue;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] = acc2[u][j][0]; acc[2 * j + 1] = acc2[u][j][1]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] = acc[j] * s0[j] + b0[j]; acc[4 + j] = acc[4 + j] * s1[j] + b1[j]; }
      if (act == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = silu_f(acc[j]);
      } else if (act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
      }
      *(u32x4_t*)(y + (((size_t)img * ho + oy) * wo + ox0 + u) * c + c0) = pack8f<DT>(acc);
    }
  }
}

// depthwise KxK conv + folded BN + activation.  One block = one chunk of output pixels of ONE image x one tile
// of up to 256 channels: the tile's K*K x 256 weights, scales and shifts sit in LDS.  A thread owns 8 channels
// and PX horizontally adjacent output pixels: each input value is loaded once for all the outputs whose window
// covers it ((PX-1)*S+K columns per row instead of PX*K: the pass is bound by L1 requests, not HBM).  Lanes of
// a wave cover consecutive channel groups (contiguous 16-B accesses).  The per-channel sums of the block's
// outputs (fp32, before the 16-bit rounding) are reduced over the block in fixed order and written as the
// squeeze-excitation pool partial [img][chunk][c_p].
template <int DT, int K, int S, int PX>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                     float* __restrict__ partial, int h, int wid, int c_p, int ho,
                                                     int wo, int act, int chunks) {
  constexpr int PAD = (K - 1) / 2;
  constexpr int COLS = (PX - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [K*K + 2][tc]; reused for the pool reduce
  const int img = blockIdx.y / chunks, chunk = blockIdx.y % chunks;
  const int c8 = c_p >> 3;
  const int cg0 = blockIdx.x * 32;
  const int ncg = min(32, c8 - cg0), tc = ncg * 8;
  dwu::stage_dw_weights<K, 8>(sm, w, scale, bias, c_p, cg0 * 8, ncg, ncg, 1.f, threadIdx.x);   // layout: dw_util.h
  __syncthreads();
  const int rows = 256 / ncg;  // pixel-group lanes
  const int cg = threadIdx.x % ncg, prow = threadIdx.x / ncg;
  const int gpr = (wo + PX - 1) / PX, ngroups = ho * gpr;  // groups of PX outputs along W
  const int per = (ngroups + chunks - 1) / chunks;
  const int g0 = chunk * per, g1 = min(ngroups, g0 + per);
  float pool[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (prow < rows) {
    const float* wl = sm + cg * 4;   // + (row * 2 + half) * ncg * 4
    const bf16_t* xi = x + (size_t)img * h * wid * c_p + (cg0 + cg) * 8;
    bf16_t* yi = y + (size_t)img * ho * wo * c_p + (cg0 + cg) * 8;
    for (int g = g0 + prow; g < g1; g += rows) {
      const int oy = g / gpr, ox0 = (g - oy * gpr) * PX;
      float acc[PX][8];
#pragma unroll
      for (int u = 0; u < PX; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[u][j] = 0.f;
      // k3: rows unrolled (all 18 loads in flight, 120-141 VGPRs); k5: a real loop (unrolled, 40 loads = 160 VGPRs of
      // raw pixels are live at once and only two waves fit a SIMD)
#pragma unroll(K == 3 ? 3 : 1)
      for (int r = 0; r < K; ++r) {
        const int iy = oy * S - PAD + r;
        if ((unsigned)iy >= (unsigned)h) continue;
        float wrow[K][8];
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const f32x4_t w0 = *(const f32x4_t*)(wl + ((r * K + q) * 2) * ncg * 4), w1 = *(const f32x4_t*)(wl + ((r * K + q) * 2 + 1) * ncg * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { wrow[q][j] = w0[j]; wrow[q][4 + j] = w1[j]; }
        }
        // all COLS loads of the row are issued before the first use: unconditional (column clamped into the
        // image, value zeroed by a select), so no branch separates them and their latencies overlap
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
          float xv[8];
          unpack8f<DT>(raw[col], xv);
#pragma unroll
          for (int u = 0; u < PX; ++u) {
            const int q = col - u * S;  // tap of output u that this column feeds
            if (q >= 0 && q < K) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[u][j] += xv[j] * wrow[q][j];
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < PX; ++u) {
        if (ox0 + u >= wo) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[u][j] = acc[u][j] * wl[((K * K) * 2 + (j >> 2)) * ncg * 4 + (j & 3)] + wl[((K * K + 1) * 2 + (j >> 2)) * ncg * 4 + (j & 3)];
        if (act == 2) {   // uniform branch: one activation's instructions, not both + selects
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[u][j] = silu_f(acc[u][j]);
        } else if (act == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[u][j] = fmaxf(acc[u][j], 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) pool[j] += acc[u][j];
        *(u32x4_t*)(yi + ((size_t)oy * wo + ox0 + u) * c_p) = pack8f<DT>(acc[u]);
      }
    }
  }
  if (!partial) return;
  __syncthreads();  // weights are dead: the buffer becomes [rows][tc] pool partials
  if (prow < rows) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sm[prow * tc + cg * 8 + j] = pool[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < tc; c += 256) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += sm[r * tc + c];
    partial[((size_t)img * chunks + chunk) * c_p + cg0 * 8 + c] = t;
  }
}

// excitation, part 1: hid = silu(W1 avg + b1).  Block (IPB images, 4 hidden units): all 256 threads stride the channels
// (the pooled mean is summed from the depthwise kernel's chunk partials on the fly); every weight that is loaded serves
// IPB images - with one image per block the two excitation kernels re-read the fc weights once per image through L2
// (c 2688: 2 x 300 MB per layer) and ran 30-100 us; fixed-order shuffle + LDS reduction.  hid[img][sq].
constexpr int SE_IPB = 8;
// CH: the number of pool-partial rows per image when it is 1 ... 4 or 8 (what the depthwise kernels write at batch >= 32),
// else 0 = run-time count.  With a compile-time count the IPB x CH partial loads of a channel are all in flight
// together; the run-time loop issued them one after the other, a chain of up to 32 L2 round trips per block (the kernel
// took ~20 us whatever the layer size: 0.63 ms per EfficientNet-B4 forward).
template <int CH>
__global__ __launch_bounds__(256) void se_fc1_kernel(const float* __restrict__ partial, int chunks_rt, float inv_hw,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     float* __restrict__ hid, int n, int c, int c_p, int sq) {
  const int chunks = CH ? CH : chunks_rt;
  __shared__ float red[4][SE_IPB][4];
  const int img0 = blockIdx.x * SE_IPB, j0 = blockIdx.y * 4;
  const int nim = min(SE_IPB, n - img0);
  float t[SE_IPB][4];
#pragma unroll
  for (int im = 0; im < SE_IPB; ++im)
#pragma unroll
    for (int q = 0; q < 4; ++q) t[im][q] = 0.f;
  for (int i = threadIdx.x; i < c; i += 256) {
    float wv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wv[q] = j0 + q < sq ? w1[(size_t)(j0 + q) * c + i] : 0.f;
    float av[SE_IPB];
    if (CH) {
      float pv[SE_IPB][CH ? CH : 1];
#pragma unroll
      for (int im = 0; im < SE_IPB; ++im)
#pragma unroll
        for (int k = 0; k < CH; ++k)   // image index clamped: every load is issued, the surplus is dropped below
          pv[im][k] = partial[((size_t)(img0 + min(im, nim - 1)) * CH + k) * c_p + i];
#pragma unroll
      for (int im = 0; im < SE_IPB; ++im) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) a += pv[im][k];   // same order as the run-time loop
        av[im] = im < nim ? a : 0.f;
      }
    } else {
      // run-time count (a half batch of 64 images on the 112^2 / 56^2 stages: 5 ... 16 rows): four rows x IPB images in
      // flight per step, added in row order as before.  One row at a time this path was a chain of `chunks` L2 round trips:
      // 55-67 us per launch, six launches per EfficientNet-B4 half batch (0.37 of its 4.5 ms).
#pragma unroll
      for (int im = 0; im < SE_IPB; ++im) av[im] = 0.f;
      for (int k0 = 0; k0 < chunks; k0 += 4) {
        float pv[SE_IPB][4];
#pragma unroll
        for (int im = 0; im < SE_IPB; ++im)
#pragma unroll
          for (int u = 0; u < 4; ++u)   // indices clamped: every load is issued, the surplus is dropped below
            pv[im][u] = partial[((size_t)(img0 + min(im, nim - 1)) * chunks + min(k0 + u, chunks - 1)) * c_p + i];
#pragma unroll
        for (int im = 0; im < SE_IPB; ++im)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < chunks && im < nim) av[im] += pv[im][u];
      }
    }
#pragma unroll
    for (int im = 0; im < SE_IPB; ++im) {
      const float a = av[im] * inv_hw;
#pragma unroll
      for (int q = 0; q < 4; ++q) t[im][q] += wv[q] * a;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int im = 0; im < SE_IPB; ++im)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = t[im][q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) red[wave][im][q] = v;
    }
  __syncthreads();
  if (threadIdx.x < SE_IPB * 4) {
    const int im = threadIdx.x >> 2, q = threadIdx.x & 3;
    if (im < nim && j0 + q < sq) {
      const float v = red[0][im][q] + red[1][im][q] + red[2][im][q] + red[3][im][q] + b1[j0 + q];
      hid[(size_t)(img0 + im) * sq + j0 + q] = silu_f(v);
    }
  }
}

// excitation, part 2: s = sigmoid(W2 hid + b2); thread = channel, fc2 weights transposed ([sq][c_p]: coalesced), each
// weight serves the block's IPB images; scale[img][c_p] (0 on padding)
__global__ __launch_bounds__(256) void se_fc2_kernel(const float* __restrict__ hid, const float* __restrict__ w2t,
                                                     const float* __restrict__ b2, float* __restrict__ scale, int n,
                                                     int c, int c_p, int sq) {
  extern __shared__ float sh[];  // hid[sq][IPB]
  const int img0 = blockIdx.x * SE_IPB;
  const int nim = min(SE_IPB, n - img0);
  for (int k0 = 0; k0 < sq * SE_IPB; k0 += 4 * 256) {   // four loads in flight per thread, then the stores
    float hv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(k0 + u * 256 + (int)threadIdx.x, sq * SE_IPB - 1);
      const int j = k / SE_IPB, im = k - j * SE_IPB;
      hv[u] = hid[(size_t)(img0 + min(im, nim - 1)) * sq + j];
      if (im >= nim) hv[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * 256 + (int)threadIdx.x;
      if (k < sq * SE_IPB) sh[k] = hv[u];
    }
  }
  __syncthreads();
  const int i = blockIdx.y * 256 + threadIdx.x;
  if (i >= c_p) return;
  float t[SE_IPB];
  const float b = i < c ? b2[i] : 0.f;
#pragma unroll
  for (int im = 0; im < SE_IPB; ++im) t[im] = b;
  if (i < c) {
#pragma unroll 4
    for (int j = 0; j < sq; ++j) {
      const float w = w2t[(size_t)j * c_p + i];
      const f32x4_t h0 = *(const f32x4_t*)(sh + j * SE_IPB), h1 = *(const f32x4_t*)(sh + j * SE_IPB + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { t[q] += w * h0[q]; t[4 + q] += w * h1[q]; }
    }
  }
#pragma unroll
  for (int im = 0; im < SE_IPB; ++im)
    if (im < nim) scale[(size_t)(img0 + im) * c_p + i] = i < c ? sigmoid_f(t[im]) : 0.f;
}

template <int DT>
__global__ void se_scale_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                bf16_t* __restrict__ y, int hw, int c_p) {
  // one block row (blockIdx.y) per image: no 64-bit divisions in the loop
  const unsigned c8 = c_p >> 3, per_img = (unsigned)hw * c8;
  const size_t base = (size_t)blockIdx.y * per_img;
  for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < per_img; k += gridDim.x * blockDim.x) {
    const unsigned cg = k % c8;
    const size_t i = base + k;
    float v[8];
    unpack8f<DT>(*(const u32x4_t*)(x + i * 8), v);
    const float* s = scale + (size_t)blockIdx.y * c_p + cg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= s[j];
    *(u32x4_t*)(y + i * 8) = pack8f<DT>(v);
  }
}

inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  if (g > 256 * 8 * 4) g = 256 * 8 * 4;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

#define DT_DISPATCH(dt, CALL_BF16, CALL_F16) \
  do { if ((dt) == DT_F16) { CALL_F16; } else { CALL_BF16; } } while (0)

int spk_launch_pack_padded(const float* w, bf16_t* out, int cout, int taps, int cin, int cout_p, int cin_p, int dt,
                           int splitw, hipStream_t s) {
  const int g = grid_for((size_t)cout_p * taps * cin_p, 256);
  DT_DISPATCH(dt,
              hipLaunchKernelGGL(pack_padded_kernel<DT_BF16>, dim3(g), dim3(256), 0, s, w, out, cout, taps, cin, cout_p, cin_p, splitw),
              hipLaunchKernelGGL(pack_padded_kernel<DT_F16>, dim3(g), dim3(256), 0, s, w, out, cout, taps, cin, cout_p, cin_p, splitw));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_pack_tapmajor(const float* w, float* out, int c, int rows, int c_p, hipStream_t s) {
  hipLaunchKernelGGL(pack_tapmajor_kernel, dim3((rows * c_p + 255) / 256), dim3(256), 0, s, w, out, c, rows, c_p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int spk_launch_stem3x3(const bf16_t* x, const float* w, const float* scale, const float* bias, bf16_t* y, int n, int h,
                       int wid, int wstride, int ho, int wo, int c, int c_p, int act, int dt, hipStream_t s) {
  const size_t total = (size_t)n * ho * wo * (c / 8);
  if (c % 8 || c > 256 || dt != DT_F16 || total >= ((size_t)1 << 31)) return -2;
  const dim3 grid((unsigned)n * ((ho + STEM_ROWS - 1) / STEM_ROWS));
#define SPK_STEM(GT) \
  hipLaunchKernelGGL((stem3x3_kernel<DT_F16, GT>), grid, dim3(256), (size_t)29 * c * 4, s, x, w, scale, bias, y, n, h, wid, \
                     wstride, ho, wo, c, c_p, act)
  switch (c / 8) {
    case 4: SPK_STEM(4); break;
    case 5: SPK_STEM(5); break;
    case 6: SPK_STEM(6); break;
    default: SPK_STEM(0);
  }
#undef SPK_STEM
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// chunks of one image's work items (same value at planning and at launch): enough blocks to fill the chip
int spk_dw_chunks(int n, int hw, int c_p) {
  const int ctiles = (c_p / 8 + 31) / 32;
  int chunks = 1;
  while (chunks < 64 && (long)n * ctiles * chunks < 1024 && hw / (chunks * 2) >= 16) chunks *= 2;
  return chunks;
}

int spk_launch_dwconv(const bf16_t* x, const float* w, const float* scale, const float* bias, bf16_t* y, float* partial,
                      int n, int h, int wid, int c_p, int ho, int wo, int k, int stride, int act, int dt, hipStream_t s) {
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2) || (dt != DT_F16 && dt != DT_BF16)) return -2;
  constexpr int PX = 4;
  const int groups = ho * ((wo + PX - 1) / PX);
  const int chunks = spk_dw_chunks(n, groups, c_p);
  const int c8 = c_p / 8, ctiles = (c8 + 31) / 32;
  const int tc = (c8 < 32 ? c8 : 32) * 8;
  const size_t lds = (size_t)std::max((k * k + 2) * tc, 256 / (tc / 8) * tc) * 4;
  const dim3 grid(ctiles, n * chunks);
  // bf16: the training step's depthwise forward (and its stride-1 data gradient, a correlation with the flipped window)
#define SPK_DW(D, K, S)                                                                                            \
  hipLaunchKernelGGL((dwconv_kernel<D, K, S, PX>), grid, dim3(256), lds, s, x, w, scale, bias, y, partial, h, wid, \
                     c_p, ho, wo, act, chunks)
#define SPK_DW_ALL(D)                         \
  if (k == 3 && stride == 1) SPK_DW(D, 3, 1); \
  else if (k == 3) SPK_DW(D, 3, 2);           \
  else if (stride == 1) SPK_DW(D, 5, 1);      \
  else SPK_DW(D, 5, 2)
  if (dt == DT_F16) { SPK_DW_ALL(DT_F16); } else { SPK_DW_ALL(DT_BF16); }
#undef SPK_DW_ALL
#undef SPK_DW
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// squeeze-excitation gate on a depthwise output whose pool partials [n][chunks][c_p] the depthwise kernel wrote
int spk_launch_se(const bf16_t* x, bf16_t* y, const float* partial, int chunks, float* scale, const float* w1,
                  const float* b1, const float* w2t, const float* b2, int n, int hw, int c, int c_p, int sq, int dt,
                  hipStream_t s) {
  if (dt != DT_F16) return -2;
  float* hid = scale + (size_t)n * c_p;  // scratch behind the scales: [n][sq]
  const int ig = (n + SE_IPB - 1) / SE_IPB;
#define SPK_FC1(CH) \
  hipLaunchKernelGGL(se_fc1_kernel<CH>, dim3(ig, (sq + 3) / 4), dim3(256), 0, s, partial, chunks, 1.0f / (float)hw, w1, b1, \
                     hid, n, c, c_p, sq)
  switch (chunks) {
    case 1: SPK_FC1(1); break;
    case 2: SPK_FC1(2); break;
    case 3: SPK_FC1(3); break;
    case 4: SPK_FC1(4); break;
    case 8: SPK_FC1(8); break;
    default: SPK_FC1(0);
  }
#undef SPK_FC1
  hipLaunchKernelGGL(se_fc2_kernel, dim3(ig, (c_p + 255) / 256), dim3(256), (size_t)sq * SE_IPB * 4, s, hid, w2t, b2, scale,
                     n, c, c_p, sq);
  if (!y) return hipGetLastError() == hipSuccess ? 0 : -1;   // gates only
  const size_t per_img = (size_t)hw * (c_p / 8);
  if (per_img >= ((size_t)1 << 31)) return -2;
  int gx = (int)((per_img + 255) / 256);
  const int want = std::max(1, 4096 / n);  // about 16 blocks per CU over the whole batch
  if (gx > want) gx = want;
  hipLaunchKernelGGL(se_scale_kernel<DT_F16>, dim3(gx, n), dim3(256), 0, s, x, scale, y, hw, c_p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
