// 7x7 stride-2 stem convolution (Cin <= 4 -> 64) for gfx950: persistent blocks,
// weights resident in LDS, input patch staged once per 16x16 output tile.
//
// The first layer of every torchvision ResNet the reference instantiates
// (`base.0`, reached through `net(x)`: sykepic/compute/probability.py:189,
// sykepic/train/train.py:240).  As an implicit GEMM it has K = 147 (padded to
// 256) and every input pixel is fetched by ~12 output positions, so the generic
// kernel is bound by L1/L2 gather traffic (measured 1.9 TB/s algorithmic, 2.6x
// off the HBM floor).  Here each input pixel crosses L2 once:
//   * the block keeps all weights ([64][8 rows][8 taps][4 ch], 32 KB; 64 KB with
//     the hi/lo split) in LDS for its whole life and walks a list of output tiles;
//   * per tile the 37 x 40-pixel input patch (11.8 KB, NHWC4) is loaded with
//     coalesced 16-B buffer loads (borders = buffer range check -> zeros),
//     double buffered against the MFMAs of the previous tile;
//   * MFMA A fragments are read STRAIGHT from the patch: K step r is filter row
//     r, the lane's 16-B chunk is the pixel pair (2*ox-4+2q, +1) of input row
//     2*oy-3+r — consecutive lanes are 16 B apart, conflict-free;
//   * B fragments come from the resident weights (512-B rows, 16-B chunk index
//     XOR-swizzled with the cout so 16 couts hit 16 slots).
// Epilogue as in conv_igemm.hip: per-wave LDS staging, BN scale/shift (eval) or
// raw + per-channel sum / sum-of-squares partials (train), ReLU, 16-bit rows.
//
// POOL variant (eval): the 3x3 stride-2 pad-1 max-pool that follows the stem (`base.3`) is applied to the tile in
// LDS and only the pooled tensor is written: the 411 MB stem output of a batch of 256 neither goes to HBM nor comes
// back (the separate pass cost 127-132 us, the stem 210).  A 16x16 stem tile whose origin sits one row and column
// before a multiple of 14 holds every input of 7x7 pooled outputs, so the tiles advance by 14 stem pixels (8x8
// tiles per 112^2 image instead of 7x7: 31 % more MFMA work on the cheapest layer of the network).  The rounded
// 16-bit values are pooled, exactly what the separate kernel reads, so the result is bit-identical to the
// two-kernel path (max commutes with the monotonic rounding anyway).
#include "spk_common.h"
#include <cstdlib>

namespace {

constexpr int TILE = 16;                 // output tile edge (pixels)
constexpr int PH = 2 * TILE + 5;         // 37 input rows
constexpr int PWP = TILE + 4;            // 20 pixel pairs per patch row (40 pixels)
constexpr int PATCH_BYTES = PH * PWP * 16;
constexpr int W_ROW_BYTES = 512;         // 256 k-elements per cout

constexpr int POOL_STEP = 14;            // POOL: stem pixels a tile advances by (7 pooled outputs)
constexpr int POOL_OUT = 7;

template <int DT, int SPLITW, int POOL>
__global__ __launch_bounds__(256) void conv_stem_kernel(ConvArgs a, int tiles_x, int tiles_y, int n_tiles) {
  constexpr int NW = SPLITW ? 2 : 1;
  // POOL: a block covers 32 of the 64 couts (pair cb = blockIdx.x & 1) and the two blocks of a pair walk the same
  // tiles: half the weight panel per block (32 KB with hi + lo) brings the block to 65 KB of LDS, i.e. TWO blocks per
  // CU - the K loop of one overlaps the epilogue / pool / patch-staging phases of the other (those were 2/3 of the
  // kernel's time with a single 4-wave block per CU).
  constexpr int NJ = POOL ? 2 : 4;                         // 16-cout MFMA tiles per block
  constexpr int WROWS = 16 * NJ;                           // weight rows in LDS
  constexpr int W_BYTES = WROWS * W_ROW_BYTES;
  const int cb = POOL ? (int)(blockIdx.x & 1) : 0;
  constexpr int EPI_LD = 68;
  constexpr int TSTEP = POOL ? POOL_STEP : TILE;           // stem pixels between tile origins
  constexpr int TORG = POOL ? 1 : 0;                       // ... and the origin's offset before the multiple
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sW = smem;                          // [NW][64][512 B]
  unsigned char* const sP = smem + NW * W_BYTES;           // [2][PATCH_BYTES]
  float* const sE = (float*)(sP + 2 * PATCH_BYTES);        // [4 waves][16][EPI_LD]   (not POOL)
  unsigned char* const sH = sP + 2 * PATCH_BYTES;          // POOL: [16 rows][7 pooled columns] x HSTRIDE bytes
  constexpr int HSTRIDE = 80;                              // 64 B (the block's 32 channels) + 16: conflict-free 16-byte writes

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);

  // ---- weights -> LDS, once ----
  for (int c = tid; c < NW * WROWS * 32; c += 256) {
    const int ch = c & 31, row = (c >> 5) % WROWS, half = c / (WROWS * 32);
    // POOL: swapped MFMA operand roles (weights = A, rows = couts).  LDS row q of the block holds cout 32 cb +
    // 8((q&15)>>2) + 4(q>>4) + (q&3), so that a lane's accumulators of the two tiles are couts 32 cb + 8(lane>>4) .. +7 of
    // ONE pixel: the epilogue runs from registers (conv_pw.hip's arrangement), no transposition through LDS.
    const int srow = POOL ? 32 * cb + 8 * ((row & 15) >> 2) + 4 * (row >> 4) + (row & 3) : row;
    const u32x4_t v = *(const u32x4_t*)(a.w + ((size_t)(half * 64 + srow) * 256 + ch * 8));
    *(u32x4_t*)(sW + half * W_BYTES + row * W_ROW_BYTES + ((ch ^ (row & 15)) << 4)) = v;
  }

  // ---- per-thread patch chunks (PH*PWP = 740 chunks of 16 B, 3 per thread) ----
  constexpr int P_IT = (PH * PWP + 255) / 256;
  u32x4_t rp[P_IT];
  auto issue_patch = [&](int tile) {
    const int img = tile / (tiles_x * tiles_y);
    const int rem = tile - img * tiles_x * tiles_y;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int iy0 = (ty * TSTEP - TORG) * 2 - 3, ix0 = (tx * TSTEP - TORG) * 2 - 4;
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int c = tid + i * 256;
      const int pr = c / PWP, pp = c - pr * PWP;
      const int iy = iy0 + pr, ix = ix0 + 2 * pp;
      const bool ok = c < PH * PWP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned off = ok ? (unsigned)((((img * a.H + iy) * a.W) + ix) * 8) : 0x80000000u;
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
  };
  auto store_patch = [&](int buf) {
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int c = tid + i * 256;
      if (c < PH * PWP) *(u32x4_t*)(sP + buf * PATCH_BYTES + c * 16) = rp[i];
    }
  };

  // epilogue constants (8 lanes per output pixel, 8 channels each)
  const int ecol = (lane & 7) * 8, erow = lane >> 3;
  float sc[8], bi[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = a.scale ? a.scale[ecol + j] : 1.f;
    bi[j] = a.bias ? a.bias[ecol + j] : 0.f;
    s1[j] = s2[j] = 0.f;
  }
  float* const epi = sE + wave * (16 * EPI_LD);
  float psc[8], pbi[8];                                    // POOL: couts 32 cb + 8fq + 0..7
  if (POOL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      psc[j] = a.scale ? a.scale[32 * cb + 8 * fq + j] : 1.f;
      pbi[j] = a.bias ? a.bias[32 * cb + 8 * fq + j] : 0.f;
    }
  }

  const int tstep = POOL ? (int)(gridDim.x >> 1) : (int)gridDim.x;
  int tile = POOL ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  int buf = 0;
  if (tile < n_tiles) {
    issue_patch(tile);
    store_patch(0);
  }
  __syncthreads();
  for (; tile < n_tiles; tile += tstep) {
    const int next = tile + tstep;
    if (next < n_tiles) issue_patch(next);

    const int img = tile / (tiles_x * tiles_y);
    const int rem = tile - img * tiles_x * tiles_y;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const unsigned char* patch = sP + buf * PATCH_BYTES;

    // wave w: output rows 4w..4w+3 of the tile (one 16-pixel MFMA row tile each), all 64 couts
    f32x4_t acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      u32x4_t fa[4], fb[NJ], fl[SPLITW ? NJ : 1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = wave * 4 + i;
        fa[i] = *(const u32x4_t*)(patch + ((2 * oy + r) * PWP + frow + fq) * 16);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = j * 16 + frow;
        fb[j] = *(const u32x4_t*)(sW + n * W_ROW_BYTES + (((r * 4 + fq) ^ (n & 15)) << 4));
        if (SPLITW) fl[j] = *(const u32x4_t*)(sW + W_BYTES + n * W_ROW_BYTES + (((r * 4 + fq) ^ (n & 15)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (POOL) {   // D rows = couts, columns = pixels
            acc[i][j] = mfma16<DT>(fb[j], fa[i], acc[i][j]);
            if (SPLITW) acc[i][j] = mfma16<DT>(fl[j], fa[i], acc[i][j]);
          } else {
            acc[i][j] = mfma16<DT>(fa[i], fb[j], acc[i][j]);
            if (SPLITW) acc[i][j] = mfma16<DT>(fa[i], fl[j], acc[i][j]);
          }
        }
    }

    if (POOL) {
      // ---- register epilogue + the horizontal half of the 3x3/2 max-pool ----
      // lane (pixel column frow, cout group fq) holds 8 consecutive couts of tile row wave*4 + i per pair P: BN, ReLU,
      // 16-bit rounding; the max over columns 2c, 2c+1, 2c+2 comes from the two right-hand neighbour lanes (DPP row
      // shifts; zeros flow in past the tile edge and for pixels outside the stem output - every real value is >= +0
      // after the ReLU, so zeros never win).  Lanes with an even column < 14 store the 7 pooled columns of the row.
      typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
      const int ox = tx * TSTEP - TORG + frow;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = ty * TSTEP - TORG + wave * 4 + i;
        const bool inside = (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
        {
          u32x4_t ov = {0u, 0u, 0u, 0u};
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = fmaxf(acc[i][0][r] * psc[r] + pbi[r], 0.f);
            v[4 + r] = fmaxf(acc[i][NJ - 1][r] * psc[4 + r] + pbi[4 + r], 0.f);
          }
          if (inside) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = pack2<DT>(v[2 * j], v[2 * j + 1]) & 0x7fff7fffu;   // -0 -> +0
          }
          u32x4_t hm;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // column + 1 from the right-hand neighbour lane, column + 2 as the neighbour's neighbour.  `own` is pinned
            // in its register across the first DPP move: hipcc 7.2 otherwise allocates the move's destination ON its
            // source (v_mov_b32_dpp v4, v4 row_shl:2) although the source value is still needed for the max - the lane's
            // own column dropped out of 3 of the 4 dwords (measured: pooled column 0, whose own column is padding,
            // right; 40 % of the others wrong).
            unsigned own = ov[j];
            asm volatile("" : "+v"(own));
            unsigned n1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0x101, 0xf, 0xf, true);
            asm volatile("" : "+v"(n1), "+v"(own));
            const u16x2_t t = __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, own), __builtin_bit_cast(u16x2_t, n1));
            unsigned n2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)n1, 0x101, 0xf, 0xf, true);
            asm volatile("" : "+v"(n2));
            const u16x2_t m = __builtin_elementwise_max(t, __builtin_bit_cast(u16x2_t, n2));
            hm[j] = __builtin_bit_cast(unsigned, m);
          }
          if (!(frow & 1) && frow < 2 * POOL_OUT)
            *(u32x4_t*)(sH + ((wave * 4 + i) * POOL_OUT + (frow >> 1)) * HSTRIDE + 8 * fq * 2) = hm;
        }
      }
    } else
    // ---- epilogue: one output row (16 pixels x 64 couts) per i ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) epi[(fq * 4 + rr) * EPI_LD + j * 16 + frow] = acc[i][j][rr];
      __builtin_amdgcn_wave_barrier();
      const int oy = ty * TSTEP - TORG + wave * 4 + i;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int px = erow + p * 8;
        const int ox = tx * TSTEP - TORG + px;
        const f32x4_t v0 = *(const f32x4_t*)(epi + px * EPI_LD + ecol);
        const f32x4_t v1 = *(const f32x4_t*)(epi + px * EPI_LD + ecol + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const bool inside = (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
        if (POOL) {
          // (the POOL variant takes the register epilogue above)
        } else if (inside) {
          if (a.stats) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = v[j] * sc[j] + bi[j];
            if (a.relu) v[j] = fmaxf(v[j], 0.f);
          }
          u32x4_t ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) ov[j] = pack2<DT>(v[2 * j], v[2 * j + 1]);
          *(u32x4_t*)(a.y + (((size_t)img * a.Ho + oy) * a.Wo + ox) * 64 + ecol) = ov;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }

    if (next < n_tiles) store_patch(buf ^ 1);
    __syncthreads();
    if (POOL) {
      // 7 x 7 pooled pixels x 8 channel chunks of 16 B: the window of pooled (py, px) is the tile's rows 2py .. 2py+2 and
      // columns 2px .. 2px+2.  Non-negative 16-bit floats order like unsigned integers (fp16 and bf16 alike).
      for (int it = tid; it < POOL_OUT * POOL_OUT * 4; it += 256) {
        const int chunk = it & 3, pp = it >> 2;
        const int ppy = pp / POOL_OUT, ppx = pp - ppy * POOL_OUT;
        const int py = ty * POOL_OUT + ppy, pxg = tx * POOL_OUT + ppx;
        if (py >= a.pool_ho || pxg >= a.pool_wo) continue;
        u32x4_t mx = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const u32x4_t t = *(const u32x4_t*)(sH + ((2 * ppy + r) * POOL_OUT + ppx) * HSTRIDE + chunk * 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned int av = mx[j], bv = t[j];
            const unsigned int lo = max(av & 0xffffu, bv & 0xffffu), hi = max(av >> 16, bv >> 16);
            mx[j] = lo | (hi << 16);
          }
        }
        *(u32x4_t*)(a.pool_y + (((size_t)img * a.pool_ho + py) * a.pool_wo + pxg) * 64 + 32 * cb + chunk * 8) = mx;
      }
      __syncthreads();   // the next tile's epilogue overwrites sH
    }
    buf ^= 1;
  }

  if (a.stats) {
    // per-channel partials of everything this block produced: lanes with equal
    // ecol (stride 8), then the 4 waves, fixed order
#pragma unroll
    for (int j = 0; j < 8; ++j)
      for (int d = 8; d < 64; d <<= 1) {
        s1[j] += __shfl_xor(s1[j], d);
        s2[j] += __shfl_xor(s2[j], d);
      }
    float* red = sE;  // [4][2][64]
    __syncthreads();
    if (erow == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * 2 + 0) * 64 + ecol + j] = s1[j];
        red[(wave * 2 + 1) * 64 + ecol + j] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 128) {
      const float t = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
      a.stats[(size_t)blockIdx.x * 128 + tid] = t;  // [block][2][64]
    }
  }
}

template <int DT, int SPLITW, int POOL>
int launch(const ConvArgs& a, hipStream_t s, int* m_tiles_out) {
  const int tiles_x = POOL ? (a.pool_wo + POOL_OUT - 1) / POOL_OUT : (a.Wo + TILE - 1) / TILE;
  const int tiles_y = POOL ? (a.pool_ho + POOL_OUT - 1) / POOL_OUT : (a.Ho + TILE - 1) / TILE;
  const int n_tiles = a.N * tiles_x * tiles_y;
  const size_t lds = (SPLITW ? 2 : 1) * (POOL ? 32 : 64) * W_ROW_BYTES + 2 * PATCH_BYTES +
                     (POOL ? TILE * POOL_OUT * 80 : 4 * 16 * 68 * 4);
  int grid = 256 * 2;  // 2 blocks per CU fit (74 KB of LDS; POOL: 49 / 65 KB, a block per 32-cout half)
  if (SPLITW && !POOL) grid = 256;   // 107 KB: one block per CU
  if (POOL) {
    if (grid > 2 * n_tiles) grid = 2 * n_tiles;   // pairs of blocks: (tile stream, cout half)
  } else if (grid > n_tiles) {
    grid = n_tiles;
  }
  auto k = conv_stem_kernel<DT, SPLITW, POOL>;
  static std::atomic<unsigned long long> attr;
  (void)spk_lds_limit_once(attr, (const void*)k, (int)lds);
  if (m_tiles_out) *m_tiles_out = grid;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a, tiles_x, tiles_y, n_tiles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// Stem forward (7x7/2 pad 3, Cout = 64, NHWC4 input).  Returns -2 when the
// problem is not this shape (the caller then uses the generic kernel).
int spk_conv_stem_launch(const ConvArgs& a, hipStream_t s, int* m_tiles_out) {
  if (a.Cout != 64 || a.kh != 7 || a.stride != 2 || a.pad != 3 || a.res) return -2;
  if (a.pool_y) {   // fused max-pool (eval): the pooled dims must be those of a 3x3 / 2 pad-1 pool of this stem output
    if (!a.relu || a.stats || a.pool_ho != (a.Ho - 1) / 2 + 1 || a.pool_wo != (a.Wo - 1) / 2 + 1) return -2;
    if (a.dt == DT_F16) return a.splitw ? launch<DT_F16, 1, 1>(a, s, m_tiles_out) : launch<DT_F16, 0, 1>(a, s, m_tiles_out);
    return launch<DT_BF16, 0, 1>(a, s, m_tiles_out);
  }
  if (a.dt == DT_F16) return a.splitw ? launch<DT_F16, 1, 0>(a, s, m_tiles_out) : launch<DT_F16, 0, 0>(a, s, m_tiles_out);
  return launch<DT_BF16, 0, 0>(a, s, m_tiles_out);
}
