// 3x3 stride-1 pad-1 convolution with a HALO SLAB per filter row (forward; eval and train).
//
// The implicit-GEMM kernel (conv_igemm.hip) fetches one [BM x 64] activation tile per tap: 9 fetches of what
// is, for a tile of BM consecutive output pixels, the same BM (+halo) input pixels shifted by one column or
// one image row.  Its main loop is bound by the L2->LDS fill (LDS-DMA sustains ~70 GB/s per CU, DESIGN.md
// section 5), so here ONE slab of BM+2 consecutive input pixels is fetched per (filter row r, 64-channel
// block) and serves the three taps s = 0,1,2 of that row: tap s reads the MFMA A fragments of output row m
// from slab row (m - m0) + s.  Activation traffic through the fill path drops 3x; the weight tiles of the
// three taps ride along in the same stage.
//
// Padding and image borders: the slab is a run of consecutive pixels of the flattened [N*H*W] axis, so a row
// of it may belong to the previous/next image row or image.  Whether tap (r,s) of output pixel (h,w) is a real
// input pixel (0 <= h+r-1 < H, 0 <= w+s-1 < W) is a per-lane bit; lanes whose tap is padding read a zero row of
// LDS instead (the fragment address is per lane anyway, so the select costs one v_cndmask).
//
// Same operand layouts, LDS swizzle, MFMA (16x16x32 f16/bf16) and epilogue as conv_igemm.hip; K order per output
// element (filter row, channel block, tap column): identical to it when Cin = 64, else a few 1-ulp fp16 roundings move.
#include "spk_common.h"

#include <cstdio>
#include <cstdlib>

namespace {

constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int DT, int SPLITW>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void conv3x3_slab_kernel(ConvArgs a, int m_tiles, int n_tiles) {
  constexpr int NTHREADS = WARPS_M * WARPS_N * 64;
  constexpr int RPP = NTHREADS / 8;  // rows per DMA pass
  constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int NB = SPLITW ? 2 : 1;
  // The slab region holds BM rows: BM-1 slab rows (pixels m0-1 ... m0+BM-3) and, as its LAST row, a row that is
  // always fetched out of range - the DMA refills it with zeros on every fill: the zero row of that stage.  A
  // tile therefore produces BM_EFF = BM-3 output rows (the MFMA rows past them are masked): 2.3 % of the MFMA
  // work at BM = 128 for an LDS footprint of exactly BM + 3*NB*BN rows (two 40 KB stages, two blocks per CU).
  constexpr int BM_EFF = BM - 3;
  constexpr int A_PASSES = BM / RPP;
  constexpr int A_ROWS = BM;
  constexpr int B_ROWS = 3 * NB * BN;              // [tap s][hi, lo][cout]
  constexpr int B_ITERS = B_ROWS / RPP;
  constexpr int A_BYTES = A_ROWS * ROW_BYTES, B_BYTES = B_ROWS * ROW_BYTES;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int PER_TILE = A_PASSES + B_ITERS;
  static_assert(BM % RPP == 0 && B_ROWS % RPP == 0 && RPP % 16 == 0, "tile / block mismatch");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;

  // XCD-aware bijective tile map (as conv_igemm.hip)
  const int ntiles = m_tiles * n_tiles;
  const int wk = blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = wk & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wk >> 3);
  const int nt_idx = swz % n_tiles, mt_idx = swz / n_tiles;
  const int m0 = mt_idx * BM_EFF, n0 = nt_idx * BN;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  // ---- staging coordinates ----
  const int srow = tid >> 3;
  const int chunk_b = (tid & 7) ^ ((srow >> 1) & 7);  // DMA lands lane l in slot l&7: fetch the chunk whose slot that is
  const int cb = a.Cin / BK;                          // channel blocks
  const int KT = 3 * cb;                              // K steps: (filter row, channel block)
  int a_off[A_PASSES];                                // byte offset of slab row j (filter row 1, channel block 0)
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int j = srow + i * RPP;
    // slab row j <-> flattened pixel m0 - 1 + j; the last row (and pixels before the tensor, through the
    // unsigned range check) fetches nothing and lands as zeros
    a_off[i] = j < BM - 1 ? ((m0 - 1 + j) * a.Cin + chunk_b * 8) * 2 : (int)0x80000000;
  }
  int b_off[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    const int rr = srow + i * RPP;  // [0, 3*NB*BN): tap, half, cout
    const int s = rr / (NB * BN), rem = rr - s * (NB * BN);
    const int half = rem / BN, n = rem - half * BN;
    b_off[i] = ((half * a.Cout + n0 + n) * a.K + s * a.Cin + chunk_b * 8) * 2;
  }
  const int row_pitch = a.W * a.Cin * 2;  // bytes between image rows

  auto issue = [&](int kt, int stage) {
    const int r = kt / cb, c0 = kt - r * cb;
    unsigned char* const dA = smem + stage * STAGE_BYTES + wave * (8 * ROW_BYTES);
    unsigned char* const dB = dA + A_BYTES;
    const int arel = (r - 1) * row_pitch + c0 * (BK * 2);
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const unsigned off = a_off[i] == (int)0x80000000 ? 0x80000000u : (unsigned)(a_off[i] + arel);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(dA + i * (RPP * ROW_BYTES)), 16, off, 0, 0, 0);
    }
    const int brel = (r * 3 * a.Cin + c0 * BK) * 2;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dB + i * (RPP * ROW_BYTES)), 16,
                                               (unsigned)(b_off[i] + brel), 0, 0, 0);
  };

  // ---- per-lane validity of the taps of this lane's A-fragment rows ----
  const int frow = lane & 15, fq = lane >> 4;
  const int HW = a.H * a.W;
  unsigned okw[MT], okh[MT];  // bit s: column tap s is inside the image; bit r: row tap r is
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int ml = wm * WM + i * 16 + frow;
    const int m = m0 + ml;
    const int rem = m % HW;
    const int h = rem / a.W, w = rem - h * a.W;
    const bool live = ml < BM_EFF && m < a.M;
    okw[i] = live ? ((w > 0 ? 1u : 0u) | 2u | (w < a.W - 1 ? 4u : 0u)) : 0u;
    okh[i] = live ? ((h > 0 ? 1u : 0u) | 2u | (h < a.H - 1 ? 4u : 0u)) : 0u;
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // epilogue geometry (needed before the K loop for the shortcut prefetch)
  constexpr int EPI_LD = WN + 4;
  constexpr int LPR = WN / 8;
  constexpr int ERPP = 64 / LPR;
  constexpr int PASSES = 16 / ERPP;
  static_assert(PASSES >= 1, "WN too large");
  static_assert(WARPS_M * WARPS_N * 16 * EPI_LD * 4 <= STAGE_BYTES, "epilogue LDS");
  const int ecol = (lane % LPR) * 8, erow = lane / LPR;
  constexpr bool PREFETCH_RES = MT * PASSES <= 8;
  const int gcol = n0 + wn * WN + ecol;
  u32x4_t rres[PREFETCH_RES ? MT : 1][PREFETCH_RES ? PASSES : 1];
  if (PREFETCH_RES && a.res) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int ml = wm * WM + i * 16 + erow + p * ERPP;
        const int m = m0 + ml;
        rres[i][p] = (ml < BM_EFF && m < a.M) ? *(const u32x4_t*)(a.res + (size_t)m * a.Cout + gcol)
                                              : u32x4_t{0, 0, 0, 0};
      }
  }

  // ---- main loop: two stages, one barrier per K step (cf. conv_igemm.hip flavour 3) ----
  issue(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    const int cs = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile kt is complete; everyone is done reading the other stage
    if (kt + 1 < KT) issue(kt + 1, cs ^ 1);
    const unsigned char* pa = smem + cs * STAGE_BYTES;
    const unsigned char* pb = pa + A_BYTES;
    const unsigned char* zero_row = pa + (BM - 1) * ROW_BYTES;  // refilled with zeros by every fill of this stage
    const int r = kt / cb;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4_t fa[MT], fb[NT], fl[SPLITW ? NT : 1];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const bool ok = ((okw[i] >> s) & (okh[i] >> r) & 1u) != 0;
          const unsigned char* p = ok ? pa + lds_off(wm * WM + i * 16 + frow + s, ks * 4 + fq) : zero_row;
          fa[i] = *(const u32x4_t*)p;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
          fb[j] = *(const u32x4_t*)(pb + lds_off(s * NB * BN + wn * WN + j * 16 + frow, ks * 4 + fq));
        if (SPLITW) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
            fl[j] = *(const u32x4_t*)(pb + lds_off(s * NB * BN + BN + wn * WN + j * 16 + frow, ks * 4 + fq));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<DT>(fa[i], fb[j], acc[i][j]);
        if (SPLITW) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mfma16<DT>(fa[i], fl[j], acc[i][j]);
        }
      }
    }
  }
  __builtin_amdgcn_s_barrier();  // tile buffers are reused by the epilogue

  // ---- epilogue (as conv_igemm.hip): acc -> per-wave LDS region -> fused pointwise -> 16-bit rows ----
  float* const epi = (float*)smem + wave * (16 * EPI_LD);
  float sc[8], bi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = a.scale ? a.scale[gcol + j] : 1.f;
    bi[j] = a.bias ? a.bias[gcol + j] : 0.f;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) epi[(fq * 4 + q) * EPI_LD + j * 16 + frow] = acc[i][j][q];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int row = erow + p * ERPP;
      const int ml = wm * WM + i * 16 + row;
      const int m = m0 + ml;
      const f32x4_t v0 = *(const f32x4_t*)(epi + row * EPI_LD + ecol);
      const f32x4_t v1 = *(const f32x4_t*)(epi + row * EPI_LD + ecol + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (ml < BM_EFF && m < a.M) {
        const size_t o = (size_t)m * a.Cout + gcol;
        if (a.stats) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + bi[j];
        if (a.res) {
          const u32x4_t rr = PREFETCH_RES ? rres[PREFETCH_RES ? i : 0][PREFETCH_RES ? p : 0]
                                          : *(const u32x4_t*)(a.res + o);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[2 * j] += lo_f32<DT>(rr[j]);
            v[2 * j + 1] += hi_f32<DT>(rr[j]);
          }
          if (a.res_lo) {
            const u32x4_t rl = *(const u32x4_t*)(a.res_lo + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[2 * j] += lo_f32<DT>(rl[j]);
              v[2 * j + 1] += hi_f32<DT>(rl[j]);
            }
          }
        }
        if (a.relu == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        } else if (a.relu == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
        }
        u32x4_t ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = pack2<DT>(v[2 * j], v[2 * j + 1]);
        *(u32x4_t*)(a.y + o) = ov;
        if (a.y_lo) {
          u32x4_t lv;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            lv[j] = pack2<DT>(v[2 * j] - lo_f32<DT>(ov[j]), v[2 * j + 1] - hi_f32<DT>(ov[j]));
          *(u32x4_t*)(a.y_lo + o) = lv;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      for (int d = LPR; d < 64; d <<= 1) {
        s1[j] += __shfl_xor(s1[j], d);
        s2[j] += __shfl_xor(s2[j], d);
      }
    }
    __syncthreads();
    float* red = (float*)smem;  // [WARPS_M][2][BN]
    if (erow == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wm * 2 + 0) * BN + wn * WN + ecol + j] = s1[j];
        red[(wm * 2 + 1) * BN + wn * WN + ecol + j] = s2[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NTHREADS) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WARPS_M; ++w) t += red[(w * 2) * BN + c];
      const int which = c / BN, col = c - which * BN;
      a.stats[((size_t)mt_idx * 2 + which) * a.Cout + n0 + col] = t;
    }
  }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int DT, int SPLITW>
int launch_slab(const ConvArgs& a, hipStream_t s, int* m_tiles_out) {
  constexpr size_t stage = (size_t)(BM + 3 * (SPLITW ? 2 : 1) * BN) * ROW_BYTES;
  constexpr size_t lds = 2 * stage;
  if (lds > 163840) return -3;
  auto k = conv3x3_slab_kernel<BM, BN, WARPS_M, WARPS_N, DT, SPLITW>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  static bool told = false;
  if (!told && getenv("SPK_TUNE_LOG") && atoi(getenv("SPK_TUNE_LOG")) > 1) {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k, WARPS_M * WARPS_N * 64, lds);
    fprintf(stderr, "[spk slab] %dx%d split %d: %zu B LDS, %d blocks per CU\n", BM, BN, SPLITW, lds, nb);
    told = true;
  }
  const int m_tiles = (a.M + (BM - 3) - 1) / (BM - 3), n_tiles = a.Cout / BN;
  if (m_tiles_out) *m_tiles_out = m_tiles;
  hipLaunchKernelGGL(k, dim3(m_tiles * n_tiles), dim3(WARPS_M * WARPS_N * 64), lds, s, a, m_tiles, n_tiles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int BM, int BN, int WARPS_M, int WARPS_N>
int launch_slab_cfg(const ConvArgs& a, hipStream_t s, int* m_tiles_out) {
  if (a.Cout % BN) return -3;
  if (a.dt == DT_F16) {
    if (a.splitw) return launch_slab<BM, BN, WARPS_M, WARPS_N, DT_F16, 1>(a, s, m_tiles_out);
    return launch_slab<BM, BN, WARPS_M, WARPS_N, DT_F16, 0>(a, s, m_tiles_out);
  }
  if (a.splitw) return -3;
  return launch_slab<BM, BN, WARPS_M, WARPS_N, DT_BF16, 0>(a, s, m_tiles_out);
}

}  // namespace

// 3x3 / stride 1 / pad 1 forward convs whose tensors are not channel-padded
bool spk_conv3x3_slab_eligible(const ConvArgs& a, int mode) {
  return mode == CONV_MODE_GENERIC && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 64 == 0 &&
         a.Cout % 64 == 0 && a.cin_s == 0 && a.cout_s == 0 && a.H == a.Ho && a.W == a.Wo && a.K == 9 * a.Cin &&
         (size_t)a.M * a.Cin * 2 < ((size_t)1 << 31);
}

// cfg: 0 = 128x64 (4 waves), 1 = 256x64 (8 waves), 2 = 128x128 (4 waves), 3 = 64x64 (4 waves)
int spk_conv3x3_slab_launch(const ConvArgs& a, int cfg, hipStream_t s, int* m_tiles_out) {
  switch (cfg) {
    case 0: return launch_slab_cfg<128, 64, 2, 2>(a, s, m_tiles_out);
    case 1: return launch_slab_cfg<256, 64, 4, 2>(a, s, m_tiles_out);
    case 2: return launch_slab_cfg<128, 128, 2, 2>(a, s, m_tiles_out);
    case 3: return launch_slab_cfg<64, 64, 2, 2>(a, s, m_tiles_out);
  }
  return -3;
}
