// 1x1 convolution for the DEEP layers of a ResNet's last stages on gfx950 (MI355X), eval path, fp16 storage, single
// weight image: out[pixel][cout] = act(BN(sum_k x[pixel][k] w[cout][k]) (+ shortcut)) with K = 256 ... 2048 and few pixels
// (14 x 14 and 7 x 7 maps).  Stands in for the torch Conv2d(k=1) + BatchNorm2d.eval() (+ add) (+ ReLU) links the reference
// reaches through `net(x)` (sykepic/compute/probability.py:189) - SURVEY.md section 2.2.
//
// Why a third 1x1 kernel.  On these layers the implicit GEMM (conv_igemm.hip: both operands through LDS, one barrier per
// 64-deep K step, 4 waves) runs at 0.46-0.65 PFLOP/s and conv_pw.hip's ring flavour (activations straight to registers,
// weights through LDS) loses to it: a half batch is 6-25 k pixels, i.e. 49-196 tiles of 128 rows, and every block spends
// its life behind K-step barriers with one wave per SIMD (a wave issues back-to-back 16x16x32 MFMAs at half the pipe's
// rate: tools/micro/mfma_shape.hip).  This kernel is phase 1 of conv_bneck.hip on its own:
//   * 8 waves, two per SIMD; a wave owns 64 couts (two 32-cout pairs) x MT pixel tiles, the block BN = 64 WN couts x
//     BM = 16 MT (8 / WN) pixels;
//   * the ACTIVATIONS stream through an LDS ring in 64-channel chunks by LDS-DMA (128-byte rows, 16-byte chunks
//     XOR-swizzled at the source), one barrier per chunk, each fragment read feeds four MFMAs;
//   * the WEIGHTS come straight from L2 into MFMA operand registers in fragment order (pack_pw_kernel's image): no LDS
//     traffic and no barrier for them, each byte is loaded once per wave that owns its couts;
//   * both run D = 2-3 chunks AHEAD of the MFMAs (phase 1 of conv_bneck.hip fetches a chunk's weights one chunk ahead:
//     0.4-0.75 us of MFMAs against an L2 round trip of ~1 us under load - it waits at the top of every chunk);
//   * swapped operand roles and the permuted couts of conv_pw.hip: the epilogue runs from registers.
// K order (32-deep steps ascending) and the fp32 epilogue are those of conv_pw.hip / conv_igemm.hip: bit-identical
// results (tests/test_gpu_pw.py runs every configuration against the others), so the tuner's choice never shows.
#include "spk_common.h"
#include <cstdio>
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) const unsigned char* lds_u8_t;
typedef __attribute__((address_space(3))) const u32x4_t* lds_u32x4_t;
typedef __attribute__((address_space(3))) const f32x4_t* lds_f32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

__device__ __forceinline__ unsigned int pack2h_nosat(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_t));
}

// A 16-byte buffer load the compiler's s_waitcnt pass does not see (conv_bneck.hip explains: hipcc 7.2 does not count
// LDS-DMA instructions when it derives the vmcnt of a register load's first use).  Ordered by the kernel's own s_waitcnt;
// the destination registers must never be spilled or copied before that wait ("VGPRs Spill: 0" for every instantiation).
__device__ __forceinline__ u32x4_t buffer_load_b128_untracked(u32x4_t rsrc, unsigned voff, unsigned soff, int imm) {
  u32x4_t d;
  if (imm == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else if (imm == 1024) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else if (imm == 2048) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:2048" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:3072" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return d;
}

template <int MT_, int WN_, int D_>
struct PwrCfg {
  static constexpr int MT = MT_, WN = WN_, D = D_, NW = 8, WM = NW / WN;   // D: chunks the memory system runs ahead of the MFMAs
  static constexpr int BM = 16 * MT * WM, BN = 64 * WN;
  static constexpr int DI = (BM / 8 + NW - 1) / NW;       // LDS-DMA instructions (8 rows x 128 B) per wave and chunk
  static constexpr int XSTAGE = DI * NW * 1024;
  static constexpr int NXS = D + 1;                        // ring stages
  static constexpr int LDS = NXS * XSTAGE + 2 * BN * 4;    // + [BN] scale, [BN] shift
  static constexpr int DEPTH = 4;                          // activation fragments in flight (ring of registers)
  static_assert(NW % WN == 0, "waves");
  static_assert(2 * MT >= DI, "a chunk has fewer MFMA groups than LDS-DMA instructions to place between them");
  static_assert((D - 1) * (DI + 8) <= 63 && D >= 1 && D <= 3, "vmcnt is a 6-bit counter");
  static_assert(LDS <= 160 * 1024, "LDS");
};

// DUAL: K-concatenated second activation source (PwConvArgs::x2: a block-closing conv and the block's 1x1 shortcut conv as
// one GEMM, conv_pw.hip): chunks from Cin / 64 on stream from x2 at the output pixel's position under stride2.
template <int MT, int WN, int D, bool HAS_RES, bool DUAL>
__global__ __launch_bounds__(512, 2) void conv_pwr_kernel(PwConvArgs a, int m_tiles, int n_tiles) {
  using K = PwrCfg<MT, WN, D>;
  constexpr int NW = K::NW, BM = K::BM, BN = K::BN, DI = K::DI, XSTAGE = K::XSTAGE, NXS = K::NXS, DEPTH = K::DEPTH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, p = lane & 15;
  const int wn = wave % WN, wm = wave / WN;

  // block -> tile: XCD-aware bijective map, the n tiles of one m tile adjacent on one XCD (the activation rows the first of
  // them pulls from HBM are an L2 hit for the others)
  const int ntiles = m_tiles * n_tiles, bw = blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bw & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bw >> 3);
  const int nt = swz % n_tiles, mt = swz / n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int Cin = a.Cin, Cout = a.Cout;
  const int nch1 = Cin >> 6;                               // 64-channel chunks of the first source
  const int NCH = DUAL ? (Cin + a.Cin2) >> 6 : nch1;       // ... in all (even: the host checks)

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.x2 : a.x), 0, DUAL ? a.x2_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_RES ? a.res : a.y), 0, HAS_RES ? a.y_bytes : 0, 0x00020000);

  unsigned char* const ring = smem;
  float* const tab = (float*)(smem + NXS * XSTAGE);        // [BN] scale, [BN] shift of this block's couts
  for (int c = tid; c < BN; c += NW * 64) {
    tab[c] = a.scale ? a.scale[n0 + c] : 1.f;
    tab[BN + c] = a.shift ? a.shift[n0 + c] : 0.f;
  }

  // LDS-DMA of one 64-channel chunk: instruction ii = wave + NW i covers tile rows 8 ii .. 8 ii + 7; lane l lands in row
  // l / 8, slot l % 8 and therefore fetches the chunk whose swizzled slot that is
  unsigned voff[DI], voff2[DUAL ? DI : 1];
#pragma unroll
  for (int i = 0; i < DI; ++i) {
    const int row = 8 * (wave + NW * i) + (lane >> 3);
    const int gm = m0 + row;
    const bool ok = row < BM && gm < a.M;
    const unsigned slot = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    voff[i] = ok ? (unsigned)gm * (unsigned)(Cin * 2) + slot : 0x80000000u;
    if (DUAL) {
      const int hw = a.Ho * a.Wo, img = gm / hw, rem = gm - img * hw, ho = rem / a.Wo, wo = rem - ho * a.Wo;
      const int px2 = (img * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2;
      voff2[i] = ok ? (unsigned)px2 * (unsigned)(a.Cin2 * 2) + slot : 0x80000000u;
    }
  }
  // (the casts of the vector offsets: hipcc 7.2's host pass drops the kernel when an lvalue array element is passed there)
  auto dma_piece = [&](int c, int stage, int i) {
    lds_ptr_t dst = (lds_ptr_t)(ring + stage * XSTAGE + (wave + NW * i) * 1024);
    if (DUAL && c >= nch1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, dst, 16, (unsigned)voff2[DUAL ? i : 0], (c - nch1) * 128, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, (unsigned)voff[i], c * 128, 0, 0);
  };
  auto dma = [&](int c, int stage) {
#pragma unroll
    for (int i = 0; i < DI; ++i) dma_piece(c, stage, i);
  };
  // activation fragment of pixel tile j, K step ks of a chunk: row 16 (wm MT + j) + p, chunk 4 ks + g, swizzled
  // ((row >> 1) & 7 == (p >> 1) & 7: tiles start on multiples of 16; K step 1 = the byte offset xor 64)
  const unsigned xlane = (unsigned)((wm * MT * 16 + p) * 128 + ((g ^ ((p >> 1) & 7)) << 4));
  // weight fragments of this wave's 64 couts: [K step][pair][tile][lane][8] images, pairs n0 / 32 + 2 wn and + 1
  const unsigned w_lane = (unsigned)(n0 / 32 + 2 * wn) * 2048 + lane * 16;
  const unsigned kstep_bytes = (unsigned)Cout * 64;        // one 32-deep K step of the whole image
  const unsigned long long wpp = (unsigned long long)a.wp;
  const u32x4_t rws = {(unsigned)wpp, (unsigned)(wpp >> 32) & 0xffffu, (unsigned)NCH * 64u * (unsigned)Cout * 2u, 0x00020000u};
  // Weight fragments of chunk k live in set k % D: [ks][tile of the wave's 64 couts].  The memory system runs D chunks ahead
  // of the MFMAs: while chunk k is multiplied, the LDS-DMA pieces of chunk k + D are issued between its first MFMA groups
  // and the weight fragments of chunk k + D are loaded INTO chunk k's own registers, each half right behind its last use
  // (K step 0 after MFMA group MT - 1, K step 1 after the last group) - D sets, not D + 1.  With D = 1 a wave waits for
  // its weights at the top of every chunk: a chunk is 0.4-0.75 us of MFMAs, an L2 round trip under load about 1 us.
  u32x4_t wa[D][2][4];
  auto load_w_ks = [&](u32x4_t (&d)[2][4], int c, int ks) {
#pragma unroll
    for (int t = 0; t < 4; ++t) d[ks][t] = buffer_load_b128_untracked(rws, w_lane, (unsigned)(2 * c + ks) * kstep_bytes, t * 1024);
  };
  f32x4_t acc[MT][4];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // prologue: chunks 0 .. D - 1 in flight, each as [LDS-DMA pieces, weight fragments] - the order of every later chunk
#pragma unroll
  for (int k = 0; k < D; ++k)
    if (k < NCH) {
      dma(k, k);
      load_w_ks(wa[k], k, 0);
      load_w_ks(wa[k], k, 1);
    }
  int st_cur = 0, st_fill = D;   // ring stage of chunk c / of chunk c + D
  auto chunk = [&](int c, u32x4_t (&w)[2][4]) {
    // Chunk c's pieces and fragments have landed once only what was issued after them is outstanding: vmcnt retires in
    // issue order and every later chunk j in (c, c + D) put DI + 8 operations behind them - as many as exist.  The barrier
    // publishes everyone's pieces and proves everyone is done with chunk c - 1, whose stage is refilled at once.
    const int newer = NCH - 1 - c;
    if (D >= 3 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DI + 8)) : "memory");
    else if (D >= 2 && newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DI + 8) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const bool more = c + D < NCH;
    lds_u8_t xb0 = (lds_u8_t)ring + st_cur * XSTAGE + xlane;
    lds_u8_t xb1 = (lds_u8_t)ring + st_cur * XSTAGE + (xlane ^ 64u);
    asm volatile("" : "+v"(xb0), "+v"(xb1));
    u32x4_t fr[DEPTH];
    constexpr int UNITS = 2 * MT;
    auto addr = [&](int u) { return (u / MT ? xb1 : xb0) + (u % MT) * 2048; };
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) fr[u] = *(lds_u32x4_t)addr(u);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      if (u + DEPTH - 1 < UNITS) fr[(u + DEPTH - 1) % DEPTH] = *(lds_u32x4_t)addr(u + DEPTH - 1);
      const int j = u % MT, ks = u / MT;
      // hard fences, not hints: left to itself hipcc keeps ONE fragment register set and reads each fragment right in front
      // of its four MFMAs - a full LDS round trip exposed per unit (conv_c3.hip, conv_bneck.hip)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[j][t] = mfma16<DT_F16>(w[ks][t], fr[u % DEPTH], acc[j][t]);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        if (u < DI) dma_piece(c + D, st_fill, u);
        if (u == MT - 1) load_w_ks(w, c + D, 0);
        if (u == UNITS - 1) load_w_ks(w, c + D, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    st_cur = st_cur == NXS - 1 ? 0 : st_cur + 1;
    st_fill = st_fill == NXS - 1 ? 0 : st_fill + 1;
  };
  for (int c = 0; c < NCH; c += D) {
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (c + i < NCH) chunk(c + i, wa[i]);
  }
  // every untracked load has been waited for by the last chunk's s_waitcnt vmcnt(0) (no dead load past the last chunk:
  // the compiler re-uses those registers from here on)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue from registers: the lane holds couts n0 + 64 wn + 32 P + 8 g .. + 7 of pixel m0 + 16 (wm MT + j) + p ----
  const float relu_floor = a.relu == 1 ? 0.f : -65504.f;
  const unsigned col = (unsigned)(n0 + 64 * wn + 8 * g) * 2;
  u32x4_t rq[2][2];
  auto res_load = [&](int j, u32x4_t (&d)[2]) {
    const int gm = m0 + 16 * (wm * MT + j) + p;
    const unsigned off = gm < a.M ? (unsigned)gm * (unsigned)(Cout * 2) + col : 0x80000000u;
    d[0] = __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0);
    d[1] = __builtin_amdgcn_raw_buffer_load_b128(rr, off + 64, 0, 0);
  };
  if (HAS_RES) res_load(0, rq[0]);
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    // the next tile's shortcut values are requested BEFORE this tile's stores (vmcnt retires in issue order: a load waited
    // for behind stores sits through their HBM write latency)
    if (HAS_RES && j + 1 < MT) res_load(j + 1, rq[(j + 1) & 1]);
    const int gm = m0 + 16 * (wm * MT + j) + p;
    const unsigned yoff = gm < a.M ? (unsigned)gm * (unsigned)(Cout * 2) + col : 0x80000000u;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      lds_f32x4_t sp = (lds_f32x4_t)(tab + 64 * wn + 32 * P + 8 * g);
      asm volatile("" : "+v"(sp));
      const f32x4_t sc0 = sp[0], sc1 = sp[1], sh0 = sp[BN / 4], sh1 = sp[BN / 4 + 1];
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = __builtin_fmaf(acc[j][2 * P][r], sc0[r], sh0[r]);
        v[4 + r] = __builtin_fmaf(acc[j][2 * P + 1][r], sc1[r], sh1[r]);
      }
      if (HAS_RES) {
        const u32x4_t q = rq[j & 1][P];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[2 * i] += lo_f32<DT_F16>(q[i]);
          v[2 * i + 1] += hi_f32<DT_F16>(q[i]);
        }
      }
      if (a.relu == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = silu_f(v[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], relu_floor, 65504.f);
      u32x4_t ov;
#pragma unroll
      for (int i = 0; i < 4; ++i) ov[i] = pack2h_nosat(v[2 * i], v[2 * i + 1]);
      __builtin_amdgcn_raw_buffer_store_b128(ov, ry, yoff + P * 64, 0, 0);
    }
  }
}

template <int MT, int WN, int D, bool HAS_RES, bool DUAL>
int launch_k(const PwConvArgs& a, hipStream_t s, int m_tiles, int n_tiles) {
  using K = PwrCfg<MT, WN, D>;
  auto k = conv_pwr_kernel<MT, WN, D, HAS_RES, DUAL>;
  static std::atomic<unsigned long long> attr;
  (void)spk_lds_limit_once(attr, (const void*)k, K::LDS);
  hipLaunchKernelGGL(k, dim3(m_tiles * n_tiles), dim3(512), K::LDS, s, a, m_tiles, n_tiles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int MT, int WN, int D>
int launch(const PwConvArgs& a, hipStream_t s) {
  using K = PwrCfg<MT, WN, D>;
  if (a.Cout % K::BN) return -3;
  const int m_tiles = (a.M + K::BM - 1) / K::BM, n_tiles = a.Cout / K::BN;
  if (a.x2) return a.res ? -3 : launch_k<MT, WN, D, false, true>(a, s, m_tiles, n_tiles);   // (the fused shortcut conv IS the shortcut)
  return a.res ? launch_k<MT, WN, D, true, false>(a, s, m_tiles, n_tiles) : launch_k<MT, WN, D, false, false>(a, s, m_tiles, n_tiles);
}

}  // namespace

int spk_pwr_num_configs() { return 6; }

// 0 ok, -1 HIP error, -3 this configuration does not fit the problem.  Plain fp16 1x1 convs of stride 1 with one weight
// image; an optional second activation source (any stride); K a multiple of 128 in all.
int spk_pwr_launch(const PwConvArgs& a, int cfg, hipStream_t s) {
  if (a.dt != DT_F16 || a.nb != 1 || a.wpz || a.stride != 1 || a.Cin % 64 || a.Cout % 256 || a.M <= 0) return -3;
  const int K = a.Cin + (a.x2 ? a.Cin2 : 0);
  if (K % 128 || K < 256) return -3;
  if (a.x2 && (a.Cin2 % 64 || (size_t)a.x2_bytes >= 0x80000000ull)) return -3;
  if ((size_t)a.y_bytes >= 0x80000000ull || (size_t)a.x_bytes >= 0x80000000ull) return -3;
  switch (cfg) {
    case 0: return launch<7, 8, 2>(a, s);    // 112 pixels x 512 couts
    case 1: return launch<4, 8, 3>(a, s);    //  64 x 512, three chunks ahead (short chunks)
    case 2: return launch<6, 4, 2>(a, s);    // 192 x 256 (7 tiles per wave: 256 VGPRs and spills - the untracked loads forbid that)
    case 3: return launch<4, 4, 3>(a, s);    // 128 x 256
    case 4: return launch<7, 8, 3>(a, s);    // 112 x 512, three chunks ahead
    case 5: return launch<5, 8, 3>(a, s);    //  80 x 512
    default: return -3;
  }
}
