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
//   * the ACTIVATIONS stream through a three-stage LDS ring in 64-channel chunks by LDS-DMA (128-byte rows, 16-byte
//     chunks XOR-swizzled at the source), one barrier per chunk, each fragment read feeds four MFMAs;
//   * the WEIGHTS come straight from L2 into MFMA operand registers in fragment order (pack_pw_kernel's image): no LDS
//     traffic and no barrier for them, each byte is loaded once per wave that owns its couts;
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

template <int MT_, int WN_>
struct PwrCfg {
  static constexpr int MT = MT_, WN = WN_, NW = 8, WM = NW / WN;
  static constexpr int BM = 16 * MT * WM, BN = 64 * WN;
  static constexpr int DI = (BM / 8 + NW - 1) / NW;       // LDS-DMA instructions (8 rows x 128 B) per wave and chunk
  static constexpr int XSTAGE = DI * NW * 1024;
  static constexpr int NXS = 3;                            // ring stages
  static constexpr int LDS = NXS * XSTAGE + 2 * BN * 4;    // + [BN] scale, [BN] shift
  static constexpr int DEPTH = 4;                          // activation fragments in flight (ring of registers)
  static_assert(NW % WN == 0, "waves");
  static_assert(2 * MT >= 2 + DI, "a chunk has fewer MFMA groups than memory instructions to place between them");
  static_assert(LDS <= 160 * 1024, "LDS");
};

template <int MT, int WN, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void conv_pwr_kernel(PwConvArgs a, int m_tiles, int n_tiles) {
  using K = PwrCfg<MT, WN>;
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
  const int NCH = Cin >> 6;                                // 64-channel chunks (even: the host checks Cin % 128)

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
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
  unsigned voff[DI];
#pragma unroll
  for (int i = 0; i < DI; ++i) {
    const int row = 8 * (wave + NW * i) + (lane >> 3);
    const int gm = m0 + row;
    voff[i] = (row < BM && gm < a.M) ? (unsigned)gm * (unsigned)(Cin * 2) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4) : 0x80000000u;
  }
  auto dma = [&](int c, int stage) {
#pragma unroll
    for (int i = 0; i < DI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(ring + stage * XSTAGE + (wave + NW * i) * 1024), 16, (unsigned)voff[i],   // (the cast: hipcc 7.2's host pass drops the kernel when an lvalue array element is passed here)
                                               c * 128, 0, 0);
  };
  auto dma_piece = [&](int c, int stage, int i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(ring + stage * XSTAGE + (wave + NW * i) * 1024), 16, (unsigned)voff[i],
                                             c * 128, 0, 0);
  };
  // activation fragment of pixel tile j, K step ks of a chunk: row 16 (wm MT + j) + p, chunk 4 ks + g, swizzled
  // ((row >> 1) & 7 == (p >> 1) & 7: tiles start on multiples of 16; K step 1 = the byte offset xor 64)
  const unsigned xlane = (unsigned)((wm * MT * 16 + p) * 128 + ((g ^ ((p >> 1) & 7)) << 4));
  // weight fragments of this wave's 64 couts: [K step][pair][tile][lane][8] images, pairs n0 / 32 + 2 wn and + 1
  const unsigned w_lane = (unsigned)(n0 / 32 + 2 * wn) * 2048 + lane * 16;
  const unsigned kstep_bytes = (unsigned)Cout * 64;        // one 32-deep K step of the whole image
  const unsigned long long wpp = (unsigned long long)a.wp;
  const u32x4_t rws = {(unsigned)wpp, (unsigned)(wpp >> 32) & 0xffffu, (unsigned)Cin * (unsigned)Cout * 2u, 0x00020000u};
  u32x4_t wa[2][2][4];   // [chunk parity][ks][tile of the wave's 64 couts]
  auto load_w_ks = [&](u32x4_t (&d)[2][4], int c, int ks) {
#pragma unroll
    for (int t = 0; t < 4; ++t) d[ks][t] = buffer_load_b128_untracked(rws, w_lane, (unsigned)(2 * c + ks) * kstep_bytes, t * 1024);
  };
  f32x4_t acc[MT][4];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  load_w_ks(wa[0], 0, 0);
  load_w_ks(wa[0], 0, 1);
  dma(0, 0);
  dma(1, 1);   // NCH >= 2
  int st_cur = 0, st_fill = 2;   // ring stage of chunk c / of chunk c + 2
  auto chunk = [&](int c, const u32x4_t (&w)[2][4], u32x4_t (&wnext)[2][4]) {
    // this wave's pieces of chunk c have landed once at most the next chunk's DMAs are outstanding; the barrier publishes
    // everyone's pieces and proves everyone is done with chunk c - 1, whose stage is refilled at once.  The chunk's weight
    // fragments - requested one chunk ago, BEFORE the DMAs that may still be in flight - are complete here too.
    if (c + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const bool more_w = c + 1 < NCH, more_x = c + 2 < NCH;
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
      // the next chunk's weight fragments, then the DMA pieces of chunk c + 2, between the MFMA groups (order: weights
      // first - it is what the s_waitcnt at the top counts on)
      if (u < 2) {
        if (more_w) load_w_ks(wnext, c + 1, u);
      } else if (u < 2 + DI) {
        if (more_x) dma_piece(c + 2, st_fill, u - 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    st_cur = st_cur == NXS - 1 ? 0 : st_cur + 1;
    st_fill = st_fill == NXS - 1 ? 0 : st_fill + 1;
  };
  for (int c = 0; c < NCH; c += 2) {
    chunk(c, wa[0], wa[1]);
    chunk(c + 1, wa[1], wa[0]);
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

template <int MT, int WN>
int launch(const PwConvArgs& a, hipStream_t s) {
  using K = PwrCfg<MT, WN>;
  if (a.Cout % K::BN) return -3;
  const int m_tiles = (a.M + K::BM - 1) / K::BM, n_tiles = a.Cout / K::BN;
  static std::atomic<unsigned long long> attr_r, attr_n;
  if (a.res) {
    auto k = conv_pwr_kernel<MT, WN, true>;
    (void)spk_lds_limit_once(attr_r, (const void*)k, K::LDS);
    hipLaunchKernelGGL(k, dim3(m_tiles * n_tiles), dim3(512), K::LDS, s, a, m_tiles, n_tiles);
  } else {
    auto k = conv_pwr_kernel<MT, WN, false>;
    (void)spk_lds_limit_once(attr_n, (const void*)k, K::LDS);
    hipLaunchKernelGGL(k, dim3(m_tiles * n_tiles), dim3(512), K::LDS, s, a, m_tiles, n_tiles);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

int spk_pwr_num_configs() { return 4; }

// 0 ok, -1 HIP error, -3 this configuration does not fit the problem.  Plain fp16 1x1 convs of stride 1 with one weight
// image and one activation source; K a multiple of 128.
int spk_pwr_launch(const PwConvArgs& a, int cfg, hipStream_t s) {
  if (a.dt != DT_F16 || a.nb != 1 || a.x2 || a.wpz || a.stride != 1 || a.Cin % 128 || a.Cin < 256 || a.Cout % 256 || a.M <= 0) return -3;
  if ((size_t)a.y_bytes >= 0x80000000ull || (size_t)a.x_bytes >= 0x80000000ull) return -3;
  switch (cfg) {
    case 0: return launch<7, 8>(a, s);    // 112 pixels x 512 couts
    case 1: return launch<4, 8>(a, s);    //  64 x 512
    case 2: return launch<6, 4>(a, s);    // 192 x 256 (7 tiles per wave: 256 VGPRs and spills - the untracked loads forbid that)
    case 3: return launch<4, 4>(a, s);    // 128 x 256
    default: return -3;
  }
}
