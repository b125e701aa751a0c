// A whole identity bottleneck block of a ResNet-50/101/152 as ONE kernel for gfx950 (MI355X), eval path, fp16 storage,
// single weight images (the calibrated / plain-fp16 modes):
//     out = ReLU(BN3(conv3_1x1(ReLU(BN2(conv2_3x3(ReLU(BN1(conv1_1x1(x)))))))) + x)
// - the three Conv2d + BatchNorm2d.eval() (+ ReLU) links of torchvision's `Bottleneck.forward` that the reference reaches
// through `net(x)` (sykepic/compute/probability.py:189; SURVEY.md section 2.2, section 8d: "only reachable if activations
// cross HBM ~ once each way").  As three launches the block moves x, y1 (write + up to 9 L2 re-reads), y2 (write + read),
// x again (shortcut) and out through HBM / L2 and pays three grid tails; here the two mid tensors never leave the CU.
//
// One block = one band of R output rows of one image (R = H: a whole 14 x 14 image), 8 waves, two per SIMD (<= 256
// registers each: past that hipcc splits the accumulators between VGPRs and AGPRs and shuffles them around every MFMA).
//   phase 1  y1 = ReLU(BN1(W1 . x)) on the band's rows plus one halo row above and below.  x streams through an LDS ring in
//            64-channel chunks by LDS-DMA (128-byte rows, 16-byte chunks XOR-swizzled at the SOURCE: conv_igemm.hip), the
//            weights come straight from L2 in MFMA fragment order (pack_pw_kernel's image).  y1 lands in LDS as a WINDOW:
//            plane q holds the 16-byte channel part q (8 channels) of every window position, positions at pitch W + 1 - the
//            one extra column per row is zero and serves as the right padding of its row AND the left padding of the next -
//            so that every 3x3 tap is a constant position offset and a fragment read at any tap is conflict-free
//            (conv_c3.hip's layout).  Rows outside the image are zero.
//   phase 2  y2 = ReLU(BN2(W2 * y1)): 9 taps x CM channels out of the window, weights from L2 in conv_c3.hip's fragment
//            order (K order chunk -> tap -> half, the same sums as that kernel).  y2 overwrites the window.
//   phase 3  out = ReLU(BN3(W3 . y2) + x), 4 passes of CM couts each; the shortcut rows of x are requested before a pass's K
//            loop and land under it.
// In every phase a wave owns 64 couts (two 32-cout pairs, permuted at pack time so that a lane's accumulators of a pair
// are 8 CONSECUTIVE couts of one position: epilogues run from registers, conv_pw.hip) and one of WM = 512 / CM shares of the position
// tiles; the ACTIVATION operand is read from LDS by all waves, each weight byte is loaded by exactly one wave of a cout
// group.  K orders and epilogue arithmetic are those of conv_pw.hip / conv_c3.hip, so the block's output is bit-identical to
// the three-launch path (tests/test_gpu_bneck.py).
#include "spk_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

typedef __attribute__((address_space(3))) const unsigned char* lds_u8_t;
typedef __attribute__((address_space(3))) const u32x4_t* lds_u32x4_t;
typedef __attribute__((address_space(3))) const f32x4_t* lds_f32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }

template <int CM_, int HW_, int R_, int NW_>
struct BnCfg {
  static constexpr int CM = CM_, HW = HW_, R = R_, NW = NW_;   // NW: waves per block (8: two per SIMD; 4: one, two blocks per CU)
  static constexpr int C4 = 4 * CM;
  static constexpr int WN = CM / 64, WM = NW / WN;      // the waves: along the couts x along the positions
  static constexpr int WP = HW + 1;                     // window pitch
  static constexpr int NROW = R + 2;                    // window rows: one halo row above, one below
  static constexpr int P2 = R * WP;                     // output positions of a band at pitch WP (column HW is no output)
  static constexpr int NT2 = (P2 + 15) / 16;
  static constexpr int MTW = (NT2 + WM - 1) / WM;       // position tiles per wave, phases 2 and 3
  static constexpr int RV = cmin(R + 2, HW);            // image rows a window can hold
  static constexpr int P1 = RV * HW;                    // ... as compact positions (phase 1 computes real positions only)
  static constexpr int NT1 = (P1 + 15) / 16;
  static constexpr int MT1 = (NT1 + WM - 1) / WM;       // position tiles per wave, phase 1
  // window positions: every tap of every (also the padding) output position of the tile grid stays inside
  static constexpr int NVA = cmax(NROW * WP + 1, WM * MTW * 16 + 2 * WP + 2);
  static constexpr int PLANE = (NVA * 16 + 255) & ~255;
  static constexpr int WIN = (CM / 8) * PLANE;
  static constexpr int DI = (WM * MT1 * 2 + NW - 1) / NW;   // LDS-DMA instructions (8 rows x 128 B) per wave and x chunk
  static constexpr int XSTAGE = DI * NW * 1024;
  // x ring stages (the ring lives in the window's space).  Three only in the 8-wave form: with three stages the weight
  // fragments of phase 1 are loaded by instructions the compiler does not track (buffer_load_b128_untracked), which must
  // never be spilled - the 4-wave instantiations sit at the 256-register limit and do spill a few values
  static constexpr int NXS = (NW == 8 && 3 * XSTAGE <= cmax(WIN, 96 * 1024)) ? 3 : 2;
  static constexpr int REGION = cmax(WIN, NXS * XSTAGE);
  // BN tables in LDS: [s1 b1 | s2 b2 | s3 b3] = (2 + 2 + 8) CM floats - or, where a 4-wave block would then pass the 80 KB
  // that let two blocks share a CU, conv3's only (phases 1 and 2 read theirs from memory, once per wave)
  static constexpr bool SMALLTAB = NW == 4 && REGION + 12 * CM * 4 > 80 * 1024;
  static constexpr int TAB = (SMALLTAB ? 8 : 12) * CM * 4;
  static constexpr int T3 = SMALLTAB ? 0 : 4 * CM;      // float offset of s3 in the table
  static constexpr int LDS = REGION + TAB;
  static constexpr int NCH = C4 / 64;                   // x chunks
  static constexpr int DEPTH = 4;                       // activation fragments in flight (ring of registers)
  static_assert(CM == 64 || CM == 128 || CM == 256, "mid channels");
  static_assert(HW % R == 0, "bands tile the image");
  static_assert(NW % WN == 0 && (NW == 4 || NW == 8), "waves");
  static_assert(LDS <= 160 * 1024 && (NW == 8 || LDS <= 80 * 1024), "LDS");
};

// two clamped floats -> one dword of two fp16 values
__device__ __forceinline__ unsigned int pack2h(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_t));
}

// A 16-byte buffer load the compiler's s_waitcnt pass does not see (phase 1's weight fragments).  hipcc 7.2 does not count
// LDS-DMA instructions when it derives the vmcnt of a register load's first use, so with builtin loads next to the DMAs it
// waits for "all but N" with an N that is too small by the DMAs issued in between - i.e. for the x chunk still in flight.
// The caller orders these loads with its own s_waitcnt (the same that publishes the DMA pieces).
__device__ __forceinline__ u32x4_t buffer_load_b128_untracked(u32x4_t rsrc, unsigned voff, unsigned soff, int imm) {
  u32x4_t d;
  if (imm == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else if (imm == 1024) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else if (imm == 2048) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:2048" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:3072" : "=v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return d;
}

template <int CM, int HW, int R, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv_bneck_kernel(BneckArgs a) {
  using K = BnCfg<CM, HW, R, NW>;
  constexpr int NT = NW * 64;
  constexpr int C4 = K::C4, WN = K::WN, WP = K::WP, MTW = K::MTW, MT1 = K::MT1, PLANE = K::PLANE, DI = K::DI;
  constexpr int NXS = K::NXS, XSTAGE = K::XSTAGE, NCH = K::NCH, DEPTH = K::DEPTH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, p = lane & 15;
  const int wn = wave % WN, wm = wave / WN;

  // ---- block -> (image, band): blocks b and b + 8 share an XCD (round-robin dispatch), so an XCD takes a contiguous run
  // of (image, band) items and the halo rows two neighbouring bands both read are an L2 hit for the second ----
  constexpr int BPI = HW / R;
  const int nblk = a.N * BPI;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
  const int img = lin / BPI, band = lin - img * BPI;
  const int r0 = band * R;
  const int rlo = r0 > 0 ? r0 - 1 : 0, rhi = r0 + R + 1 < HW ? r0 + R + 1 : HW;   // image rows [rlo, rhi) of the window
  const int P1 = (rhi - rlo) * HW;                                                  // its real positions

  // (timing experiments, a.flags: 8 every block reads image 0 in phase 1 - x from L2; 16 no shortcut loads; 32 no stores)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (a.flags & 16) ? 0 : a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (a.flags & 32) ? 0 : a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, 9 * CM * CM * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w3, 0, CM * C4 * 2, 0x00020000);

  unsigned char* const win = smem;                       // x ring (phase 1), then the y1 window, then y2
  float* const tab = (float*)(smem + K::REGION);         // [s1 | b1 | s2 | b2 |] s3 | b3
  if (!K::SMALLTAB)
    for (int c = tid; c < CM; c += NT) {
      tab[c] = a.s1[c]; tab[CM + c] = a.b1[c];
      tab[2 * CM + c] = a.s2[c]; tab[3 * CM + c] = a.b2[c];
    }
  for (int c = tid; c < C4; c += NT) { tab[K::T3 + c] = a.s3[c]; tab[K::T3 + C4 + c] = a.b3[c]; }

  // diagnostics only: wave 0 leaves the shader clock at each phase boundary (a.stamps null in the product path)
  auto stamp = [&](int i) {
    if (a.stamps && (tid == 0 || tid == NT / 2)) a.stamps[(size_t)blockIdx.x * 16 + (tid ? 8 : 0) + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

  // folded BatchNorm scale / shift of 8 consecutive couts from `c0` on (conv1 / conv2): from the LDS table, or - where the
  // block keeps conv3's table only - from memory (two 16-byte loads each, once per wave and cout pair)
  auto bn_pair = [&](const float* gs, const float* gb, int toff, int c0, f32x4_t& sc0, f32x4_t& sc1, f32x4_t& sh0, f32x4_t& sh1) {
    if (K::SMALLTAB) {
      sc0 = *(const f32x4_t*)(gs + c0); sc1 = *(const f32x4_t*)(gs + c0 + 4);
      sh0 = *(const f32x4_t*)(gb + c0); sh1 = *(const f32x4_t*)(gb + c0 + 4);
    } else {
      lds_f32x4_t sp = (lds_f32x4_t)(tab + toff + c0);
      asm volatile("" : "+v"(sp));
      sc0 = sp[0]; sc1 = sp[1]; sh0 = sp[CM / 4]; sh1 = sp[CM / 4 + 1];
    }
  };

  // weight fragments of this wave's 64 couts: [K step][pair][tile][lane][8] images, pairs 2 wn and 2 wn + 1
  const unsigned w_lane = (unsigned)(2 * wn) * 2048 + lane * 16;

  // =====================================================================================================================
  // phase 1: y1 = ReLU(BN1(W1 . x)) on the window's real positions (compact index: row-major over image rows [rlo, rhi))
  // =====================================================================================================================
  {
    // LDS-DMA of one 64-channel chunk: instruction ii = wave + NW i covers rows 8 ii .. 8 ii + 7; lane l lands in row
    // l / 8, slot l % 8 and therefore fetches the chunk whose swizzled slot that is
    unsigned voff[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int row = 8 * (wave + NW * i) + (lane >> 3);
      const int pr = row / HW, pc = row - pr * HW;
      const unsigned pix = (unsigned)((((a.flags & 8) ? 0 : img) * HW + rlo + pr) * HW + pc);
      voff[i] = row < P1 ? pix * (unsigned)(C4 * 2) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4) : 0x80000000u;
    }
    auto dma = [&](int c, int stage) {
#pragma unroll
      for (int i = 0; i < DI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(win + stage * XSTAGE + (wave + NW * i) * 1024), 16, (unsigned)voff[i],   // (the cast: hipcc 7.2 host pass silently drops the kernel when an lvalue array element is passed here)
                                                 c * 128, 0, 0);
    };
    // activation fragment of position tile j, K step ks of a chunk: row 16 (wm MT1 + j) + p, chunk 4 ks + g, swizzled
    // ((row >> 1) & 7 == (p >> 1) & 7: tiles start on multiples of 16)
    // (K step 1 of a chunk = chunk 4 + g: slot (4 + g) ^ s = (g ^ s) ^ 4 - the byte offset xor 64)
    const unsigned xlane = (unsigned)((wm * MT1 * 16 + p) * 128 + ((g ^ ((p >> 1) & 7)) << 4));
    u32x4_t wa[2][2][4];   // [chunk parity][ks][tile of the wave's 64 couts]
    // (the descriptor by hand for the asm form: base, stride 0, bytes, raw-buffer flags - make_buffer_rsrc's words)
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, CM * C4 * 2, 0x00020000);
    const unsigned long long w1p = (unsigned long long)a.w1;
    const u32x4_t rw1s = {(unsigned)w1p, (unsigned)(w1p >> 32) & 0xffffu, (unsigned)(CM * C4 * 2), 0x00020000u};
    auto load_w = [&](u32x4_t (&d)[2][4], int c) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // (two ring stages: every wait is vmcnt(0) anyway - plain loads, which the compiler may also spill or copy; the
          // untracked form must never be: its registers are not valid until the manual wait.  Check the instantiations
          // with three stages for "VGPRs Spill: 0" when this file changes.)
          if (NXS == 2) d[ks][t] = __builtin_amdgcn_raw_buffer_load_b128(rw1, w_lane + t * 1024, (2 * c + ks) * (CM * 64), 0);
          else d[ks][t] = buffer_load_b128_untracked(rw1s, w_lane, (unsigned)((2 * c + ks) * (CM * 64)), t * 1024);
        }
    };
    auto load_w_ks = [&](u32x4_t (&d)[2][4], int c, int ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (NXS == 2) d[ks][t] = __builtin_amdgcn_raw_buffer_load_b128(rw1, w_lane + t * 1024, (2 * c + ks) * (CM * 64), 0);
        else d[ks][t] = buffer_load_b128_untracked(rw1s, w_lane, (unsigned)((2 * c + ks) * (CM * 64)), t * 1024);
      }
    };
    auto dma_piece = [&](int c, int stage, int i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(win + stage * XSTAGE + (wave + NW * i) * 1024), 16, (unsigned)voff[i],
                                               c * 128, 0, 0);
    };
    f32x4_t acc[MT1][4];
#pragma unroll
    for (int j = 0; j < MT1; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Three ring stages (the 8-wave form): the memory system runs TWO chunks ahead of the MFMAs for both operands.  The weight
    // fragments of chunk c + 2 are loaded into chunk c's own registers, each half right behind its last use (K step 0 after
    // MFMA group MT1 - 1, K step 1 after the last group): two sets, as before, but a full chunk more of lead - with one chunk
    // (0.4-0.75 us of MFMAs against an L2 round trip of ~1 us under load) every wave waited for its weights at the top of
    // every chunk (conv_pwr.hip, where this was measured: 47.9 -> 43.6 us on the 14 x 14 stage opener).  Two stages (the
    // 4-wave form): plain loads one chunk ahead, every wait is vmcnt(0).
    if (NXS == 3) {
      dma(0, 0);
      load_w(wa[0], 0);
      dma(1, 1);
      load_w(wa[1], 1);
    } else {
      load_w(wa[0], 0);
      dma(0, 0);
    }
    auto chunk = [&](int c, u32x4_t (&w)[2][4], u32x4_t (&wnext)[2][4]) {
      // Chunk c's pieces and fragments have landed once only what was issued after them is outstanding (vmcnt retires in
      // issue order): with three stages that is chunk c + 1's DI pieces and 8 fragment loads - if there is a chunk c + 1.
      // The barrier publishes everyone's pieces and proves everyone is done with chunk c - 1, whose stage is refilled at once.
      if (NXS == 3 && c + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DI + 8) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // Memory instructions are issued BETWEEN the MFMA groups below, a few per group (an MFMA holds the issue port for half of
      // its 16 cycles): issued in one go they cost every wave ~0.5 k cycles per chunk in front of its first MFMA, in step on
      // all waves of the CU.  (No dead load past the last chunk: nothing would ever wait for it - see below the loop.)
      const bool more_w = c + (NXS == 3 ? 2 : 1) < NCH, more_x = c + NXS - 1 < NCH;
      lds_u8_t xb0 = (lds_u8_t)win + (c % NXS) * XSTAGE + xlane;
      lds_u8_t xb1 = (lds_u8_t)win + (c % NXS) * XSTAGE + (xlane ^ 64u);
      asm volatile("" : "+v"(xb0), "+v"(xb1));
      u32x4_t fr[DEPTH];
      constexpr int UNITS = 2 * MT1;
      auto addr = [&](int u) { return (u / MT1 ? xb1 : xb0) + (u % MT1) * 2048; };
#pragma unroll
      for (int u = 0; u < DEPTH - 1; ++u) fr[u] = *(lds_u32x4_t)addr(u);
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        if (u + DEPTH - 1 < UNITS) fr[(u + DEPTH - 1) % DEPTH] = *(lds_u32x4_t)addr(u + DEPTH - 1);
        const int j = u % MT1, ks = u / MT1;
        // hard fences, not hints: left to itself hipcc keeps ONE fragment register set and reads each fragment right in
        // front of its four MFMAs - a full LDS round trip exposed per unit (conv_c3.hip, conv_pw.hip)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[j][t] = mfma16<DT_F16>(w[ks][t], fr[u % DEPTH], acc[j][t]);
        __builtin_amdgcn_sched_barrier(0);
        if (NXS == 3) {
          if (more_x && u < DI) dma_piece(c + 2, (c + 2) % NXS, u);
          if (more_w && u == MT1 - 1) load_w_ks(w, c + 2, 0);
          if (more_w && u == UNITS - 1) load_w_ks(w, c + 2, 1);
        } else if (u < 2) {
          if (more_w) load_w_ks(wnext, c + 1, u);
        } else if (u < 2 + DI) {
          if (more_x) dma_piece(c + NXS - 1, (c + NXS - 1) % NXS, u - 2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      static_assert(2 * MT1 >= 2 + DI, "a chunk has fewer MFMA groups than memory instructions to place between them");
    };
    for (int c = 0; c < NCH; c += 2) {
      chunk(c, wa[0], wa[1]);
      chunk(c + 1, wa[1], wa[0]);
    }
    // Every untracked load has been waited for by the last chunk's s_waitcnt vmcnt(0).  This matters: the compiler believes
    // their destination registers dead from here on and re-uses them - a load still in flight would land in whatever lives
    // there by then (seen as one wrong image in a few hundred blocks while the last chunk still issued a dead load).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(1);
    __syncthreads();   // everyone is done with the x ring: the window takes its place

    // zeros wherever the window has no image pixel: the shared padding column, rows outside the image, the tail
    for (int u = tid; u < (CM / 8) * K::NVA; u += NT) {
      const int pl = u / K::NVA, v = u - pl * K::NVA;
      const int wr = v / WP, wc = v - wr * WP;
      const int ir = r0 - 1 + wr;
      if (!(wc >= 1 && ir >= 0 && ir < HW && wr < K::NROW)) *(u32x4_t*)(win + pl * PLANE + v * 16) = u32x4_t{0, 0, 0, 0};
    }
    // y1 from registers: lane holds couts 64 wn + 32 P + 8 g .. + 7 of position 16 (wm MT1 + j) + p = plane 8 wn + 4 P + g
#pragma unroll
    for (int j = 0; j < MT1; ++j) {
      const int row = 16 * (wm * MT1 + j) + p;
      const int pr = row / HW, pc = row - pr * HW;
      const int v = (rlo + pr - (r0 - 1)) * WP + pc + 1;
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        f32x4_t sc0, sc1, sh0, sh1;
        bn_pair(a.s1, a.b1, 0, 64 * wn + 32 * P + 8 * g, sc0, sc1, sh0, sh1);
        u32x4_t ov;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ov[i] = pack2h(__builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P][2 * i], sc0[2 * i], sh0[2 * i]), 0.f, 65504.f),
                         __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P][2 * i + 1], sc0[2 * i + 1], sh0[2 * i + 1]), 0.f, 65504.f));
          ov[2 + i] = pack2h(__builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P + 1][2 * i], sc1[2 * i], sh1[2 * i]), 0.f, 65504.f),
                             __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P + 1][2 * i + 1], sc1[2 * i + 1], sh1[2 * i + 1]), 0.f, 65504.f));
        }
        if (row < P1) *(u32x4_t*)(win + (8 * wn + 4 * P + g) * PLANE + v * 16) = ov;
      }
    }
    __syncthreads();
  }

  stamp(2);
  // position tiles of this wave in phases 2 and 3: window-pitch positions 16 (wm MTW + j) + p
  const unsigned plane_lane = (unsigned)(g * PLANE + (wm * MTW * 16 + p) * 16);

  // =====================================================================================================================
  // phase 2: y2 = ReLU(BN2(W2 * y1)), K order chunk (64 channels) -> tap -> half (conv_c3.hip's, bit for bit)
  // =====================================================================================================================
  {
    constexpr int PW = 3;                      // weight K steps in flight (divides the 18 steps of a chunk)
    constexpr int KSTEPS = (CM / 64) * 18;
    u32x4_t wq[PW][4];
    auto load_w = [&](u32x4_t (&d)[4], int step) {
#pragma unroll
      for (int t = 0; t < 4; ++t) d[t] = __builtin_amdgcn_raw_buffer_load_b128(rw2, w_lane + t * 1024, step * (CM * 64), 0);
    };
    f32x4_t acc[MTW][4];
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < PW; ++s) load_w(wq[s], s);
    for (int c = 0; c < CM / 64; ++c) {
      lds_u8_t cb = (lds_u8_t)win + plane_lane + c * 8 * PLANE;
      asm volatile("" : "+v"(cb));
      constexpr int UNITS = 18 * MTW;
      // unit u = (K step q of the chunk, position tile j): tap q / 2 shifts the positions, half q % 2 the planes
      auto addr = [&](int u) {
        const int q = u / MTW, j = u % MTW, tap = q >> 1, kk = q & 1;
        return cb + kk * 4 * PLANE + (j * 16 + (tap / 3) * WP + (tap % 3)) * 16;
      };
      u32x4_t fr[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH - 1; ++u) fr[u] = *(lds_u32x4_t)addr(u);
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        if (u + DEPTH - 1 < UNITS) fr[(u + DEPTH - 1) % DEPTH] = *(lds_u32x4_t)addr(u + DEPTH - 1);
        const int q = u / MTW, j = u % MTW;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[j][t] = mfma16<DT_F16>(wq[q % PW][t], fr[u % DEPTH], acc[j][t]);
        __builtin_amdgcn_sched_barrier(0);
        if (j == MTW - 1) {   // (past the last K step the same fragments are fetched again: no branch in the step)
          const int step = c * 18 + q + PW;
          load_w(wq[q % PW], step < KSTEPS ? step : KSTEPS - 1);
        }
      }
    }
    stamp(3);
    __syncthreads();   // everyone has read its last y1 fragment: y2 takes the window's place (positions at pitch WP)
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        f32x4_t sc0, sc1, sh0, sh1;
        bn_pair(a.s2, a.b2, 2 * CM, 64 * wn + 32 * P + 8 * g, sc0, sc1, sh0, sh1);
        u32x4_t ov;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ov[i] = pack2h(__builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P][2 * i], sc0[2 * i], sh0[2 * i]), 0.f, 65504.f),
                         __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P][2 * i + 1], sc0[2 * i + 1], sh0[2 * i + 1]), 0.f, 65504.f));
          ov[2 + i] = pack2h(__builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P + 1][2 * i], sc1[2 * i], sh1[2 * i]), 0.f, 65504.f),
                             __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P + 1][2 * i + 1], sc1[2 * i + 1], sh1[2 * i + 1]), 0.f, 65504.f));
        }
        *(u32x4_t*)(win + plane_lane + (8 * wn + 4 * P) * PLANE + j * 256) = ov;
      }
    }
    __syncthreads();
  }

  // =====================================================================================================================
  // phase 3: out = ReLU(BN3(W3 . y2) + x) in 8 half passes: in half pass hp a wave computes the 32 couts
  // (hp / 2) 64 WN + 64 wn + 32 (hp % 2) .. + 31 of its position tiles.  Everything a half pass reads from memory is
  // requested one half pass earlier, BEFORE the previous epilogue's stores (vmcnt retires in issue order: a load waited for
  // behind stores sits through their write latency): its weight fragments when the previous K loop has consumed the
  // register set, its shortcut values into the second of two register sets.
  // =====================================================================================================================
  {
    constexpr int KS3 = CM / 32;
    stamp(4);
    // output / shortcut row segments: position q = 16 (wm MTW + j) + p -> pixel (r0 + q / WP, q % WP), 8 couts from 8 g
    unsigned yoff[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      const int q = 16 * (wm * MTW + j) + p;
      const int orow = q / WP, ocol = q - orow * WP;
      yoff[j] = (orow < R && ocol < HW) ? (unsigned)((img * HW + r0 + orow) * HW + ocol) * (unsigned)(C4 * 2) + (unsigned)(64 * wn + 8 * g) * 2
                                        : 0x80000000u;
    }
    // The two waves of a SIMD run the same program from the same barrier: left alone they multiply together (sharing the
    // matrix pipe) and then run their epilogues together (sharing the vector issue).  A static priority for waves 4-7 lets
    // them take the pipe first; the other half then multiplies while they are in their epilogue, and the halves stay
    // out of phase (MI355X_MICROARCH.md, "Two waves per SIMD", items 4 and 9).
    if (NW == 8 && wave >= 4 && !(a.flags & 1)) __builtin_amdgcn_s_setprio(1);
    u32x4_t wq[KS3][2];      // the half pass's weight fragments: K step x tile of the pair
    u32x4_t rq[2][MTW];      // shortcut values: this half pass's and the next one's
    auto col = [&](int hp) { return (hp >> 1) * 64 * WN + 32 * (hp & 1); };   // first cout of the half pass (this wave: + 64 wn)
    auto load_w = [&](int hp) {
#pragma unroll
      for (int ks = 0; ks < KS3; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          wq[ks][t] = __builtin_amdgcn_raw_buffer_load_b128(rw3, w_lane + t * 1024, (ks * (C4 / 32) + (hp >> 1) * 2 * WN + (hp & 1)) * 2048, 0);
    };
    auto load_r = [&](u32x4_t (&d)[MTW], int hp) {
#pragma unroll
      for (int j = 0; j < MTW; ++j) d[j] = __builtin_amdgcn_raw_buffer_load_b128(rxr, yoff[j], col(hp) * 2, 0);
    };
    load_w(0);
    load_r(rq[0], 0);
    lds_u8_t sb0 = (lds_u8_t)win + plane_lane, sb1 = sb0 + (KS3 > 4 ? 16 * PLANE : 0);
    asm volatile("" : "+v"(sb0), "+v"(sb1));
    auto half_pass = [&](int hp, const u32x4_t (&rcur)[MTW], u32x4_t (&rnext)[MTW]) {
      f32x4_t acc[MTW][2];
#pragma unroll
      for (int j = 0; j < MTW; ++j) acc[j][0] = acc[j][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      constexpr int UNITS = KS3 * MTW;
      // unit u = (K step ks, position tile j): planes 4 ks + g (two bases: the ds_read immediate reaches 64 KB)
      auto addr = [&](int u) {
        const int ks = u / MTW, j = u % MTW;
        return (ks < 4 ? sb0 + ks * 4 * PLANE : sb1 + (ks - 4) * 4 * PLANE) + j * 256;
      };
      // (eight fragments in flight instead of four - a unit is two MFMAs here, half of phase 2's - measured the same: the phase
      // is bound by instruction issue, 2 MFMAs + ~5 VALU per unit and wave, not by the LDS latency)
      constexpr int D3 = DEPTH;
      u32x4_t fr[D3];
#pragma unroll
      for (int u = 0; u < D3 - 1; ++u) fr[u] = *(lds_u32x4_t)addr(u);
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        if (u + D3 - 1 < UNITS) fr[(u + D3 - 1) % D3] = *(lds_u32x4_t)addr(u + D3 - 1);
        const int ks = u / MTW, j = u % MTW;
        __builtin_amdgcn_sched_barrier(0);
        acc[j][0] = mfma16<DT_F16>(wq[ks][0], fr[u % D3], acc[j][0]);
        acc[j][1] = mfma16<DT_F16>(wq[ks][1], fr[u % D3], acc[j][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next half pass's operands, in front of this one's stores
      if (hp + 1 < 8) {
        load_w(hp + 1);
        load_r(rnext, hp + 1);
      }
      if (hp == 0) stamp(7);
      __builtin_amdgcn_sched_barrier(0);
      const int co = col(hp);
      lds_f32x4_t sp = (lds_f32x4_t)(tab + K::T3 + co + 64 * wn + 8 * g);
      asm volatile("" : "+v"(sp));
      const f32x4_t sc0 = sp[0], sc1 = sp[1], sh0 = sp[C4 / 4], sh1 = sp[C4 / 4 + 1];
#pragma unroll
      for (int j = 0; j < MTW; ++j) {
        // two values per VALU instruction where the ISA has a packed fp32 form (v_pk_fma_f32, v_pk_add_f32: the same IEEE
        // results as the scalar forms): 28 instead of 36 per 8 outputs in the phase that is bound by instruction issue
        f32x2_t v2[4];
        v2[0] = __builtin_elementwise_fma(__builtin_shufflevector(acc[j][0], acc[j][0], 0, 1), __builtin_shufflevector(sc0, sc0, 0, 1),
                                          __builtin_shufflevector(sh0, sh0, 0, 1));
        v2[1] = __builtin_elementwise_fma(__builtin_shufflevector(acc[j][0], acc[j][0], 2, 3), __builtin_shufflevector(sc0, sc0, 2, 3),
                                          __builtin_shufflevector(sh0, sh0, 2, 3));
        v2[2] = __builtin_elementwise_fma(__builtin_shufflevector(acc[j][1], acc[j][1], 0, 1), __builtin_shufflevector(sc1, sc1, 0, 1),
                                          __builtin_shufflevector(sh1, sh1, 0, 1));
        v2[3] = __builtin_elementwise_fma(__builtin_shufflevector(acc[j][1], acc[j][1], 2, 3), __builtin_shufflevector(sc1, sc1, 2, 3),
                                          __builtin_shufflevector(sh1, sh1, 2, 3));
        const u32x4_t q = rcur[j];
        const unsigned q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        v2[0] += __builtin_convertvector(__builtin_bit_cast(f16x2_t, q0), f32x2_t);
        v2[1] += __builtin_convertvector(__builtin_bit_cast(f16x2_t, q1), f32x2_t);
        v2[2] += __builtin_convertvector(__builtin_bit_cast(f16x2_t, q2), f32x2_t);
        v2[3] += __builtin_convertvector(__builtin_bit_cast(f16x2_t, q3), f32x2_t);
        u32x4_t ov;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          ov[i] = pack2h(__builtin_amdgcn_fmed3f(v2[i][0], 0.f, 65504.f), __builtin_amdgcn_fmed3f(v2[i][1], 0.f, 65504.f));
        // (the column offset in the VECTOR operand, never an SGPR soffset on a store: conv_pw.hip's store-data hazard note)
        __builtin_amdgcn_raw_buffer_store_b128(ov, ry, yoff[j] + (unsigned)(co * 2), 0, 0);
      }
      if (hp == 0) stamp(6);
    };
    for (int hp = 0; hp < 8; hp += 2) {
      half_pass(hp, rq[0], rq[1]);
      half_pass(hp + 1, rq[1], rq[0]);
    }
    stamp(5);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The TAIL of an identity bottleneck block as one kernel, for the stage whose trunk is too wide to fuse the whole block
// (ResNet-50 stage 1, 56 x 56 x 256: a whole-block kernel would read x twice - conv1's operand and, a phase later, the
// shortcut - and move MORE bytes than the launches it replaces):
//     out = ReLU(BN3(W3 . ReLU(BN2(W2 * y1))) + x)            y1 = the block's conv1 output, already in memory
//     z   = ReLU(BNz(Wz . out))   (NPZ > 0)                   the NEXT block's conv1, from the output tile in registers
// i.e. conv2 3x3 + conv3 1x1 + shortcut (+ the chained conv of conv_pw.hip's NPZ flavour).  The 3x3 conv's output and its
// re-read never exist; per block the kernel moves y1 (with one halo row each side), x, out and z.  Phases 2 and 3 of the
// kernel above on a band of R rows: the band's y1 rows go straight into the LDS window, W2 / W3 / Wz stream from L2 in
// fragment order.  With CM = 64 every wave owns ALL 64 mid couts and, over the 8 half passes of phase 3, all 256 trunk
// couts of its position tiles: the packed output registers of half pass hp are the MFMA B fragments of K step hp of
// the chained conv (conv_pw.hip), accumulated across the half passes.  Same K orders and epilogues as conv_c3.hip /
// conv_pw.hip: out and z are bit-identical to the launches they replace.
// ---------------------------------------------------------------------------------------------------------------------
template <int CM_, int HW_, int R_, int NPZ_>
struct BtCfg {
  static constexpr int CM = CM_, HW = HW_, R = R_, NPZ = NPZ_;
  static constexpr int C4 = 4 * CM, COZ = 32 * NPZ;
  static constexpr int WN = CM / 64, WM = 8 / WN;
  static constexpr int WP = HW + 1, NROW = R + 2;
  static constexpr int P2 = R * WP;
  static constexpr int NT2 = (P2 + 15) / 16;
  static constexpr int MTW = (NT2 + WM - 1) / WM;
  static constexpr int NVA = cmax(NROW * WP + 1, WM * MTW * 16 + 2 * WP + 2);
  static constexpr int PLANE = (NVA * 16 + 255) & ~255;
  static constexpr int WIN = (CM / 8) * PLANE;
  static constexpr int TAB = (2 * CM + 2 * C4 + 2 * COZ) * 4;   // [s2 | b2 | s3 | b3 | sz | bz]
  static constexpr int LDS = WIN + TAB;
  static constexpr int UL = (NROW * HW * (CM / 8) + 511) / 512;  // 16-byte units of the y1 band per thread
  static constexpr int DEPTH = 4;
  static_assert(NPZ == 0 || WN == 1, "the chained conv needs every trunk cout of a position in one wave");
  static_assert(HW % R == 0, "bands tile the image");
  static_assert(LDS <= 160 * 1024, "LDS");
};

// (<= 128 registers where the chained conv is narrow: two blocks per CU - 52 KB of LDS each - whose phases overlap; these
// layers wait for HBM, and one block alone runs load -> conv2 -> conv3 strictly one after the other)
template <int CM, int HW, int R, int NPZ>
__global__ __launch_bounds__(512, NPZ <= 2 ? 4 : 2) void conv_btail_kernel(BneckArgs a) {
  using K = BtCfg<CM, HW, R, NPZ>;
  constexpr int C4 = K::C4, COZ = K::COZ, WN = K::WN, WP = K::WP, MTW = K::MTW, PLANE = K::PLANE, DEPTH = K::DEPTH;
  constexpr int UL = K::UL, NPL = CM / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, p = lane & 15;
  const int wn = wave % WN, wm = wave / WN;

  constexpr int BPI = HW / R;
  const int nblk = a.N * BPI;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
  const int img = lin / BPI, band = lin - img * BPI;
  const int r0 = band * R;
  const int rlo = r0 > 0 ? r0 - 1 : 0, rhi = r0 + R + 1 < HW ? r0 + R + 1 : HW;
  const int P1 = (rhi - rlo) * HW;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.y1, 0, a.x_bytes / 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(NPZ ? a.z : a.y), 0, NPZ ? a.z_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, 9 * CM * CM * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w3, 0, CM * C4 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwz = __builtin_amdgcn_make_buffer_rsrc((void*)(NPZ ? a.wz : a.w3), 0, NPZ ? COZ * C4 * 2 : 0, 0x00020000);

  unsigned char* const win = smem;                  // the y1 window, then y2
  float* const tab = (float*)(smem + K::WIN);       // [s2 | b2 | s3 | b3 | sz | bz]
  for (int c = tid; c < CM; c += 512) { tab[c] = a.s2[c]; tab[CM + c] = a.b2[c]; }
  for (int c = tid; c < C4; c += 512) { tab[2 * CM + c] = a.s3[c]; tab[2 * CM + C4 + c] = a.b3[c]; }
  if (NPZ)
    for (int c = tid; c < COZ; c += 512) { tab[2 * CM + 2 * C4 + c] = a.sz[c]; tab[2 * CM + 2 * C4 + COZ + c] = a.bz[c]; }

  // ---- the band's y1 rows -> window planes (a position's CM channels are 16 CM / 8 contiguous bytes: NPL neighbouring
  // threads read them as one run), zeros wherever the window has no image pixel ----
  {
    u32x4_t st[UL];
    unsigned dst[UL];
#pragma unroll
    for (int i = 0; i < UL; ++i) {
      const int u = tid + 512 * i;
      const int pos = u / NPL, pl = u - pos * NPL;
      const int pr = pos / HW, pc = pos - pr * HW;
      const bool ok = pos < P1;
      st[i] = __builtin_amdgcn_raw_buffer_load_b128(
          ry1, ok ? (unsigned)((img * HW + rlo + pr) * HW + pc) * (unsigned)(CM * 2) + (unsigned)pl * 16 : 0x80000000u, 0, 0);
      dst[i] = ok ? (unsigned)(pl * PLANE + ((rlo + pr - (r0 - 1)) * WP + pc + 1) * 16) : 0xffffffffu;
    }
    for (int u = tid; u < NPL * K::NVA; u += 512) {
      const int pl = u / K::NVA, v = u - pl * K::NVA;
      const int wr = v / WP, wc = v - wr * WP;
      const int ir = r0 - 1 + wr;
      if (!(wc >= 1 && ir >= 0 && ir < HW && wr < K::NROW)) *(u32x4_t*)(win + pl * PLANE + v * 16) = u32x4_t{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < UL; ++i)
      if (dst[i] != 0xffffffffu) *(u32x4_t*)(win + dst[i]) = st[i];
    __syncthreads();
  }

  const unsigned w_lane = (unsigned)(2 * wn) * 2048 + lane * 16;
  const unsigned plane_lane = (unsigned)(g * PLANE + (wm * MTW * 16 + p) * 16);

  // ---- phase 2: y2 = ReLU(BN2(W2 * y1)), K order chunk -> tap -> half (conv_c3.hip's) ----
  {
    constexpr int PW = 3;
    constexpr int KSTEPS = (CM / 64) * 18;
    u32x4_t wq[PW][4];
    auto load_w = [&](u32x4_t (&d)[4], int step) {
#pragma unroll
      for (int t = 0; t < 4; ++t) d[t] = __builtin_amdgcn_raw_buffer_load_b128(rw2, w_lane + t * 1024, step * (CM * 64), 0);
    };
    f32x4_t acc[MTW][4];
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < PW; ++s) load_w(wq[s], s);
    for (int c = 0; c < CM / 64; ++c) {
      lds_u8_t cb = (lds_u8_t)win + plane_lane + c * 8 * PLANE;
      asm volatile("" : "+v"(cb));
      constexpr int UNITS = 18 * MTW;
      auto addr = [&](int u) {
        const int q = u / MTW, j = u % MTW, tap = q >> 1, kk = q & 1;
        return cb + kk * 4 * PLANE + (j * 16 + (tap / 3) * WP + (tap % 3)) * 16;
      };
      u32x4_t fr[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH - 1; ++u) fr[u] = *(lds_u32x4_t)addr(u);
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        if (u + DEPTH - 1 < UNITS) fr[(u + DEPTH - 1) % DEPTH] = *(lds_u32x4_t)addr(u + DEPTH - 1);
        const int q = u / MTW, j = u % MTW;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[j][t] = mfma16<DT_F16>(wq[q % PW][t], fr[u % DEPTH], acc[j][t]);
        __builtin_amdgcn_sched_barrier(0);
        if (j == MTW - 1) {
          const int step = c * 18 + q + PW;
          load_w(wq[q % PW], step < KSTEPS ? step : KSTEPS - 1);
        }
      }
    }
    __syncthreads();   // everyone has read its last y1 fragment: y2 takes the window's place
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        lds_f32x4_t sp = (lds_f32x4_t)(tab + 64 * wn + 32 * P + 8 * g);
        asm volatile("" : "+v"(sp));
        const f32x4_t sc0 = sp[0], sc1 = sp[1], sh0 = sp[CM / 4], sh1 = sp[CM / 4 + 1];
        u32x4_t ov;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ov[i] = pack2h(__builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P][2 * i], sc0[2 * i], sh0[2 * i]), 0.f, 65504.f),
                         __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P][2 * i + 1], sc0[2 * i + 1], sh0[2 * i + 1]), 0.f, 65504.f));
          ov[2 + i] = pack2h(__builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P + 1][2 * i], sc1[2 * i], sh1[2 * i]), 0.f, 65504.f),
                             __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[j][2 * P + 1][2 * i + 1], sc1[2 * i + 1], sh1[2 * i + 1]), 0.f, 65504.f));
        }
        *(u32x4_t*)(win + plane_lane + (8 * wn + 4 * P) * PLANE + j * 256) = ov;
      }
    }
    __syncthreads();
  }

  // ---- phase 3: out = ReLU(BN3(W3 . y2) + x) in 8 half passes of 32 couts (+ the chained conv's K step per half pass) ----
  {
    constexpr int KS3 = CM / 32;
    constexpr int NTZ = NPZ ? 2 * NPZ : 1;      // 16-cout tiles of the chained conv
    unsigned yoff[MTW], zoff[NPZ ? MTW : 1];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      const int q = 16 * (wm * MTW + j) + p;
      const int orow = q / WP, ocol = q - orow * WP;
      const bool ok = orow < R && ocol < HW;
      const unsigned pix = (unsigned)((img * HW + r0 + orow) * HW + ocol);
      yoff[j] = ok ? pix * (unsigned)(C4 * 2) + (unsigned)(64 * wn + 8 * g) * 2 : 0x80000000u;
      if (NPZ) zoff[j] = ok ? pix * (unsigned)(COZ * 2) + (unsigned)(8 * g) * 2 : 0x80000000u;
    }
    if (wave >= 4 && !(a.flags & 1)) __builtin_amdgcn_s_setprio(1);
    u32x4_t wq[KS3][2];
    u32x4_t rq[2][MTW];
    u32x4_t wz[NTZ];
    f32x4_t az[NPZ ? MTW : 1][NTZ];
#pragma unroll
    for (int j = 0; j < (NPZ ? MTW : 1); ++j)
#pragma unroll
      for (int t = 0; t < NTZ; ++t) az[j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto col = [&](int hp) { return (hp >> 1) * 64 * WN + 32 * (hp & 1); };
    auto load_w = [&](int hp) {
#pragma unroll
      for (int ks = 0; ks < KS3; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          wq[ks][t] = __builtin_amdgcn_raw_buffer_load_b128(rw3, w_lane + t * 1024, (ks * (C4 / 32) + (hp >> 1) * 2 * WN + (hp & 1)) * 2048, 0);
    };
    auto load_z = [&](int hp) {   // K step hp of the chained conv: every cout tile
      if (NPZ) {
#pragma unroll
        for (int t = 0; t < NTZ; ++t) wz[t] = __builtin_amdgcn_raw_buffer_load_b128(rwz, lane * 16 + t * 1024, hp * (NPZ * 2048), 0);
      }
    };
    auto load_r = [&](u32x4_t (&d)[MTW], int hp) {
#pragma unroll
      for (int j = 0; j < MTW; ++j) d[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, yoff[j], col(hp) * 2, 0);
    };
    load_w(0);
    load_z(0);
    load_r(rq[0], 0);
    lds_u8_t sb0 = (lds_u8_t)win + plane_lane, sb1 = sb0 + (KS3 > 4 ? 16 * PLANE : 0);
    asm volatile("" : "+v"(sb0), "+v"(sb1));
    auto half_pass = [&](int hp, const u32x4_t (&rcur)[MTW], u32x4_t (&rnext)[MTW]) {
      f32x4_t acc[MTW][2];
#pragma unroll
      for (int j = 0; j < MTW; ++j) acc[j][0] = acc[j][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      constexpr int UNITS = KS3 * MTW;
      auto addr = [&](int u) {
        const int ks = u / MTW, j = u % MTW;
        return (ks < 4 ? sb0 + ks * 4 * PLANE : sb1 + (ks - 4) * 4 * PLANE) + j * 256;
      };
      u32x4_t fr[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH - 1 && u < UNITS; ++u) fr[u] = *(lds_u32x4_t)addr(u);
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        if (u + DEPTH - 1 < UNITS) fr[(u + DEPTH - 1) % DEPTH] = *(lds_u32x4_t)addr(u + DEPTH - 1);
        const int ks = u / MTW, j = u % MTW;
        __builtin_amdgcn_sched_barrier(0);
        acc[j][0] = mfma16<DT_F16>(wq[ks][0], fr[u % DEPTH], acc[j][0]);
        acc[j][1] = mfma16<DT_F16>(wq[ks][1], fr[u % DEPTH], acc[j][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (hp + 1 < 8) {
        load_w(hp + 1);
        load_r(rnext, hp + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int co = col(hp);
      lds_f32x4_t sp = (lds_f32x4_t)(tab + 2 * CM + co + 64 * wn + 8 * g);
      asm volatile("" : "+v"(sp));
      const f32x4_t sc0 = sp[0], sc1 = sp[1], sh0 = sp[C4 / 4], sh1 = sp[C4 / 4 + 1];
      u32x4_t ovs[MTW];
#pragma unroll
      for (int j = 0; j < MTW; ++j) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = __builtin_fmaf(acc[j][0][r], sc0[r], sh0[r]);
          v[4 + r] = __builtin_fmaf(acc[j][1][r], sc1[r], sh1[r]);
        }
        const u32x4_t q = rcur[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[2 * i] += lo_f32<DT_F16>(q[i]);
          v[2 * i + 1] += hi_f32<DT_F16>(q[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          ovs[j][i] = pack2h(__builtin_amdgcn_fmed3f(v[2 * i], 0.f, 65504.f), __builtin_amdgcn_fmed3f(v[2 * i + 1], 0.f, 65504.f));
        if (NPZ) {   // the values about to be stored ARE K step hp of the chained conv for this position tile
#pragma unroll
          for (int t = 0; t < NTZ; ++t) az[j][t] = mfma16<DT_F16>(wz[t], ovs[j], az[j][t]);
        }
      }
      if (hp + 1 < 8) load_z(hp + 1);   // (into the registers the MFMAs above have read: lands under the next K loop)
      // The stores LAST: vmcnt retires in issue order, so every load this wave waits for in the next half pass (weights,
      // shortcut values, the chained conv's fragments) must be older than them - a load behind a store sits through the
      // store's HBM write latency, once per half pass.
#pragma unroll
      for (int j = 0; j < MTW; ++j) __builtin_amdgcn_raw_buffer_store_b128(ovs[j], ry, yoff[j] + (unsigned)(co * 2), 0, 0);
    };
    for (int hp = 0; hp < 8; hp += 2) {
      half_pass(hp, rq[0], rq[1]);
      half_pass(hp + 1, rq[1], rq[0]);
    }
    if (NPZ) {
#pragma unroll
      for (int j = 0; j < MTW; ++j)
#pragma unroll
        for (int Pz = 0; Pz < NPZ; ++Pz) {
          lds_f32x4_t sp = (lds_f32x4_t)(tab + 2 * CM + 2 * C4 + 32 * Pz + 8 * g);
          asm volatile("" : "+v"(sp));
          const f32x4_t sc0 = sp[0], sc1 = sp[1], sh0 = sp[COZ / 4], sh1 = sp[COZ / 4 + 1];
          u32x4_t ov;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            ov[i] = pack2h(__builtin_amdgcn_fmed3f(__builtin_fmaf(az[j][2 * Pz][2 * i], sc0[2 * i], sh0[2 * i]), 0.f, 65504.f),
                           __builtin_amdgcn_fmed3f(__builtin_fmaf(az[j][2 * Pz][2 * i + 1], sc0[2 * i + 1], sh0[2 * i + 1]), 0.f, 65504.f));
            ov[2 + i] = pack2h(__builtin_amdgcn_fmed3f(__builtin_fmaf(az[j][2 * Pz + 1][2 * i], sc1[2 * i], sh1[2 * i]), 0.f, 65504.f),
                               __builtin_amdgcn_fmed3f(__builtin_fmaf(az[j][2 * Pz + 1][2 * i + 1], sc1[2 * i + 1], sh1[2 * i + 1]), 0.f, 65504.f));
          }
          __builtin_amdgcn_raw_buffer_store_b128(ov, rz, zoff[NPZ ? j : 0] + Pz * 64, 0, 0);
        }
    }
  }
}

template <int CM, int HW, int R, int NPZ>
int launch_btail(const BneckArgs& a, hipStream_t s) {
  using K = BtCfg<CM, HW, R, NPZ>;
  static std::atomic<unsigned long long> attr;
  if (!spk_lds_limit_once(attr, (const void*)&conv_btail_kernel<CM, HW, R, NPZ>, 160 * 1024)) return -1;
  hipLaunchKernelGGL((conv_btail_kernel<CM, HW, R, NPZ>), dim3(a.N * (HW / R)), dim3(512), K::LDS, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int CM, int HW, int R, int NW>
int launch_bneck(const BneckArgs& a, hipStream_t s) {
  using K = BnCfg<CM, HW, R, NW>;
  static std::atomic<unsigned long long> attr;
  if (!spk_lds_limit_once(attr, (const void*)&conv_bneck_kernel<CM, HW, R, NW>, 160 * 1024)) return -1;
  hipLaunchKernelGGL((conv_bneck_kernel<CM, HW, R, NW>), dim3(a.N * (HW / R)), dim3(NW * 64), K::LDS, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// conv2 + conv3 + shortcut (+ the chained conv a.wz) from the block's conv1 output a.y1; -3: no kernel for this shape
int spk_btail_launch(const BneckArgs& a0, hipStream_t s) {
  static const int env_flags = getenv("SPK_BNECK_FLAGS") ? atoi(getenv("SPK_BNECK_FLAGS")) : 0;
  BneckArgs a = a0;
  a.flags |= env_flags;
  if (a.N <= 0 || a.H != a.W || a.C4 != 4 * a.CM || !a.y1) return -3;
  if ((size_t)a.N * a.H * a.W * a.C4 * 2 >= 0x80000000ull) return -3;
  if (a.CM == 64 && a.H == 56) {
    if (!a.wz) return launch_btail<64, 56, 4, 0>(a, s);
    if (a.Coutz == 64) return launch_btail<64, 56, 4, 2>(a, s);
    if (a.Coutz == 128) return launch_btail<64, 56, 4, 4>(a, s);
  }
  return -3;
}

// 0 ok, -1 HIP error, -3 no kernel for this shape (the caller runs the three convs one by one)
int spk_bneck_launch(const BneckArgs& a0, hipStream_t s) {
  static const int env_flags = getenv("SPK_BNECK_FLAGS") ? atoi(getenv("SPK_BNECK_FLAGS")) : 0;
  BneckArgs a = a0;
  a.flags |= env_flags;
  if (a.N <= 0 || a.H != a.W || a.C4 != 4 * a.CM) return -3;
  if ((size_t)a.N * a.H * a.W * a.C4 * 2 >= 0x80000000ull) return -3;
  // flags & 4: the small-block form - bands of 7 rows, 4 waves, <= 80 KB of LDS: two blocks per CU, whose HBM-bound phases
  // (1: x in, 3: shortcut in + out) then run beside each other's MFMA-bound phase 2 instead of all at once on every CU
  if (a.CM == 256 && a.H == 14)   // ResNet-50 stage 3: a block owns an image / half an image
    return (a.flags & 4) ? launch_bneck<256, 14, 7, 4>(a, s) : launch_bneck<256, 14, 14, 8>(a, s);
  if (a.CM == 128 && a.H == 28)   // stage 2: two bands of 14 rows / four of 7 per image
    return (a.flags & 4) ? launch_bneck<128, 28, 7, 4>(a, s) : launch_bneck<128, 28, 14, 8>(a, s);
  return -3;
}
