// Micro-benchmark: what bounds the implicit-GEMM main loop on gfx950?
// The conv kernel's K loop stripped to its skeleton (no addressing, no epilogue) in steps:
//   V0  MFMAs only (operands stay in registers)            -> sustained MFMA rate / clock
//   V1  + the per-K-step ds_read_b128 operand fragments     -> LDS read cost
//   V2  + LDS-DMA of the next tiles, vmcnt + one barrier    -> the real loop (flavour 1/3)
//   V3  V2 in the staggered two-barrier form                -> flavour 4
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_pipe mfma_pipe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int BM, int BN_, int WARPS_M, int WARPS_N, int V, int STAGES, int NBM>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void k(const unsigned char* src, unsigned window, int KT,
                                                           float* out) {
  constexpr int NW = WARPS_M * WARPS_N, NT_ = NW * 64;
  // NBM = 2: split weights, the B tile holds BN_ hi rows then BN_ lo rows and every MFMA is issued twice
  constexpr int BN = BN_ * NBM;
  constexpr int WM = BM / WARPS_M, WN = BN_ / WARPS_N, MT = WM / 16, NTA = WN / 16, NT = NTA * NBM;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int ROWS_PER_PASS = NT_ / 8;
  constexpr int PER_TILE = (BM + BN) / ROWS_PER_PASS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int frow = lane & 15, fq = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, window, 0x00020000);
  f32x4 acc[MT][NTA];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j % NTA] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (V >= 1) {  // defined LDS contents
    for (int i = tid * 16; i < STAGES * STAGE; i += NT_ * 16) *(u32x4*)(smem + i) = u32x4{0x3c003c00u, 0, 0, 0};
    __syncthreads();
  }
  int dma_stage = 0;
  unsigned goff = (blockIdx.x * 7919u * 4096u) % window;
  auto issue = [&]() {
    unsigned char* d = smem + dma_stage * STAGE + wave * (8 * 128);
#pragma unroll
    for (int i = 0; i < PER_TILE; ++i) {
      const unsigned off = (goff + (unsigned)(i * ROWS_PER_PASS * 128 + tid * 16)) % window;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(d + i * ROWS_PER_PASS * 128), 16, off, 0, 0, 0);
    }
    goff = (goff + STAGE) % window;
    dma_stage = dma_stage + 1 == STAGES ? 0 : dma_stage + 1;
  };
  unsigned char* piece_dst = nullptr;
  auto issue_begin = [&]() { piece_dst = smem + dma_stage * STAGE + wave * (8 * 128); };
  auto issue_piece = [&](int i) {
    const unsigned off = (goff + (unsigned)(i * ROWS_PER_PASS * 128 + tid * 16)) % window;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(piece_dst + i * ROWS_PER_PASS * 128), 16, off, 0, 0, 0);
  };
  auto issue_end = [&]() {
    goff = (goff + STAGE) % window;
    dma_stage = dma_stage + 1 == STAGES ? 0 : dma_stage + 1;
  };
  u32x4 ca = u32x4{0x3c003c00u, 0x3c003c00u, 0, 0}, cb = u32x4{0x3c003c00u, 0, 0, 0};

  if (V == 0) {
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j % NTA] = mma(ca, cb, acc[i][j % NTA]);
    }
  } else if (V == 1 || V == 2) {
    if (V == 2) {
#pragma unroll
      for (int t = 0; t < STAGES - 1; ++t) issue();
    }
    int cs = 0;
    for (int kt = 0; kt < KT; ++kt) {
      if (V == 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_TILE) : "memory");
        __builtin_amdgcn_s_barrier();
        issue();
      }
      const unsigned char* pa = smem + cs * STAGE;
      const unsigned char* pb = pa + BM * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fa[MT], fb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(pa + lds_off(wm * WM + i * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + j * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j % NTA] = mma(fa[i], fb[j], acc[i][j % NTA]);
      }
      cs = cs + 1 == STAGES ? 0 : cs + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (V == 12) {
    // V12: ONE stage per block (no intra-block prefetch): load, wait, multiply; the overlap comes from the
    // other blocks resident on the CU (a third of the LDS of the two-stage loop at the same tile)
    for (int kt = 0; kt < KT; ++kt) {
      __builtin_amdgcn_s_barrier();  // everyone is done reading the stage
      issue();
      dma_stage = 0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned char* pa = smem;
      const unsigned char* pb = pa + BM * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fa[MT], fb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(pa + lds_off(wm * WM + i * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + j * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j % NTA] = mma(fa[i], fb[j], acc[i][j % NTA]);
      }
    }
  } else if (V == 4 || V == 5 || V == 6) {
    // V4: all fragments up front, DMA pieces spread between the MFMA rows
    // V5: DMA + MFMA on constant operands (no ds_read); V6: ds_read + DMA, no MFMA
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t) issue();
    int cs = 0;
    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_TILE) : "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned char* pa = smem + cs * STAGE;
      const unsigned char* pb = pa + BM * 128;
      u32x4 fa[2][MT], fb[2][NT];
      if (V != 5) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < MT; ++i) fa[ks][i] = *(const u32x4*)(pa + lds_off(wm * WM + i * 16 + frow, ks * 4 + fq));
#pragma unroll
          for (int j = 0; j < NT; ++j) fb[ks][j] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + j * 16 + frow, ks * 4 + fq));
        }
      }
      issue_begin();
      int piece = 0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (V == 6) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j % NTA][0] += __builtin_bit_cast(float, fa[ks][i][0] ^ fb[ks][j][1]);
          } else {
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j % NTA] = V == 5 ? mma(ca, cb, acc[i][j % NTA]) : mma(fa[ks][i], fb[ks][j], acc[i][j % NTA]);
          }
          if (piece < PER_TILE) { issue_piece(piece); ++piece; }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (; piece < PER_TILE; ++piece) issue_piece(piece);
      issue_end();
      cs = cs + 1 == STAGES ? 0 : cs + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (V == 7 || V == 8) {
    // V7: A tile by LDS-DMA, B tile through registers (global_load -> ds_write after the MFMAs)
    // V8: both tiles through registers (the register-staged flavour)
    constexpr int A_IT = BM / ROWS_PER_PASS, B_IT = BN / ROWS_PER_PASS;
    const int srow = tid >> 3, chunk = tid & 7;
    u32x4 ra[A_IT], rb[B_IT];
    auto gload = [&]() {
      if (V == 8) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
          ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (goff + (unsigned)(i * ROWS_PER_PASS * 128 + tid * 16)) % window, 0, 0);
      } else {
        unsigned char* d = smem + dma_stage * STAGE + wave * (8 * 128);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const unsigned off = (goff + (unsigned)(i * ROWS_PER_PASS * 128 + tid * 16)) % window;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(d + i * ROWS_PER_PASS * 128), 16, off, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (goff + (unsigned)(BM * 128 + i * ROWS_PER_PASS * 128 + tid * 16)) % window, 0, 0);
      goff = (goff + STAGE) % window;
    };
    auto lstore = [&]() {
      unsigned char* st = smem + dma_stage * STAGE;
      if (V == 8) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *(u32x4*)(st + lds_off(srow + i * ROWS_PER_PASS, chunk)) = ra[i];
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) *(u32x4*)(st + BM * 128 + lds_off(srow + i * ROWS_PER_PASS, chunk)) = rb[i];
      dma_stage ^= 1;
    };

    gload();
    lstore();
    int cs = 0;
    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      gload();  // tile kt+1: DMA part lands in the other stage, register part is stored after the MFMAs
      const unsigned char* pa = smem + cs * STAGE;
      const unsigned char* pb = pa + BM * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fa[MT], fb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(pa + lds_off(wm * WM + i * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + j * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j % NTA] = mma(fa[i], fb[j], acc[i][j % NTA]);
      }
      lstore();
      cs ^= 1;
    }
  } else if (V == 9) {
    // V9: register path, two register sets (tile kt+1 landed, tile kt+2 in flight), the ds_writes of
    // tile kt+1 and the ks=1 fragment reads interleaved with the MFMA rows by hand
    constexpr int A_IT = BM / ROWS_PER_PASS, B_IT = BN / ROWS_PER_PASS, NP = A_IT + B_IT;
    constexpr int NG = 2 * MT;  // MFMA rows per K step
    const int srow = tid >> 3, chunk = tid & 7;
    u32x4 r0[NP], r1[NP];
    auto gload = [&](u32x4 (&r)[NP]) {
#pragma unroll
      for (int i = 0; i < NP; ++i)
        r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (goff + (unsigned)(i * ROWS_PER_PASS * 128 + tid * 16)) % window, 0, 0);
      goff = (goff + STAGE) % window;
    };
    auto lstore1 = [&](const u32x4 (&r)[NP], int i, int stage) {
      *(u32x4*)(smem + stage * STAGE + lds_off(srow + i * ROWS_PER_PASS, chunk)) = r[i];
    };
    auto step = [&](const u32x4 (&cur)[NP], u32x4 (&nxt)[NP], int cs) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      gload(nxt);
      const unsigned char* pa = smem + cs * STAGE;
      const unsigned char* pb = pa + BM * 128;
      u32x4 fa[2][MT], fb[2][NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[0][i] = *(const u32x4*)(pa + lds_off(wm * WM + i * 16 + frow, fq));
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[0][j] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + j * 16 + frow, fq));
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");  // `cur` has landed
      __builtin_amdgcn_sched_barrier(0);
      int piece = 0, rd = 0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ks = g / MT, i = g % MT;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j % NTA] = mma(fa[ks][i], fb[ks][j], acc[i][j % NTA]);
        // this group's share of the ks=1 fragment reads
        if (ks == 0) {
#pragma unroll
          for (int q = 0; q < (MT + NT + MT - 1) / MT; ++q, ++rd) {
            if (rd < MT) fa[1][rd] = *(const u32x4*)(pa + lds_off(wm * WM + rd * 16 + frow, 4 + fq));
            else if (rd < MT + NT) fb[1][rd - MT] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + (rd - MT) * 16 + frow, 4 + fq));
          }
        }
        // and of the stores of tile kt+1
#pragma unroll
        for (int q = 0; q < (NP + NG - 1) / NG; ++q, ++piece)
          if (piece < NP) lstore1(cur, piece, cs ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    gload(r0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NP; ++i) lstore1(r0, i, 0);
    gload(r0);
    for (int kt = 0; kt < KT; kt += 2) {
      step(r0, r1, 0);
      step(r1, r0, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (V == 11) {
    // V11: A tile through registers (two sets, stores interleaved with the MFMA rows), B tile by LDS-DMA
    constexpr int A_IT = BM / ROWS_PER_PASS, B_IT = BN / ROWS_PER_PASS;
    constexpr int NG = 2 * MT;
    const int srow = tid >> 3, chunk = tid & 7;
    u32x4 r0[A_IT], r1[A_IT];
    auto gloadA = [&](u32x4 (&r)[A_IT]) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (goff + (unsigned)(i * ROWS_PER_PASS * 128 + tid * 16)) % window, 0, 0);
    };
    auto dmaB = [&](int stage) {
      unsigned char* d = smem + stage * STAGE + BM * 128 + wave * (8 * 128);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const unsigned off = (goff + (unsigned)(BM * 128 + i * ROWS_PER_PASS * 128 + tid * 16)) % window;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(d + i * ROWS_PER_PASS * 128), 16, off, 0, 0, 0);
      }
    };
    auto lstore1 = [&](const u32x4 (&r)[A_IT], int i, int stage) {
      *(u32x4*)(smem + stage * STAGE + lds_off(srow + i * ROWS_PER_PASS, chunk)) = r[i];
    };
    auto step = [&](const u32x4 (&cur)[A_IT], u32x4 (&nxt)[A_IT], int cs) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      dmaB(cs ^ 1);      // B of tile kt+1
      goff = (goff + STAGE) % window;
      gloadA(nxt);       // A of tile kt+2
      const unsigned char* pa = smem + cs * STAGE;
      const unsigned char* pb = pa + BM * 128;
      u32x4 fa[2][MT], fb[2][NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[0][i] = *(const u32x4*)(pa + lds_off(wm * WM + i * 16 + frow, fq));
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[0][j] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + j * 16 + frow, fq));
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT + B_IT) : "memory");  // `cur` (A of tile kt+1) has landed
      __builtin_amdgcn_sched_barrier(0);
      int piece = 0, rd = 0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ks = g / MT, i = g % MT;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j % NTA] = mma(fa[ks][i], fb[ks][j], acc[i][j % NTA]);
        if (ks == 0) {
#pragma unroll
          for (int q = 0; q < (MT + NT + MT - 1) / MT; ++q, ++rd) {
            if (rd < MT) fa[1][rd] = *(const u32x4*)(pa + lds_off(wm * WM + rd * 16 + frow, 4 + fq));
            else if (rd < MT + NT) fb[1][rd - MT] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + (rd - MT) * 16 + frow, 4 + fq));
          }
        }
#pragma unroll
        for (int q = 0; q < (A_IT + NG - 1) / NG; ++q, ++piece)
          if (piece < A_IT) lstore1(cur, piece, cs ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT) : "memory");  // B of tile kt+1 has landed
    };
    gloadA(r0);
    dmaB(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < A_IT; ++i) lstore1(r0, i, 0);
    goff = (goff + STAGE) % window;
    gloadA(r0);
    for (int kt = 0; kt < KT; kt += 2) {
      step(r0, r1, 0);
      step(r1, r0, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {  // V == 3: staggered
    const bool g1 = wave >= NW / 2;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t) issue();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_TILE) : "memory");
    __builtin_amdgcn_s_barrier();
    if (g1) __builtin_amdgcn_s_barrier();
    int cs = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const unsigned char* pa = smem + cs * STAGE;
      const unsigned char* pb = pa + BM * 128;
      u32x4 fa[2][MT], fb[2][NT];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[ks][i] = *(const u32x4*)(pa + lds_off(wm * WM + i * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[ks][j] = *(const u32x4*)(pb + lds_off(wn * WN * NBM + j * 16 + frow, ks * 4 + fq));
      }
      issue();
      if (g1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_TILE) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j % NTA] = mma(fa[ks][i], fb[ks][j], acc[i][j % NTA]);
      __builtin_amdgcn_s_setprio(0);
      if (!g1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_TILE) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      cs = cs + 1 == STAGES ? 0 : cs + 1;
    }
    if (!g1) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) t += acc[i][j % NTA][0] + acc[i][j % NTA][1] + acc[i][j % NTA][2] + acc[i][j % NTA][3];
  if (t == 12345.678f) out[blockIdx.x * NT_ + tid] = t;
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int V, int STAGES, int NBM = 1>
void run(const char* name, const unsigned char* src, unsigned window, float* out, int blocks, int KT) {
  auto kern = k<BM, BN, WARPS_M, WARPS_N, V, STAGES, NBM>;
  const int lds = V == 0 ? 0 : STAGES * (BM + BN * NBM) * 128;
  if (lds > 163840) return;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int threads = WARPS_M * WARPS_N * 64;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, src, window, KT, out);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, src, window, KT, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double fl = (double)blocks * KT * BM * BN * NBM * 64 * 2;
  printf("%-34s %dx%dx%d w%dx%d V%d S%d blocks %5d KT %3d lds %6d: %8.1f us  %7.1f TFLOP/s  (%s)\n", name, BM, BN, NBM,
         WARPS_M, WARPS_N, V, STAGES, blocks, KT, lds, ms * 1000, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

// V2 with 32-deep K steps (64-B LDS rows): half the stage, twice the resident blocks
template <int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void k32(const unsigned char* src, unsigned window, int KT,
                                                             float* out) {
  constexpr int NW = WARPS_M * WARPS_N, NT_ = NW * 64;
  constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N, MT = WM / 16, NT = WN / 16;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int ROWS_PER_PASS = NT_ / 4;  // 4 chunks per 64-B row
  constexpr int PER_TILE = (BM + BN) / ROWS_PER_PASS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int frow = lane & 15, fq = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, window, 0x00020000);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid * 16; i < 2 * STAGE; i += NT_ * 16) *(u32x4*)(smem + i) = u32x4{0x3c003c00u, 0, 0, 0};
  __syncthreads();
  int dma_stage = 0;
  unsigned goff = (blockIdx.x * 7919u * 4096u) % window;
  auto issue = [&]() {
    unsigned char* d = smem + dma_stage * STAGE + wave * (16 * 64);
#pragma unroll
    for (int i = 0; i < PER_TILE; ++i) {
      const unsigned off = (goff + (unsigned)(i * ROWS_PER_PASS * 64 + tid * 16)) % window;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(d + i * ROWS_PER_PASS * 64), 16, off, 0, 0, 0);
    }
    goff = (goff + STAGE) % window;
    dma_stage ^= 1;
  };
  auto off32 = [&](int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); };
  issue();
  int cs = 0;
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue();
    const unsigned char* pa = smem + cs * STAGE;
    const unsigned char* pb = pa + BM * 64;
    u32x4 fa[MT], fb[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(pa + off32(wm * WM + i * 16 + frow, fq));
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = *(const u32x4*)(pb + off32(wn * WN + j * 16 + frow, fq));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = mma(fa[i], fb[j], acc[i][j]);
    cs ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (t == 12345.678f) out[blockIdx.x * NT_ + tid] = t;
}

template <int BM, int BN, int WARPS_M, int WARPS_N>
void run32(const char* name, const unsigned char* src, unsigned window, float* out, int blocks, int KT) {
  auto kern = k32<BM, BN, WARPS_M, WARPS_N>;
  const int lds = 2 * (BM + BN) * 64;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int threads = WARPS_M * WARPS_N * 64;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, src, window, KT, out);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, src, window, KT, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double fl = (double)blocks * KT * BM * BN * 32 * 2;
  printf("%-34s %dx%d BK32 w%dx%d blocks %5d KT %3d lds %6d: %8.1f us  %7.1f TFLOP/s  (%s)\n", name, BM, BN, WARPS_M,
         WARPS_N, blocks, KT, lds, ms * 1000, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const unsigned big = 512u << 20, small = 8u << 20;
  unsigned char* src;
  float* out;
  hipMalloc(&src, big);
  hipMemset(src, 0, big);
  hipMalloc(&out, 64 << 20);
  const int KT = 36;
  for (int pass = 0; pass < 1; ++pass) {
    const unsigned win = argc > 1 ? (unsigned)atoi(argv[1]) << 20 : small;
    printf("---- DMA window %u MiB, KT %d ----\n", win >> 20, KT);
    if (argc > 2 && argv[2][0] == '1') {  // one stage per block, more blocks per CU
      run<256, 128, 4, 2, 2, 2>("2 stages", src, win, out, 2048, KT);
      run<256, 128, 4, 2, 12, 1>("1 stage", src, win, out, 2048, KT);
      run<128, 128, 2, 2, 2, 2>("2 stages", src, win, out, 8192, KT);
      run<128, 128, 2, 2, 12, 1>("1 stage", src, win, out, 8192, KT);
      run<256, 256, 2, 4, 2, 2>("2 stages", src, win, out, 1024, KT);
      run<256, 256, 2, 4, 12, 1>("1 stage", src, win, out, 1024, KT);
      run<128, 64, 2, 2, 12, 1>("1 stage", src, win, out, 8192, KT);
      run<256, 128, 4, 2, 2, 2, 2>("split 2 stages", src, win, out, 2048, KT);
      run<256, 128, 4, 2, 12, 1, 2>("split 1 stage", src, win, out, 2048, KT);
      run<128, 128, 2, 2, 12, 1, 2>("split 1 stage", src, win, out, 4096, KT);
      run<128, 64, 2, 2, 12, 1, 2>("split 1 stage", src, win, out, 8192, KT);
    } else if (argc > 2 && argv[2][0] == 'k') {  // 32-deep K steps
      run<128, 128, 2, 2, 2, 2>("BK64: dma 2 stages", src, win, out, 8192, KT);
      run32<128, 128, 2, 2>("BK32: dma 2 stages", src, win, out, 8192, 2 * KT);
      run<128, 64, 2, 2, 2, 2>("BK64: dma 2 stages", src, win, out, 8192, KT);
      run32<128, 64, 2, 2>("BK32: dma 2 stages", src, win, out, 8192, 2 * KT);
      run<64, 64, 2, 2, 2, 2>("BK64: dma 2 stages", src, win, out, 16384, KT);
      run32<64, 64, 2, 2>("BK32: dma 2 stages", src, win, out, 16384, 2 * KT);
      run<256, 128, 4, 2, 2, 2>("BK64: dma 2 stages", src, win, out, 2048, KT);
      run32<256, 128, 4, 2>("BK32: dma 2 stages", src, win, out, 2048, 2 * KT);
      run32<256, 256, 2, 4>("BK32: dma 2 stages", src, win, out, 1024, 2 * KT);
    } else if (argc > 2 && argv[2][0] == '3') {  // 3x3 halo slab: one activation tile feeds 3 taps (B tile and MFMAs x3)
      run<128, 128, 2, 2, 2, 2, 1>("now: dma 2 stages", src, win, out, 8192, KT);
      run<128, 128, 2, 2, 2, 2, 3>("slab: dma 2 stages", src, win, out, 2048, KT);
      run<128, 128, 2, 2, 4, 2, 3>("slab: dma spread", src, win, out, 2048, KT);
      run<128, 64, 2, 2, 2, 2, 1>("now: dma 2 stages", src, win, out, 8192, KT);
      run<128, 64, 2, 2, 2, 2, 3>("slab: dma 2 stages", src, win, out, 4096, KT);
      run<128, 64, 2, 2, 4, 2, 3>("slab: dma spread", src, win, out, 4096, KT);
      run<256, 64, 4, 2, 2, 2, 1>("now: dma 2 stages", src, win, out, 4096, KT);
      run<256, 64, 4, 2, 2, 2, 3>("slab: dma 2 stages", src, win, out, 2048, KT);
      run<256, 64, 4, 2, 4, 2, 3>("slab: dma spread", src, win, out, 2048, KT);
      run<256, 128, 4, 2, 2, 2, 1>("now: dma 2 stages", src, win, out, 2048, KT);
      run<64, 64, 2, 2, 2, 2, 1>("now: dma 2 stages", src, win, out, 16384, KT);
      run<64, 64, 2, 2, 2, 2, 3>("slab: dma 2 stages", src, win, out, 8192, KT);
    } else if (argc > 2) {  // split-weight (hi+lo) shapes
      run<128, 64, 2, 2, 1, 2, 2>("+ds_read", src, win, out, 8192, KT);
      run<128, 64, 2, 2, 2, 2, 2>("dma 2 stages", src, win, out, 8192, KT);
      run<128, 64, 2, 2, 4, 2, 2>("dma spread", src, win, out, 8192, KT);
      run<128, 64, 2, 2, 9, 2, 2>("registers interleaved", src, win, out, 8192, KT);
      run<128, 64, 2, 2, 11, 2, 2>("A regs interleaved, B dma", src, win, out, 8192, KT);
      run<128, 128, 2, 2, 1, 2, 2>("+ds_read", src, win, out, 4096, KT);
      run<128, 128, 2, 2, 2, 2, 2>("dma 2 stages", src, win, out, 4096, KT);
      run<128, 128, 2, 2, 4, 2, 2>("dma spread", src, win, out, 4096, KT);
      run<128, 128, 2, 2, 9, 2, 2>("registers interleaved", src, win, out, 4096, KT);
      run<128, 128, 2, 2, 11, 2, 2>("A regs interleaved, B dma", src, win, out, 4096, KT);
      run<256, 128, 4, 2, 1, 2, 2>("+ds_read", src, win, out, 2048, KT);
      run<256, 128, 4, 2, 2, 2, 2>("dma 2 stages", src, win, out, 2048, KT);
      run<256, 128, 4, 2, 4, 2, 2>("dma spread", src, win, out, 2048, KT);
      run<256, 128, 4, 2, 11, 2, 2>("A regs interleaved, B dma", src, win, out, 2048, KT);
      run<256, 64, 4, 2, 2, 2, 2>("dma 2 stages", src, win, out, 4096, KT);
      run<256, 64, 4, 2, 4, 2, 2>("dma spread", src, win, out, 4096, KT);
      run<256, 64, 4, 2, 9, 2, 2>("registers interleaved", src, win, out, 4096, KT);
      run<256, 64, 4, 2, 11, 2, 2>("A regs interleaved, B dma", src, win, out, 4096, KT);
    } else {
      run<128, 128, 2, 2, 2, 2>("dma 2 stages", src, win, out, 8192, KT);
      run<128, 128, 2, 2, 9, 2>("registers interleaved", src, win, out, 8192, KT);
      run<128, 128, 2, 2, 11, 2>("A regs interleaved, B dma", src, win, out, 8192, KT);
      run<256, 128, 4, 2, 2, 2>("dma 2 stages", src, win, out, 2048, KT);
      run<256, 128, 4, 2, 9, 2>("registers interleaved", src, win, out, 2048, KT);
      run<256, 128, 4, 2, 11, 2>("A regs interleaved, B dma", src, win, out, 2048, KT);
      run<256, 256, 2, 4, 4, 2>("dma spread", src, win, out, 1024, KT);
      run<256, 256, 2, 4, 11, 2>("A regs interleaved, B dma", src, win, out, 1024, KT);
      run<128, 64, 2, 2, 2, 2>("dma 2 stages", src, win, out, 8192, KT);
      run<128, 64, 2, 2, 9, 2>("registers interleaved", src, win, out, 8192, KT);
      run<128, 64, 2, 2, 11, 2>("A regs interleaved, B dma", src, win, out, 8192, KT);
      run<64, 64, 2, 2, 2, 2>("dma 2 stages", src, win, out, 16384, KT);
      run<64, 64, 2, 2, 11, 2>("A regs interleaved, B dma", src, win, out, 16384, KT);
    }
  }
  return 0;
}
