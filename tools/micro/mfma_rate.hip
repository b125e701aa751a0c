// MFMA issue-rate probe (gfx950): how fast does a wave-pair per SIMD retire v_mfma_f32_16x16x32_f16 streams shaped like
// the conv_pw K loop?  Variants: number of accumulators, reuse distance of an accumulator, operand register variety.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
// NACC accumulators, each hit REP times in a row-robin sweep per iteration (REP=2: the hi then the lo product), NOPND
// distinct A operand register sets, NX distinct B operand sets.
template <int NACC, int REP, int NOPND, int NX>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, unsigned seed) {
  f32x4 acc[NACC];
  u32x4 wa[NOPND], xb[NX];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NOPND; ++i) wa[i] = u32x4{seed + i + threadIdx.x, 0x3c003c00u, seed * 3 + i, 0x3c003c00u};
#pragma unroll
  for (int i = 0; i < NX; ++i) xb[i] = u32x4{0x3c003c00u, seed + 7 * i, 0x3c003c00u, seed ^ (i + threadIdx.x)};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = mma(wa[(i + r * NACC) % NOPND], xb[i % NX], acc[i]);
#pragma unroll
    for (int i = 0; i < NOPND; ++i) asm volatile("" : "+v"(wa[i]));   // operands opaque: no hoisting / merging
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] == 123.456f) out[threadIdx.x] = s[1];
}
template <int NACC, int REP, int NOPND, int NX>
void run(const char* name, int threads, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, grid = 256 * (512 / threads);   // 8 waves per CU in every case
  hipLaunchKernelGGL((k<NACC, REP, NOPND, NX>), dim3(grid), dim3(threads), 0, 0, out, 10, 1u);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, REP, NOPND, NX>), dim3(grid), dim3(threads), 0, 0, out, iters, 1u);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * (threads / 64) * iters * NACC * REP * 16384.0;
  printf("%-44s acc %2d rep %d opnd %2d x %d: %7.0f TFLOP/s\n", name, NACC, REP, NOPND, NX, flop / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 4096);
  run<16, 1, 1, 1>("16 acc, one operand pair", 512, out);
  run<16, 2, 1, 1>("16 acc x2 (hi,lo), one operand pair", 512, out);
  run<16, 2, 4, 2>("16 acc x2, 4 A sets, 2 B sets", 512, out);
  run<16, 2, 8, 2>("16 acc x2, 8 A sets, 2 B sets (conv_pw pair)", 512, out);
  run<8, 2, 8, 2>("8 acc x2", 512, out);
  run<4, 2, 4, 2>("4 acc x2", 512, out);
  run<32, 2, 8, 4>("32 acc x2, 8 A sets, 4 B sets", 512, out);
  run<32, 1, 8, 4>("32 acc", 512, out);
  run<16, 2, 8, 2>("same, 4-wave blocks (2 per CU)", 256, out);
  return 0;
}
