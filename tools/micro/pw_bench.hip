// A/B bench + check of the 1x1-conv kernel (csrc/conv_pw.hip) against the implicit-GEMM kernel (csrc/conv_igemm.hip,
// autotuned: its best tile x flavour) on the 1x1 convolutions of ResNet-50 at batch 256, fp16 eval with hi+lo weights.
// Build (tools/micro/build_pw_bench.sh):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../syke-pic_amd/csrc
//     pw_bench.hip ../../syke-pic_amd/csrc/{conv_pw,conv_igemm,conv_stem,pointwise}.hip -o pw_bench
// Run on the GPU box:  ./pw_bench [batch] [layer-filter]
#include "spk_common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int hw, cin, cout, stride, res, relu, count; };
// ResNet-50 @224^2: every 1x1 conv (count = how many layers of that shape the network has)
static const Shape kShapes[] = {
    {"s1.conv1a 64->64", 56, 64, 64, 1, 0, 1, 1},
    {"s1.conv1 256->64", 56, 256, 64, 1, 0, 1, 2},
    {"s1.conv3 64->256+res", 56, 64, 256, 1, 1, 1, 3},
    {"s1.ds 64->256", 56, 64, 256, 1, 0, 0, 1},
    {"s2.conv1a 256->128@56", 56, 256, 128, 1, 0, 1, 1},
    {"s2.conv1 512->128", 28, 512, 128, 1, 0, 1, 3},
    {"s2.conv3 128->512+res", 28, 128, 512, 1, 1, 1, 4},
    {"s2.ds 256->512/2", 56, 256, 512, 2, 0, 0, 1},
    {"s3.conv1a 512->256@28", 28, 512, 256, 1, 0, 1, 1},
    {"s3.conv1 1024->256", 14, 1024, 256, 1, 0, 1, 5},
    {"s3.conv3 256->1024+res", 14, 256, 1024, 1, 1, 1, 6},
    {"s3.ds 512->1024/2", 28, 512, 1024, 2, 0, 0, 1},
    {"s4.conv1a 1024->512@14", 14, 1024, 512, 1, 0, 1, 1},
    {"s4.conv1 2048->512", 7, 2048, 512, 1, 0, 1, 2},
    {"s4.conv3 512->2048+res", 7, 512, 2048, 1, 1, 1, 3},
    {"s4.ds 1024->2048/2", 14, 1024, 2048, 2, 0, 0, 1},
};

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static inline float urand() {  // [0,1)
  g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
  return (float)((g_rng >> 40) & 0xFFFFFF) / 16777216.f;
}


// ---- streaming-rate probes: what HBM gives a pure read / pure write / copy stream of 16-byte lanes ----
__global__ void bw_read_kernel(const u32x4_t* __restrict__ p, size_t n, unsigned* out) {
  u32x4_t acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4_t v = p[i];
    acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *out = 1;
}
__global__ void bw_write_kernel(u32x4_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = u32x4_t{(unsigned)i, 1, 2, 3};
}
__global__ void bw_write_nt_kernel(u32x4_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(u32x4_t{(unsigned)i, 1, 2, 3}, p + i);
}
__global__ void bw_write_dword_kernel(unsigned* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
// each block owns a contiguous 64 KB chunk at a time (instead of the grid-wide interleave)
__global__ void bw_write_chunk_kernel(u32x4_t* __restrict__ p, size_t n) {
  const size_t chunk = 4096;  // 16-byte elements = 64 KB
  for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x)
    for (size_t i = threadIdx.x; i < chunk; i += blockDim.x) p[c * chunk + i] = u32x4_t{(unsigned)i, 1, 2, 3};
}
__global__ void bw_copy_nt_kernel(const u32x4_t* __restrict__ a, u32x4_t* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
__global__ void bw_copy_kernel(const u32x4_t* __restrict__ a, u32x4_t* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// 1 read : 4 write, the shape of a 64 -> 256 channel 1x1 conv: 64-byte store segments per lane quad
__global__ void bw_r1w4_kernel(const u32x4_t* __restrict__ a, u32x4_t* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4_t v = a[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) b[i * 4 + k] = v;
  }
}
// conv-output write shapes: rows of ROWB bytes; a wave owns 32 consecutive rows and writes them with 16-byte lanes,
// SEG contiguous bytes per row and instruction (64: what the register epilogue of conv_pw does; ROWB: whole rows)
template <int ROWB, int SEG>
__global__ void bw_rows_kernel(unsigned char* __restrict__ p, size_t rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  constexpr int LPR = SEG / 16, RPI = 64 / LPR, SEGS = ROWB / SEG;
  for (size_t r0 = ((size_t)blockIdx.x * waves + wave) * 32; r0 + 32 <= rows; r0 += (size_t)gridDim.x * waves * 32) {
#pragma unroll
    for (int rb = 0; rb < 32; rb += RPI)
#pragma unroll
      for (int sg = 0; sg < SEGS; ++sg) {
        const size_t off = (r0 + rb + lane / LPR) * ROWB + sg * SEG + (lane % LPR) * 16;
        *(u32x4_t*)(p + off) = u32x4_t{(unsigned)off, 1, 2, 3};
      }
  }
}
static void bw_probe(hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  u32x4_t *a, *b; unsigned* flag;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&flag, 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  auto timeit = [&](const char* name, double moved, auto launch) {
    launch();
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("  stream probe %-12s %.2f TB/s\n", name, moved / ms / 1e9);
  };
  for (int grid : {512, 2048, 16384}) {
    printf(" grid %d x 256:\n", grid);
    timeit("read", (double)bytes, [&] { hipLaunchKernelGGL(bw_read_kernel, dim3(grid), dim3(256), 0, st, a, n, flag); });
    timeit("write", (double)bytes, [&] { hipLaunchKernelGGL(bw_write_kernel, dim3(grid), dim3(256), 0, st, b, n); });
    timeit("write nt", (double)bytes, [&] { hipLaunchKernelGGL(bw_write_nt_kernel, dim3(grid), dim3(256), 0, st, b, n); });
    timeit("write dword", (double)bytes, [&] { hipLaunchKernelGGL(bw_write_dword_kernel, dim3(grid), dim3(256), 0, st, (unsigned*)b, n * 4); });
    timeit("write chunk", (double)bytes, [&] { hipLaunchKernelGGL(bw_write_chunk_kernel, dim3(grid), dim3(256), 0, st, b, n); });
    timeit("copy nt", 2.0 * bytes, [&] { hipLaunchKernelGGL(bw_copy_nt_kernel, dim3(grid), dim3(256), 0, st, a, b, n); });
    timeit("copy", 2.0 * bytes, [&] { hipLaunchKernelGGL(bw_copy_kernel, dim3(grid), dim3(256), 0, st, a, b, n); });
    timeit("rows512 s64", (double)bytes, [&] { hipLaunchKernelGGL((bw_rows_kernel<512, 64>), dim3(grid), dim3(256), 0, st, (unsigned char*)b, bytes / 512); });
    timeit("rows512 s128", (double)bytes, [&] { hipLaunchKernelGGL((bw_rows_kernel<512, 128>), dim3(grid), dim3(256), 0, st, (unsigned char*)b, bytes / 512); });
    timeit("rows512 s256", (double)bytes, [&] { hipLaunchKernelGGL((bw_rows_kernel<512, 256>), dim3(grid), dim3(256), 0, st, (unsigned char*)b, bytes / 512); });
    timeit("rows512 s512", (double)bytes, [&] { hipLaunchKernelGGL((bw_rows_kernel<512, 512>), dim3(grid), dim3(256), 0, st, (unsigned char*)b, bytes / 512); });
    timeit("rows2048 s64", (double)bytes, [&] { hipLaunchKernelGGL((bw_rows_kernel<2048, 64>), dim3(grid), dim3(256), 0, st, (unsigned char*)b, bytes / 2048); });
    timeit("rows2048 s256", (double)bytes, [&] { hipLaunchKernelGGL((bw_rows_kernel<2048, 256>), dim3(grid), dim3(256), 0, st, (unsigned char*)b, bytes / 2048); });
    timeit("rows2048 s1024", (double)bytes, [&] { hipLaunchKernelGGL((bw_rows_kernel<2048, 1024>), dim3(grid), dim3(256), 0, st, (unsigned char*)b, bytes / 2048); });
    timeit("1r:4w", 1.25 * bytes, [&] { hipLaunchKernelGGL(bw_r1w4_kernel, dim3(grid), dim3(256), 0, st, a, b, n / 4); });
  }
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(flag));
}


// ---- 3x3 convs: conv_c3.hip against the implicit GEMM ----
static int bench_c3(int batch, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  struct S3 { const char* name; int hw, c; int count; };
  const S3 shapes[] = {{"s1.conv2 64@56", 56, 64, 3}, {"s2.conv2 128@28", 28, 128, 3}, {"s3.conv2 256@14", 14, 256, 5}, {"s4.conv2 512@7", 7, 512, 2}};
  const int nb = getenv("PW_NB") ? atoi(getenv("PW_NB")) : 1;
  const char* filt = getenv("C3_FILT");
  double tot_old = 0, tot_new = 0;
  for (const S3& sh : shapes) {
    if (filt && !strstr(sh.name, filt)) continue;
    const int H = sh.hw, W = sh.hw, C = sh.c;
    const size_t M = (size_t)batch * H * W, nx = M * C, nw = (size_t)C * 9 * C;
    std::vector<unsigned short> hx(nx);
    std::vector<float> hw(nw), hs(C), hb(C);
    for (auto& v : hx) { float f = urand() * 2.f - 0.7f; f = f > 0 ? f : 0.f; v = __builtin_bit_cast(unsigned short, (_Float16)f); }
    const float bound = sqrtf(6.f / (9 * C));
    for (auto& v : hw) v = (urand() * 2.f - 1.f) * bound;
    for (auto& v : hs) v = 0.5f + urand();
    for (auto& v : hb) v = urand() - 0.5f;
    bf16_t *dx, *dy0, *dy1, *wp_old, *wp_new;
    float *dw, *dsc, *dbi;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dy0, nx * 2)); CK(hipMalloc(&dy1, nx * 2));
    CK(hipMalloc(&wp_old, nw * 4)); CK(hipMalloc(&wp_new, nw * 4));
    CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dsc, C * 4)); CK(hipMalloc(&dbi, C * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, hs.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbi, hb.data(), C * 4, hipMemcpyHostToDevice));
    if (spk_launch_pack_weights(dw, wp_old, C, 3, 3, C, CONV_MODE_GENERIC, DT_F16, nb == 2, st)) return 2;
    if (spk_launch_pack_c3(dw, wp_new, C, C, nb, st)) return 2;
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.cfg = a.dma = -1; a.cls_ph = a.cls_pw = -1;
    a.x = dx; a.w = wp_old; a.y = dy0; a.scale = dsc; a.bias = dbi;
    a.N = batch; a.H = H; a.W = W; a.Cin = C; a.Ho = H; a.Wo = W; a.Cout = C;
    a.kh = a.kw = 3; a.stride = 1; a.pad = 1; a.M = (int)M; a.K = 9 * C; a.relu = 1; a.dt = DT_F16; a.splitw = nb == 2;
    a.x_bytes = (unsigned)(nx * 2); a.w_bytes = (unsigned)(nw * 2 * nb);
    if (spk_conv_launch(a, CONV_MODE_GENERIC, st, nullptr)) { fprintf(stderr, "igemm launch failed\n"); return 3; }
    CK(hipStreamSynchronize(st));
    const int reps = 10;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) spk_conv_launch(a, CONV_MODE_GENERIC, st, nullptr);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms_old; CK(hipEventElapsedTime(&ms_old, e0, e1)); ms_old /= reps;
    std::vector<unsigned short> y0(nx), y1(nx);
    CK(hipMemcpy(y0.data(), dy0, nx * 2, hipMemcpyDeviceToHost));
    C3Args q;
    memset(&q, 0, sizeof q);
    q.x = dx; q.wp = wp_new; q.y = dy1; q.scale = dsc; q.shift = dbi;
    q.N = batch; q.H = H; q.W = W; q.Cin = C; q.Cout = C; q.M = (int)M; q.relu = 1; q.dt = DT_F16; q.nb = nb;
    q.x_bytes = (unsigned)(nx * 2); q.y_bytes = (unsigned)(nx * 2); q.wp_bytes = (unsigned)(nw * 2 * nb);
    const double flop = 2.0 * M * 9 * C * C;
    printf("%-20s M=%zu  igemm %.1f us (%.0f TF)\n", sh.name, M, ms_old * 1e3, flop / ms_old / 1e9);
    float best = 1e30f; int best_cfg = -1;
    for (int cfg = 0; cfg < spk_c3_num_configs(); ++cfg) {
      CK(hipMemsetAsync(dy1, 0xee, nx * 2, st));
      const int r0 = spk_c3_launch(q, cfg, st);
      if (r0 == -3) continue;
      if (r0) { fprintf(stderr, "c3 launch cfg %d failed: %d\n", cfg, r0); return 4; }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(y1.data(), dy1, nx * 2, hipMemcpyDeviceToHost));
      double maxd = 0; size_t bad = 0;
      for (size_t i = 0; i < nx; ++i) {
        const float u = (float)__builtin_bit_cast(_Float16, y0[i]), v = (float)__builtin_bit_cast(_Float16, y1[i]);
        const double d = fabs((double)u - v);
        if (!(d <= 4e-3 * (1.0 + fabs(u)))) { if (bad < 8) printf("      bad px %zu cout %zu: want %g got %g\n", i / C, i % C, u, v); ++bad; }
        if (d > maxd) maxd = d;
      }
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) spk_c3_launch(q, cfg, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("    c3 cfg %d: %7.1f us (%4.0f TF)  max|d| %.2e  bad %zu%s\n", cfg, ms * 1e3, flop / ms / 1e9, maxd, bad, bad ? "  <-- MISMATCH" : "");
      if (!bad && ms < best) { best = ms; best_cfg = cfg; }
    }
    printf("    => best c3 cfg %d: %.1f us vs igemm %.1f us (x%.2f)\n", best_cfg, best * 1e3, ms_old * 1e3, ms_old / best);
    fflush(stdout);
    tot_old += ms_old * sh.count; tot_new += (best < ms_old ? best : ms_old) * sh.count;
    hipFree(dx); hipFree(dy0); hipFree(dy1); hipFree(wp_old); hipFree(wp_new); hipFree(dw); hipFree(dsc); hipFree(dbi);
  }
  printf("TOTAL over the network's stride-1 3x3 convs: igemm %.3f ms -> best-of %.3f ms\n", tot_old, tot_new);
  return 0;
}

int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 256;
  const char* filt = argc > 2 ? argv[2] : nullptr;
  const int nb = getenv("PW_NB") ? atoi(getenv("PW_NB")) : 2;
  const int only_cfg = getenv("PW_CFG") ? atoi(getenv("PW_CFG")) : -1;
  const int dump = getenv("PW_DUMP") ? atoi(getenv("PW_DUMP")) : 0;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (getenv("PW_BW")) bw_probe(st, e0, e1);
  if (getenv("C3")) return bench_c3(batch, st, e0, e1);
  double tot_old = 0, tot_new = 0;
  for (const Shape& sh : kShapes) {
    if (filt && !strstr(sh.name, filt)) continue;
    const int H = sh.hw, W = sh.hw, Ho = H / sh.stride, Wo = W / sh.stride;
    const size_t in_px = (size_t)batch * H * W, M = (size_t)batch * Ho * Wo;
    const size_t nx = in_px * sh.cin, ny = M * sh.cout, nw = (size_t)sh.cout * sh.cin;
    std::vector<unsigned short> hx(nx), hr(ny);
    std::vector<float> hw(nw), hs(sh.cout), hb(sh.cout);
    for (auto& v : hx) { float f = urand() * 2.f - 0.7f; f = f > 0 ? f : 0.f; v = __builtin_bit_cast(unsigned short, (_Float16)f); }
    for (auto& v : hr) { float f = urand() * 2.f - 0.7f; f = f > 0 ? f : 0.f; v = __builtin_bit_cast(unsigned short, (_Float16)f); }
    const float bound = sqrtf(6.f / sh.cin);
    for (auto& v : hw) v = (urand() * 2.f - 1.f) * bound;
    for (auto& v : hs) v = 0.5f + urand();
    for (auto& v : hb) v = urand() - 0.5f;
    bf16_t *dx, *dr, *dy0, *dy1, *wp_old, *wp_new;
    float *dw, *dsc, *dbi;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dr, ny * 2)); CK(hipMalloc(&dy0, ny * 2)); CK(hipMalloc(&dy1, ny * 2));
    CK(hipMalloc(&wp_old, nw * 2 * 2)); CK(hipMalloc(&wp_new, nw * 2 * 2));
    CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dsc, sh.cout * 4)); CK(hipMalloc(&dbi, sh.cout * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, hr.data(), ny * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, hs.data(), sh.cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbi, hb.data(), sh.cout * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dy0, 0xff, ny * 2)); CK(hipMemset(dy1, 0xee, ny * 2));
    if (spk_launch_pack_weights(dw, wp_old, sh.cout, 1, 1, sh.cin, CONV_MODE_GENERIC, DT_F16, nb == 2, st)) return 2;
    if (spk_launch_pack_pw(dw, nullptr, wp_new, sh.cout, sh.cin, DT_F16, nb, st)) return 2;

    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.cfg = a.dma = -1; a.cls_ph = a.cls_pw = -1;
    a.x = dx; a.w = wp_old; a.y = dy0; a.res = sh.res ? dr : nullptr;
    a.scale = dsc; a.bias = dbi;
    a.N = batch; a.H = H; a.W = W; a.Cin = sh.cin; a.Ho = Ho; a.Wo = Wo; a.Cout = sh.cout;
    a.kh = a.kw = 1; a.stride = sh.stride; a.pad = 0; a.M = (int)M; a.K = sh.cin; a.relu = sh.relu;
    a.dt = DT_F16; a.splitw = nb == 2;
    a.x_bytes = (unsigned)(nx * 2); a.w_bytes = (unsigned)(nw * 2 * (nb == 2 ? 2 : 1));
    if (spk_conv_launch(a, CONV_MODE_GENERIC, st, nullptr)) { fprintf(stderr, "igemm launch failed\n"); return 3; }  // tunes
    CK(hipStreamSynchronize(st));
    const int reps = 10;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) spk_conv_launch(a, CONV_MODE_GENERIC, st, nullptr);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms_old; CK(hipEventElapsedTime(&ms_old, e0, e1)); ms_old /= reps;

    PwConvArgs q;
    memset(&q, 0, sizeof q);
    q.x = dx; q.wp = wp_new; q.y = dy1; q.res = sh.res ? dr : nullptr; q.scale = dsc; q.shift = dbi;
    q.N = batch; q.H = H; q.W = W; q.Ho = Ho; q.Wo = Wo; q.stride = sh.stride; q.Cin = sh.cin; q.Cout = sh.cout;
    q.M = (int)M; q.relu = sh.relu; q.dt = DT_F16; q.nb = nb;
    q.x_bytes = (unsigned)(nx * 2); q.y_bytes = (unsigned)(ny * 2);
    q.ablate = getenv("PW_ABLATE") ? atoi(getenv("PW_ABLATE")) : 0;
    std::vector<unsigned short> y0(ny), y1(ny);
    CK(hipMemcpy(y0.data(), dy0, ny * 2, hipMemcpyDeviceToHost));
    const double bytes = (double)(nx + ny * (1 + sh.res)) * 2, flop = 2.0 * M * sh.cin * sh.cout;
    float best = 1e30f; int best_cfg = -1;
    printf("%-26s M=%zu  igemm %.1f us (%.2f TB/s, %.0f TF)\n", sh.name, M, ms_old * 1e3, bytes / ms_old / 1e9, flop / ms_old / 1e9);
    for (int cfg = 0; cfg < spk_pw_num_configs(); ++cfg) {
      if (only_cfg >= 0 && cfg != only_cfg) continue;
      CK(hipMemsetAsync(dy1, 0xee, ny * 2, st));
      const int r0 = spk_pw_launch(q, cfg, st);
      if (r0 == -3) continue;
      if (r0) { fprintf(stderr, "pw launch cfg %d failed: %d\n", cfg, r0); return 4; }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(y1.data(), dy1, ny * 2, hipMemcpyDeviceToHost));
      double maxd = 0, maxv = 0; size_t bad = 0;
      for (size_t i = 0; i < ny; ++i) {
        const float u = (float)__builtin_bit_cast(_Float16, y0[i]), v = (float)__builtin_bit_cast(_Float16, y1[i]);
        const double d = fabs((double)u - v);
        if (!(d <= 4e-3 * (1.0 + fabs(u)))) {
          if (bad < (size_t)dump) printf("      bad px %zu cout %zu: want %g got %g (raw %04x)\n", i / sh.cout, i % sh.cout, u, v, y1[i]);
          ++bad;
        }
        if (d > maxd) maxd = d;
        if (fabs(u) > maxv) maxv = fabs(u);
      }
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) spk_pw_launch(q, cfg, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("    pw cfg %2d: %7.1f us (%.2f TB/s, %4.0f TF alg)  max|d| %.2e of %.1f  bad %zu%s\n", cfg, ms * 1e3,
             bytes / ms / 1e9, flop / ms / 1e9, maxd, maxv, bad, bad ? "  <-- MISMATCH" : "");
      if (!bad && ms < best) { best = ms; best_cfg = cfg; }
    }
    printf("    => best pw cfg %d: %.1f us vs igemm %.1f us (x%.2f)\n", best_cfg, best * 1e3, ms_old * 1e3, ms_old / best);
    fflush(stdout);
    tot_old += ms_old * sh.count; tot_new += (best < ms_old ? best : ms_old) * sh.count;
    hipFree(dx); hipFree(dr); hipFree(dy0); hipFree(dy1); hipFree(wp_old); hipFree(wp_new); hipFree(dw); hipFree(dsc); hipFree(dbi);
  }
  printf("TOTAL over the network's 1x1 convs: igemm %.3f ms -> best-of %.3f ms\n", tot_old, tot_new);
  return 0;
}
