// Host-side sanitizer driver (the reference has no native code, so no sanitizer story; this
// library has ~9 k lines of it).  Built by `build.sh asan` with -fsanitize=address,undefined on the HOST pass of every
// translation unit and run on the CPU (tests/test_abi.py): it walks the host logic that does not need a GPU - handle
// creation and its error paths, the parameter table, spk_last_error, and the three tuner-cache parsers - so that
// heap overflows, use-after-free and undefined behaviour there abort the run.  With no GPU every HIP call fails and
// the error paths are what runs; on a GPU box the same binary works on real (small) buffers.
#include "model.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int g_fail = 0;
#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "asan_driver: %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

static spk_layer_desc conv(const char* name, const char* bn, int cin, int cout, int k, int s, int p, int src, int dst,
                           int res, int relu, int child) {
  spk_layer_desc d;
  memset(&d, 0, sizeof d);
  d.kind = SPK_OP_CONV; d.cin = cin; d.cout = cout; d.k = k; d.stride = s; d.pad = p; d.relu = relu;
  d.src = src; d.dst = dst; d.res = res; d.child = child;
  snprintf(d.name, sizeof d.name, "%s", name);
  snprintf(d.bn, sizeof d.bn, "%s", bn);
  return d;
}

int main() {
  int ndev = 0;
  const bool gpu = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
  printf("asan_driver: %d GPU(s)\n", gpu ? ndev : 0);

  // ---- argument errors ----
  spk_model* m = nullptr;
  EXPECT(spk_model_create(nullptr, 0, 3, 10, 0, &m) != SPK_OK);
  EXPECT(strlen(spk_last_error()) > 0);
  EXPECT(spk_model_num_params(nullptr) == 0);
  spk_model_destroy(nullptr);

  // ---- a small bottleneck graph: stem, pool, 1x1 / 3x3 / 1x1 + shortcut, pool, head ----
  std::vector<spk_layer_desc> g;
  g.push_back(conv("base.0", "base.1", 3, 64, 7, 2, 3, 0, 1, -1, 1, 0));
  { spk_layer_desc d; memset(&d, 0, sizeof d); d.kind = SPK_OP_MAXPOOL; d.cin = d.cout = 64; d.k = 3; d.stride = 2; d.pad = 1;
    d.src = 1; d.dst = 2; d.res = -1; d.child = 3; snprintf(d.name, sizeof d.name, "base.3"); g.push_back(d); }
  g.push_back(conv("base.4.0.conv1", "base.4.0.bn1", 64, 64, 1, 1, 0, 2, 3, -1, 1, 4));
  g.push_back(conv("base.4.0.conv2", "base.4.0.bn2", 64, 64, 3, 1, 1, 3, 4, -1, 1, 4));
  g.push_back(conv("base.4.0.downsample.0", "base.4.0.downsample.1", 64, 256, 1, 1, 0, 2, 5, -1, 0, 4));
  g.push_back(conv("base.4.0.conv3", "base.4.0.bn3", 64, 256, 1, 1, 0, 4, 6, 5, 1, 4));
  { spk_layer_desc d; memset(&d, 0, sizeof d); d.kind = SPK_OP_GAVGPOOL; d.cin = d.cout = 256; d.src = 6; d.dst = 7; d.res = -1;
    d.child = 8; snprintf(d.name, sizeof d.name, "base.8"); g.push_back(d); }
  { spk_layer_desc d; memset(&d, 0, sizeof d); d.kind = SPK_OP_LINEAR; d.cin = 256; d.cout = 10; d.src = 7; d.dst = 8; d.res = -1;
    d.child = -1; snprintf(d.name, sizeof d.name, "head.0"); g.push_back(d); }
  const int rc = spk_model_create(g.data(), (int)g.size(), 3, 10, 0, &m);
  if (!gpu) {
    EXPECT(rc != SPK_OK && m == nullptr);          // no device: every allocation fails, nothing may leak or dangle
    EXPECT(strlen(spk_last_error()) > 0);
  } else {
    EXPECT(rc == SPK_OK && m != nullptr);
  }
  if (m) {
    const int np = spk_model_num_params(m);
    EXPECT(np > 10);
    for (int i = -1; i <= np; ++i) {
      char key[8];                                  // deliberately short: the name must be truncated, not overrun
      int64_t shape[4];
      int ndim = 0, dtype = 0;
      const int r = spk_model_param_info(m, i, key, (int)sizeof key, shape, &ndim, &dtype);
      EXPECT((r == SPK_OK) == (i >= 0 && i < np));
      if (r == SPK_OK) EXPECT(strlen(key) < sizeof key);
    }
    std::vector<float> w(64 * 3 * 7 * 7, 0.5f);
    EXPECT(spk_model_load_param(m, "no.such.key", w.data(), (int64_t)w.size()) == SPK_ERR_KEY);
    EXPECT(spk_model_load_param(m, "base.0.weight", w.data(), 7) != SPK_OK);           // wrong element count
    EXPECT(spk_model_load_param(m, "base.0.weight", w.data(), (int64_t)w.size()) == SPK_OK);
    EXPECT(spk_model_set_requires_grad(m, "no.such.key", 1) == SPK_ERR_KEY);
    EXPECT(spk_model_set_param_group(m, "base.0.weight", 2) == SPK_OK);
    spk_model_destroy(m);
  }

  // ---- tuner-cache parsers: valid lines of all three kinds, garbage, over-long and truncated lines ----
  const std::string path = std::string(getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp") + "/spk_asan_tune_cache.txt";
  if (FILE* f = fopen(path.c_str(), "w")) {
    fprintf(f, "conv 0 1 1 256 14 14 256 256 3 1 16 0 0 4 3\n");
    fprintf(f, "pw1x1 2 14 14 256 1024 1 1 1 256 7\n");
    fprintf(f, "c3 1 14 14 256 256 256 2\n");
    fprintf(f, "pw1x1 2 14 14 256 1024 1 1 1 256 99999\n");       // configuration id out of range: ignored
    fprintf(f, "c3 1 14 14\n");                                    // truncated
    fprintf(f, "conv x y z\n\n,,,,\n");
    for (int i = 0; i < 3000; ++i) fputc('9', f);                  // longer than the parsers' line buffers
    fprintf(f, "\nwgrad 1 2 3\n");
    fclose(f);
  }
  setenv("SPK_TUNE_CACHE", path.c_str(), 1);
  setenv("SPK_AUTOTUNE", "0", 1);
  // a problem that is NOT in the file (channels 128), on real buffers when there is a GPU
  const int n = 1, h = 8, wd = 8, cin = 128, cout = 128, M = n * h * wd;
  bf16_t *x = nullptr, *y = nullptr, *wp = nullptr;
  float* sb = nullptr;
  if (gpu) {
    EXPECT(hipMalloc(&x, (size_t)M * cin * 2) == hipSuccess && hipMalloc(&y, (size_t)M * 256 * 2) == hipSuccess);
    EXPECT(hipMalloc(&wp, (size_t)256 * 9 * cin * 4) == hipSuccess && hipMalloc(&sb, (size_t)2 * 256 * 4) == hipSuccess);
    (void)hipMemset(x, 0, (size_t)M * cin * 2);
    (void)hipMemset(wp, 0, (size_t)256 * 9 * cin * 4);
    (void)hipMemset(sb, 0, (size_t)2 * 256 * 4);
  }
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.cfg = a.dma = -1; a.cls_ph = a.cls_pw = -1;
  a.x = x; a.w = wp; a.y = y; a.scale = sb; a.bias = sb ? sb + cout : nullptr;
  a.N = n; a.H = h; a.W = wd; a.Cin = cin; a.Ho = h; a.Wo = wd; a.Cout = cout; a.kh = a.kw = 1; a.stride = 1; a.M = M;
  a.K = cin; a.relu = 1; a.dt = DT_F16; a.splitw = 1;
  a.x_bytes = (unsigned)((size_t)M * cin * 2); a.w_bytes = (unsigned)((size_t)cout * cin * 4);
  PwConvArgs q;
  memset(&q, 0, sizeof q);
  q.x = x; q.wp = wp; q.y = y; q.scale = sb; q.shift = sb ? sb + cout : nullptr;
  q.N = n; q.H = h; q.W = wd; q.Ho = h; q.Wo = wd; q.stride = 1; q.Cin = cin; q.Cout = cout; q.M = M; q.relu = 1;
  q.dt = DT_F16; q.nb = 2; q.x_bytes = a.x_bytes; q.y_bytes = (unsigned)((size_t)M * cout * 2);
  const int r1 = spk_conv1x1_launch(a, q, nullptr);               // loads the "pw1x1" lines, then the "conv" lines
  EXPECT(gpu ? r1 == 0 : r1 != 0);
  C3Args c;
  memset(&c, 0, sizeof c);
  c.x = x; c.wp = wp; c.y = y; c.scale = sb; c.shift = sb ? sb + 256 : nullptr;
  c.N = n; c.H = h; c.W = wd; c.Cin = cin; c.Cout = 256; c.M = M; c.relu = 1; c.dt = DT_F16; c.nb = 1;
  c.x_bytes = a.x_bytes; c.y_bytes = (unsigned)((size_t)M * 256 * 2); c.wp_bytes = (unsigned)((size_t)256 * 9 * cin * 2);
  const int r3 = spk_conv3x3_launch(c, nullptr);                  // loads the "c3" lines
  EXPECT(gpu ? r3 == 0 : r3 != 0);
  if (gpu) {
    (void)hipDeviceSynchronize();
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(wp); (void)hipFree(sb);
  }
  remove(path.c_str());
  printf("asan_driver: %s\n", g_fail ? "FAILED" : "ok");
  return g_fail ? 1 : 0;
}
