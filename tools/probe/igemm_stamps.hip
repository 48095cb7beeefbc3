// In-kernel phase stamps of the igemm K loop (s_memtime), for the dense GEMM and the conv layers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DREPO_IGEMM_STAMPS -DPROBE_GEMM tools/probe/igemm_stamps.hip -o /tmp/st_gemm
//   hipcc ... -DPROBE_CONV ... -o /tmp/st_conv
#include <cstdio>
#include <cstdlib>
#include <vector>
#if defined(PROBE_GEMM)
#include "../../repo_amd/csrc/gemm.hip"
#else
#include "../../repo_amd/csrc/conv.hip"
#endif

namespace repo { int arch_status() { return 0; } }  // the probe links no api.hip

static void report(const char* name, double flop, float ms) {
  unsigned long long h[8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(repo::g_igemm_stamps), sizeof(h));
  const double w = (double)h[6], sl = (double)h[7];
  printf("%-28s %8.1f us %6.1f TF | waves %6.0f slices/wave %5.1f | per slice: issue %5.0f mfma %5.0f lds-write %5.0f barrier %5.0f | prologue %6.0f epilogue %6.0f\n",
         name, ms * 1e3, flop / ms / 1e9, w, sl / w, h[0] / sl, h[1] / sl, h[2] / sl, h[3] / sl, h[5] / w, h[4] / w);
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(repo::g_igemm_stamps), z, sizeof(z));
}

template <class F>
static float run(F f) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  f();
  (void)hipDeviceSynchronize();
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(repo::g_igemm_stamps), z, sizeof(z));
  (void)hipEventRecord(a, 0);
  int rc = f();
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  if (rc) printf("rc=%d\n", rc);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

static float* dev_rand(size_t n) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  float* d;
  (void)hipMalloc(&d, n * sizeof(float));
  (void)hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

int main() {
#if defined(PROBE_GEMM)
  struct { long M, N, K; } cases[] = {{4096, 4096, 4096}, {2450, 3200, 1024}, {34300, 200, 230}, {2450, 600, 200}};
  for (auto c : cases)
    for (int tb = 0; tb < 2; ++tb) {
      float *A = dev_rand(c.M * c.K), *B = dev_rand(c.N * c.K), *C = dev_rand(c.M * c.N);
      float ms = run([&] { return repo_gemm(0, tb, c.M, c.N, c.K, A, c.K, B, tb ? c.K : c.N, nullptr, 1, C, c.N, 0, nullptr, 0, 0, 0); });
      char nm[64];
      snprintf(nm, sizeof nm, "gemm %ldx%ldx%ld %s", c.M, c.N, c.K, tb ? "nt" : "nn");
      report(nm, 2.0 * c.M * c.N * c.K, ms);
      (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
    }
#else
  const long n = 2450;
  // layer ids as in include/repo_hip.h: 0..3 encoder conv1..4, 4..6 decoder conv2..4
  struct { int id; const char* nm; long cb, cs, hb, hs, ks; } L[] = {
      {1, "enc2", 32, 64, 31, 14, 4}, {2, "enc3", 64, 128, 14, 6, 4}, {3, "enc4", 128, 256, 6, 2, 4},
      {4, "dec2", 64, 128, 13, 5, 5}, {5, "dec3", 32, 64, 30, 13, 6}, {6, "dec4", 3, 32, 64, 30, 6}};
  for (auto l : L) {
    float *big = dev_rand(n * l.cb * l.hb * l.hb), *small = dev_rand(n * l.cs * l.hs * l.hs);
    float *w = dev_rand(l.cb * l.cs * l.ks * l.ks), *dw = dev_rand(l.cb * l.cs * l.ks * l.ks);
    const double flop = 2.0 * n * l.hs * l.hs * l.cs * l.cb * l.ks * l.ks;
    size_t wsb = repo_conv_wgrad_workspace_bytes(l.id, n);
    void* ws;
    (void)hipMalloc(&ws, wsb ? wsb : 4);
    char nm[64];
    float ms = run([&] { return repo_conv_down(l.id, n, big, 0, w, nullptr, small, 0, nullptr, nullptr, 0, nullptr, 0, 0); });
    snprintf(nm, sizeof nm, "%s down", l.nm);
    report(nm, flop, ms);
    ms = run([&] { return repo_conv_wgrad(l.id, n, small, big, 0, dw, nullptr, 0, ws, wsb, 0); });
    snprintf(nm, sizeof nm, "%s wgrad", l.nm);
    report(nm, flop, ms);
    (void)hipFree(big); (void)hipFree(small); (void)hipFree(w); (void)hipFree(dw); (void)hipFree(ws);
  }
#endif
  return 0;
}
