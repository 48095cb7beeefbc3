// Probe: where does a workgroup of the fused MLP forward (csrc/mlp16.hip) spend its cycles?  Builds the kernel with
// per-wave cycle stamps around every layer's MFMA phase, epilogue and barrier and prints the averages.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DMLP_TRACE tools/probe/mlp_trace.hip -o tools/probe/bin/mlp_trace
#include "../../repo_amd/csrc/mlp16.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace repo;

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 34300, F = 230, Hd = 200, O = 1, L = 4;
  std::vector<float> hx((size_t)rows * F);
  for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
  float *x, *out, *hid[4], *ws, *par[10];
  hipMalloc(&x, hx.size() * 4);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, (size_t)rows * 4);
  for (int l = 0; l < L - 1; ++l) hipMalloc(&hid[l], (size_t)rows * Hd * 4);
  hipMalloc(&ws, mlp_fused_ws_floats(F, Hd, O, L) * 4);
  for (int l = 0; l < L; ++l) {
    const int n = l == L - 1 ? O : Hd, k = l == 0 ? F : Hd;
    std::vector<float> w((size_t)n * k), b(n);
    for (auto& v : w) v = (rand() % 2001 - 1000) / 1000.f / 14.f;
    for (auto& v : b) v = 0.01f;
    hipMalloc(&par[2 * l], w.size() * 4);
    hipMalloc(&par[2 * l + 1], b.size() * 4);
    hipMemcpy(par[2 * l], w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(par[2 * l + 1], b.data(), b.size() * 4, hipMemcpyHostToDevice);
  }
  // packs + args exactly as mlp_fused_fwd builds them
  MlpFwdArgs a;
  a.rows = rows, a.in_dim = F, a.hidden = Hd, a.out_dim = O, a.ldx = F, a.ldo = O, a.x = x, a.out = out;
  float* w = ws;
  a.wpack = w;
  PackArgs pa;
  pa.njobs = 0;
  for (int l = 0; l < L; ++l) {
    const int n = l == L - 1 ? O : Hd, k = l == 0 ? F : Hd;
    pa.job[pa.njobs++] = PackJob{par[2 * l], w, n, k, k, 1};
    a.W[l] = (unsigned)((w - a.wpack) * 4);
    a.b[l] = par[2 * l + 1];
    if (l < L - 1) a.hid[l] = hid[l];
    w += pack_floats(n, k);
  }
  a.wbytes = (unsigned)((w - a.wpack) * 4);
  launch_pack(pa, 0);
  const int grid = grid_for(rows);
  const size_t nstamp = (size_t)grid * 8 * 8 * 32;
  long long* trace;
  hipMalloc(&trace, nstamp * 8);
  hipMemset(trace, 0, nstamp * 8);
  a.trace = nullptr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_fwd<4>(a, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) launch_fwd<4>(a, 0);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("rows %d grid %d: %.1f us per launch (no trace)\n", rows, grid, ms * 100);
  a.trace = trace;
  launch_fwd<4>(a, 0);
  hipDeviceSynchronize();
  std::vector<long long> t(nstamp);
  hipMemcpy(t.data(), trace, nstamp * 8, hipMemcpyDeviceToHost);
  // per layer: mfma phase (stamp1 - stamp0) for waves that ran, epilogue (2 - 1), barrier wait (3 - 2); tile 0 and 1
  for (int tile = 0; tile < 3; ++tile) {
    double tile_total = 0;
    int tile_n = 0;
    for (int l = 0; l < L; ++l) {
      double mf[2] = {0, 0}, ep = 0, bar = 0, span = 0;
      int nm[2] = {0, 0}, ne = 0, nw = 0;
      for (int g = 0; g < grid; ++g) {
        long long lo = 0, hi = 0;
        for (int wv = 0; wv < 8; ++wv) {
          const long long* s = &t[((((size_t)g * 8 + tile) * 8 + l) * 32) + wv * 4];
          if (!s[0]) continue;
          if (!lo || s[0] < lo) lo = s[0];
          if (s[3] > hi) hi = s[3];
          if (s[1]) {
            const double d = (double)(s[1] - s[0]);
            const int heavy = d > 0 ? 0 : 0;
            (void)heavy;
            mf[0] += d, nm[0]++;
            if (d > mf[1]) mf[1] = d;
            ep += (double)(s[2] - s[1]), ne++;
          }
          bar += (double)(s[3] - s[2]), nw++;
        }
        if (lo) span += (double)(hi - lo), tile_n += (l == 0);
        if (lo) tile_total += (double)(hi - lo);
      }
      if (!nw) continue;
      printf("tile %d layer %d: layer span %.0f clk; mfma phase mean %.0f (max %.0f) over %d waves; epilogue %.0f; barrier wait %.0f\n",
             tile, l, span / (nw / 8.0), mf[0] / (nm[0] ? nm[0] : 1), mf[1], nm[0], ep / (ne ? ne : 1), bar / nw);
    }
    if (tile_n) printf("tile %d: sum of layer spans %.0f clk per workgroup (%d workgroups)\n", tile, tile_total / tile_n, tile_n);
  }
  return 0;
}
