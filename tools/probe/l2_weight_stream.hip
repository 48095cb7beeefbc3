// Probe: how fast can EVERY workgroup of the chip stream the SAME few hundred KB of packed weights out of L2 (16-byte loads,
// registers only)?  The gather-form down convolution sized in DESIGN section 7 (Next (0)) needs 11.6 B/clk and CU = ~6 TB/s
// over the chip for decoder conv3's 442 KB; this measures the ceiling, with the workgroups in lockstep or staggered and with
// 4 or 8 waves per CU asking.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/l2_weight_stream.hip -o tools/probe/bin/l2_weight_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int DEPTH>   // DEPTH 16-byte loads in flight per lane
__global__ __launch_bounds__(NT) void stream(const u32x4* __restrict__ w, int nvec, int passes, int stagger, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  const int start = stagger ? (int)(((long)blockIdx.x * 9973 * NT) % nvec) : 0;
  for (int p = 0; p < passes; ++p) {
    for (int v0 = 0; v0 < nvec; v0 += NT * DEPTH) {
      u32x4 r[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        int v = start + v0 + d * NT + (int)threadIdx.x;
        v = v >= nvec ? v - nvec : v;
        v = v >= nvec ? v - nvec : v;
        r[d] = __builtin_nontemporal_load(&w[v]);   // (plain loads measured too: see main)
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += r[d];
    }
  }
  out[blockIdx.x * NT + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int NT, int DEPTH>
__global__ __launch_bounds__(NT) void stream_plain(const u32x4* __restrict__ w, int nvec, int passes, int stagger, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  const int start = stagger ? (int)(((long)blockIdx.x * 9973 * NT) % nvec) : 0;
  for (int p = 0; p < passes; ++p) {
    for (int v0 = 0; v0 < nvec; v0 += NT * DEPTH) {
      u32x4 r[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        int v = start + v0 + d * NT + (int)threadIdx.x;
        v = v >= nvec ? v - nvec : v;
        v = v >= nvec ? v - nvec : v;
        r[d] = w[v];
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += r[d];
    }
  }
  out[blockIdx.x * NT + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <class K>
void run(const char* name, K kern, int nt, int wgs, const u32x4* w, int kb, int stagger, unsigned* out) {
  const int nvec = kb * 1024 / 16, passes = 64;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(nt), 0, 0, w, nvec, passes, stagger, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(nt), 0, 0, w, nvec, passes, stagger, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * passes * nvec * 16;
  printf("%-28s %4d KB x %4d workgroups of %3d threads, %s: %8.1f us  %6.2f TB/s  = %5.1f B per CU and ns\n", name, kb, wgs, nt,
         stagger ? "staggered" : "lockstep ", ms * 1e3, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
}

int main() {
  u32x4* w;
  unsigned* out;
  hipMalloc(&w, 4 << 20);
  hipMemset(w, 1, 4 << 20);
  hipMalloc(&out, 2048 * 512 * 4);
  for (int kb : {48, 442, 1024}) {
    for (int st = 0; st < 2; ++st) {
      run("plain, 4 waves, depth 4", stream_plain<256, 4>, 256, 256, w, kb, st, out);
      run("plain, 4 waves, depth 8", stream_plain<256, 8>, 256, 256, w, kb, st, out);
      run("plain, 8 waves, depth 4", stream_plain<512, 4>, 512, 256, w, kb, st, out);
      run("plain, 2 x 4 waves, depth 4", stream_plain<256, 4>, 256, 512, w, kb, st, out);
      run("nontemporal, 4 waves, d 4", stream<256, 4>, 256, 256, w, kb, st, out);
    }
  }
  return 0;
}
