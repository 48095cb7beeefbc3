// Does a memset node of a captured HIP graph take effect before the kernel node that follows it on the same stream?
// The column-split scans armed their exchange buffer with hipMemsetAsync(0xff) right before the scan kernel; under a
// HIP-graph replay of the acting path the posterior came out as garbage on some replays (round 6, tools/act_graph_debug.py).
// Each replay: [kernel A touches other memory AND scribbles buf / flags] -> memset(flags, 0) -> memset(buf, 0xff) -> kernel B counts the cells of
// buf that are NOT 0xFFFFFFFF / flags that are not 0, then overwrites both with data.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/bin/graph_memset tools/probe/graph_memset.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// kernel A also SCRIBBLES over buf / flags (as the encoder's weight packs do in the shared scratch before the scan's
// arming): a memset that runs anywhere BEFORE kernel A in the replay -- not only one that runs too late -- is caught
__global__ void touch(float* p, int n, unsigned* buf, int nb, unsigned* flags, int nf) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 0.5f + 1.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) buf[i] = 0x40000000u + i;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) flags[i] = 5u;
}
__global__ void check_and_scribble(unsigned* buf, int n, unsigned* flags, int nf, unsigned* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (buf[i] != 0xFFFFFFFFu) atomicAdd(bad, 1u);
    buf[i] = 0x3f800000u + i;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
    if (flags[i] != 0u) atomicAdd(bad + 1, 1u);
    flags[i] = 7u;
  }
}

int main(int argc, char** argv) {
  const int replays = argc > 1 ? atoi(argv[1]) : 2000;
  const int n = 8 * 208 * 16, nf = 13 * 32 + 32, nt = 1 << 20;
  float* other; unsigned *buf, *flags, *bad;
  CK(hipMalloc(&other, nt * 4)); CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&flags, nf * 4)); CK(hipMalloc(&bad, 8));
  CK(hipMemset(other, 0, nt * 4)); CK(hipMemset(bad, 0, 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int mode = 0; mode < 2; ++mode) {   // 0: eager launches, 1: graph replays
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(buf, 0, n * 4)); CK(hipMemset(flags, 0xff, nf * 4));
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    auto body = [&]() {
      hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, s, other, nt, buf, n, flags, nf);
      CK(hipMemsetAsync(flags, 0, nf * 4, s));
      CK(hipMemsetAsync(buf, 0xff, n * 4, s));
      hipLaunchKernelGGL(check_and_scribble, dim3(13), dim3(256), 0, s, buf, n, flags, nf, bad);
    };
    if (mode == 1) {
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      body();
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    for (int r = 0; r < replays; ++r) {
      if (mode == 1) CK(hipGraphLaunch(ge, s)); else body();
    }
    CK(hipStreamSynchronize(s));
    unsigned h[2]; CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    printf("%s x %d: buf cells not 0xFFFFFFFF at kernel start: %u, flags not 0: %u\n", mode ? "graph replay" : "eager", replays, h[0], h[1]);
  }
  return 0;
}
