// Probe: ds_read_b128 from LDS at 2-byte-aligned (not 16-byte-aligned) addresses on gfx950: does it work, what does it cost?
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_unaligned.hip -o tools/probe/bin/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U16B { u32x4 v; };

template <int SHIFT>   // byte misalignment of every lane's 16-byte read
__global__ __launch_bounds__(256) void probe(unsigned* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32768];
  for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  const char* base = reinterpret_cast<const char*>(lds) + SHIFT;
  for (int it = 0; it < iters; ++it) {
    const int off = ((lane * 16 + it * 4096) & 0xfff0) % 65000;
    u32x4 v;
    if (SHIFT % 16 == 0) v = *reinterpret_cast<const u32x4*>(base + off);
    else v = reinterpret_cast<const U16B*>(base + off)->v;
    acc += v;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int SHIFT>
void run(const char* name) {
  unsigned* out;
  hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  probe<SHIFT><<<1024, 256>>>(out, 2000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<SHIFT><<<1024, 256>>>(out, 2000);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned h[4];
  hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  printf("%-28s %8.3f ms   (out[0] = %u)  err=%d\n", name, ms, h[0], (int)hipGetLastError());
  hipFree(out);
}
int main() {
  run<0>("aligned (shift 0)");
  run<2>("shift 2 bytes");
  run<4>("shift 4 bytes");
  run<6>("shift 6 bytes");
  run<8>("shift 8 bytes");
  return 0;
}
