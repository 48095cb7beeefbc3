// Probe: latency / issue rate of v_mfma_f32_32x32x2_f32 and 16x16x4 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 8 * NACC);
}
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 8 * NACC);
}
int main() {
  float* d; hipMalloc(&d, 1 << 20);
  float h;
#define RUN(K, name, blocks, threads) \
  K<<<blocks, threads>>>(d, 2000, 1.0f, 0.5f); hipDeviceSynchronize(); \
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("%-44s %8.1f clk/mfma (s_memtime ticks, per wave)\n", name, h);
  RUN((k32<1>), "32x32x2 1 acc, 1 wave/CU", 1, 64)
  RUN((k32<2>), "32x32x2 2 acc, 1 wave/CU", 1, 64)
  RUN((k32<4>), "32x32x2 4 acc, 1 wave/CU", 1, 64)
  RUN((k32<1>), "32x32x2 1 acc, 4 waves/CU (1/SIMD)", 1, 256)
  RUN((k32<1>), "32x32x2 1 acc, 8 waves/CU (2/SIMD)", 1, 512)
  RUN((k32<1>), "32x32x2 1 acc, 256 CUs x 4 waves", 256, 256)
  RUN((k32<2>), "32x32x2 2 acc, 256 CUs x 4 waves", 256, 256)
  RUN((k16<1>), "16x16x4 1 acc, 1 wave/CU", 1, 64)
  RUN((k16<2>), "16x16x4 2 acc, 1 wave/CU", 1, 64)
  RUN((k16<4>), "16x16x4 4 acc, 1 wave/CU", 1, 64)
  RUN((k16<4>), "16x16x4 4 acc, 256 CUs x 4 waves", 256, 256)
  return 0;
}
