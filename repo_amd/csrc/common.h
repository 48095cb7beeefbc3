// Shared device/host helpers for librepo_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/repo_hip.h"

#define REPO_CHECK_LAUNCH()                      \
  do {                                           \
    hipError_t _e = hipGetLastError();           \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)

#define REPO_REQUIRE(cond, code) \
  do {                           \
    if (!(cond)) return (code);  \
  } while (0)

namespace repo {

constexpr float kLog2Pi = 1.8378770664093453f;  // ln(2*pi)
constexpr int kMaxIdx = 0x7fffffff;

__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : expm1f(x); }
// derivative of ELU expressed through its OUTPUT h (h = e^x - 1 for x <= 0)
__device__ __forceinline__ float elu_grad_from_out(float h) { return h > 0.f ? 1.f : h + 1.f; }
// torch F.softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
// u8 pixel -> [-1,1], same expression/rounding as common/utils.py:79 ((x/255)*2)-1
__device__ __forceinline__ float pix_norm(uint8_t v) { return ((float)v / 255.f) * 2.f - 1.f; }

__device__ __forceinline__ float load_as_float(const float* p, int i) { return p[i]; }
__device__ __forceinline__ float load_as_float(const uint8_t* p, int i) { return pix_norm(p[i]); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). Result valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  __syncthreads();
  return r;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace repo
