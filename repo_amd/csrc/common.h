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

// First statement of every entry point that launches work: REPO_E_ARCH unless the current device is gfx950
// (queried once per device, api.hip).
#define REPO_ARCH_GUARD()                 \
  do {                                    \
    int _a = repo::arch_status();         \
    if (_a != REPO_OK) return _a;         \
  } while (0)

#define REPO_REQUIRE(cond, code) \
  do {                           \
    if (!(cond)) return (code);  \
  } while (0)

namespace repo {

// REPO_OK when the calling thread's current HIP device is gfx950, REPO_E_ARCH when it is anything else, a
// positive hipError_t when the device cannot be queried.  Cached per device ordinal after the first query.
int arch_status();

constexpr float kLog2Pi = 1.8378770664093453f;  // ln(2*pi)
constexpr int kMaxIdx = 0x7fffffff;
// Operands addressed through raw buffer descriptors (32-bit num_records / byte offsets, with 0x80000000 as
// the "out of range" offset) must stay below 2 GiB: 2^29 floats.
constexpr int64_t kMaxBufElems = (int64_t)1 << 29;

// Activation math on the hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp
// each) instead of libm's expm1f/log1pf/tanhf call sequences, which cost 20-40 VALU instructions per
// element and dominated the epilogues of the fused row-tile kernels.  Absolute error <= ~1e-7, i.e.
// the same order as the fp32 accumulation-order differences already present versus the reference.
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
// derivative of ELU expressed through its OUTPUT h (h = e^x - 1 for x <= 0)
__device__ __forceinline__ float elu_grad_from_out(float h) { return h > 0.f ? 1.f : h + 1.f; }
// torch F.softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : __logf(1.f + __expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return rcp_fast(1.f + __expf(-x)); }
// tanh(x) = 1 - 2/(e^{2x}+1): saturates correctly at +-1 for large |x|
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * rcp_fast(__expf(2.f * x) + 1.f); }
// u8 pixel -> [-1,1], same expression/rounding as common/utils.py:79 ((x/255)*2)-1
__device__ __forceinline__ float pix_norm(uint8_t v) { return ((float)v / 255.f) * 2.f - 1.f; }

__device__ __forceinline__ float load_as_float(const float* p, unsigned i) { return p[i]; }
__device__ __forceinline__ float load_as_float(const uint8_t* p, unsigned i) { return pix_norm(p[i]); }

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

// "The last block finishes": the tail of a SMALL-GRID reduction kernel whose blocks have each written their partials
// parts[v][blk] (v < nvals, row pitch n = gridDim.x).  Every block calls it once, at its end, with all threads; the block
// that takes the last ticket sums the partials in a FIXED order -- the same order and arithmetic as the one-block follow-up
// launch (final_sum_kernel), so the result is bitwise what the two-launch form gives -- writes out[v] and puts the ticket
// word back to 0.  `ticket` must be 0 when the kernel starts: it lives in the zeroed header of the reduction workspace
// (repo_reduce_workspace_bytes; the Python side allocates that buffer zeroed, per stream, and nothing else writes it).
// ticket == nullptr: nothing happens (the caller launches the follow-up kernel).
// Only for grids of at most kLastBlockMaxGrid blocks: the tickets are same-address atomics and serialise in L2 at ~30-75 ns
// each -- with 1000 blocks the "saved" launch cost 40-75 us (measured, round 6: sqnorm 6 -> 38 us, normal_entropy 5 -> 80,
// kl 6 -> 51), with <= 64 it costs ~2 us against a ~5 us dependent launch.  blockDim.x == 256.
constexpr int kRedHeaderBytes = 256;
constexpr int kLastBlockMaxGrid = 64;
__device__ __forceinline__ void last_block_finishes(const float* parts, int nvals, float* out, unsigned* ticket, float* smem) {
  if (!ticket) return;   // (kernel argument: uniform)
  __shared__ int s_last;
  __threadfence();          // this block's partials are visible device-wide before its ticket is
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const int n = gridDim.x;
  for (int v = 0; v < nvals; ++v) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      s += __hip_atomic_load(parts + v * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past this CU's L1
    s = block_sum(s, smem);
    if (threadIdx.x == 0) out[v] = s;
  }
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------------- noise
// Standard normals from a counter-based generator (Philox4x32-10, Salmon et al. SC'11; key = seed, counter =
// element index / 4) + Box-Muller, so that a kernel can DRAW the reparameterisation noise it needs instead of
// reading a tensor the host filled first ("noise tensors are explicit inputs, nullable =>
// in-kernel Philox with (seed, offset)").  Normal number i of stream `seed` is a pure function of (seed, i): the
// forward and the backward kernel of an op, and repo_philox_normal (which materialises the tensor for the
// parity tests and the per-step fallback engine), all see the same values.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned h0 = __umulhi(0xD2511F53u, c[0]), l0 = 0xD2511F53u * c[0];
    const unsigned h1 = __umulhi(0xCD9E8D57u, c[2]), l1 = 0xCD9E8D57u * c[2];
    const unsigned n0 = h1 ^ c[1] ^ k0, n2 = h0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = l1; c[2] = n2; c[3] = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// the four normals of counter block `blk` (elements 4 blk .. 4 blk + 3)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t blk, float (&z)[4]) {
  unsigned c[4] = {(unsigned)blk, (unsigned)(blk >> 32), 0u, 0u};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.f / 16777216.f);      // (0, 1)
    const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.f / 16777216.f);
    // hardware transcendentals (v_log_f32 = log2, v_sqrt_f32, v_sin/v_cos_f32 take REVOLUTIONS: sin(2 pi x)), ~1 ulp
    // each: the noise only has to be N(0,1), and every consumer (the kernels that draw it, their backward passes,
    // repo_philox_normal) evaluates this one function
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln 2 * log2 u1
    z[2 * h] = rad * __builtin_amdgcn_cosf(u2);
    z[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
  }
}
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t i) {
  float z[4];
  philox_normal4(seed, i >> 2, z);
  const int l = (int)(i & 3);
  return l == 0 ? z[0] : l == 1 ? z[1] : l == 2 ? z[2] : z[3];
}
// A noise operand: an explicit tensor, or (p == nullptr) normals base, base+1, ... of stream `seed`.
struct NoiseSrc {
  const float* p;
  uint64_t seed, base;
  __device__ __forceinline__ float at(size_t i) const { return p ? p[i] : philox_normal(seed, base + i); }
};

// Several dense weight gradients dW_i[n][k] = sum_m dY_i[m][n] X_i[m][k] (+ bias gradients) in one launch pair
// (gemm.hip); at most kMaxWgradGroup - 1 jobs per call.
constexpr int kMaxWgradGroup = 9;
struct WgradDesc {
  int64_t M, N, K;
  const float* dY;
  int64_t lddy;
  const float* X;
  int64_t ldx;
  float* dW;
  int64_t lddw;
  float* db;
};
size_t gemm_wgrad_group_ws_bytes(const WgradDesc* d, int n);
int gemm_wgrad_group(const WgradDesc* d, int n, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

}  // namespace repo
