// Row-tile primitives shared by the persistent rollout (imagine16.hip) and the fused MLP heads (mlp16.hip):
// 16-row activation tiles in LDS (k4-interleaved), k4-interleaved weight packs, v_mfma_f32_16x16x4_f32 with both
// operands moved 16 bytes per lane per 4 k-steps, weight streams opened one stage ahead of their use.
#pragma once
#include "common.h"

namespace repo {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kR = 16;      // rows per workgroup
constexpr int kW = 8;       // waves per workgroup (512 threads)
constexpr int kMaxBlk = 15; // K <= 240

__host__ __device__ constexpr int pad16(int k) { return (k + 15) & ~15; }
// k4-interleaved activation tile: element (k, row)
__device__ __forceinline__ int ai(int k, int row) { return (((k >> 2) * kR + row) << 2) + (k & 3); }

// ---- weight pack: dst[(kg*N + n)*4 + u] = W(n, 4kg+u) = src[n*sn + (4kg+u)*sk] for 4kg+u < K, else 0; kg < pad16(K)/4
struct PackJob {
  const float* src;
  float* dst;
  int N, K, sn, sk;
};
constexpr int kMaxJobs = 24;  // weight matrices + bias vectors of one call
struct PackArgs {
  PackJob job[kMaxJobs];
  int njobs;
};
// A FILL job (src == nullptr): dst[0 .. 16 N) = the 32-bit pattern `sn`.  The scans arm their exchange buffers this way,
// inside the pack launch they issue anyway (see fill_job).
static inline PackJob fill_job(void* dst, size_t words, unsigned pattern) {   // words % 16 == 0
  return PackJob{nullptr, (float*)dst, (int)(words / 16), 16, (int)pattern, 0};
}
static __global__ __launch_bounds__(256) void pack16_kernel(PackArgs a) {
  const PackJob j = a.job[blockIdx.y];
  const int total = pad16(j.K) * j.N;
  if (!j.src) {
    const float v = __builtin_bit_cast(float, j.sn);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) j.dst[i] = v;
    return;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int u = i & 3, q = i >> 2;
    const int n = q % j.N, k = 4 * (q / j.N) + u;
    j.dst[i] = k < j.K ? j.src[(size_t)n * j.sn + (size_t)k * j.sk] : 0.f;
  }
}
static inline size_t pack_floats(int64_t N, int64_t K) { return (size_t)pad16((int)K) * N; }
// one launch for all jobs
static inline int launch_pack(PackArgs& pa, hipStream_t s) {
  int mx = 1;
  for (int i = 0; i < pa.njobs; ++i) {
    const int blocks = (pad16(pa.job[i].K) * pa.job[i].N + 255) / 256;
    if (blocks > mx) mx = blocks;
  }
  if (mx > 256) mx = 256;
  hipLaunchKernelGGL(pack16_kernel, dim3(mx, pa.njobs), dim3(256), 0, s, pa);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}


// acc0 (+ acc1) += A[16 x K] * packed W column tile.  A0/A1: LDS tiles (k4-interleaved), W0/W1: packs with N0/N1
// columns, col0/col1: this lane's (clamped) column.  The B fragments run kPD blocks (kPD * 8 MFMAs = ~1300 cycles,
// an L2 round trip) ahead of the MFMAs in a rotating register window.
constexpr int kPD = 4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wrsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
// A weight stream of one or two column tiles: its first kPD blocks are requested by wopen() -- which the caller
// places BEFORE the barrier / the MFMA chain that precedes the stream's use, so no layer starts with an exposed
// L2 round trip -- and wrun() keeps the window kPD blocks ahead.  W0/W1 are BYTE offsets of the packs inside the
// buffer `rw` (wave-uniform): a block's address is lane part (VGPR, one per stream) + scalar offset -- a 64-bit
// VGPR address per block and call site would be hoisted out of the step loop by the compiler and spill the rest.
struct WWin {
  f32x4v b0[kPD], b1[kPD];
  unsigned v0, v1, W0, W1, s0, s1;
  bool act, two;
};
__device__ __forceinline__ f32x4v wld(__amdgpu_buffer_rsrc_t rw, unsigned v, unsigned so) {
  return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rw, v, so, 0));
}
template <int NBLK>
__device__ __forceinline__ void wopen(WWin& w, __amdgpu_buffer_rsrc_t rw, bool act, unsigned W0, int N0, int col0,
                                      unsigned W1, int N1, int col1, bool two, int lane) {
  constexpr int PD = NBLK < kPD ? NBLK : kPD;
  const int kq = lane >> 4;
  w.act = act;
  w.two = two;
  w.v0 = 16u * (unsigned)(kq * N0 + col0);
  w.v1 = 16u * (unsigned)(kq * N1 + col1);
  w.W0 = W0;
  w.W1 = W1;
  w.s0 = 64u * (unsigned)N0;  // bytes per 16-k block of a pack
  w.s1 = 64u * (unsigned)N1;
  // always both tiles, also for an idle wave (clamped columns: the loads are valid): conditional stores into the
  // window make the compiler keep it in scratch memory with run-time offsets
#pragma unroll
  for (int b = 0; b < PD; ++b) {
    w.b0[b] = wld(rw, w.v0, W0 + b * w.s0);
    w.b1[b] = wld(rw, w.v1, W1 + b * w.s1);
  }
}
// acc0 (+ acc1) += (A[16 x K] * the stream's column tile(s))^T.  A0/A1: LDS tiles (k4-interleaved).  The weights
// are the MFMA's A operand and the activations its B operand (the two fragments have the same lane layout), so the
// accumulator is transposed: lane l holds tile row l % 16 and the FOUR CONSECUTIVE columns 4 * (l / 16) + r of the
// column tile -- a "quad": one 16-byte LDS access in the k4-interleaved tile of the next layer, one 16-byte global
// access in a row-major activation.
template <int NBLK>
__device__ __forceinline__ void wrun(f32x4v& acc0, f32x4v& acc1, const float* A0, const float* A1, WWin& w,
                                     __amdgpu_buffer_rsrc_t rw, int lane) {
  constexpr int PD = NBLK < kPD ? NBLK : kPD;
  if (!w.act) return;
  const int row = lane & 15, kq = lane >> 4;
  const float* a0p = A0 + (kq * kR + row) * 4;
  const float* a1p = A1 + (kq * kR + row) * 4;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    const f32x4v a0 = *reinterpret_cast<const f32x4v*>(a0p + b * 16 * kR);
    const f32x4v a1 = (A1 == A0) ? a0 : *reinterpret_cast<const f32x4v*>(a1p + b * 16 * kR);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.b0[b % PD][j], a0[j], acc0, 0, 0, 0);
      if (w.two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.b1[b % PD][j], a1[j], acc1, 0, 0, 0);
    }
    if (b + PD < NBLK) {
      w.b0[b % PD] = wld(rw, w.v0, w.W0 + (b + PD) * w.s0);
      w.b1[b % PD] = wld(rw, w.v1, w.W1 + (b + PD) * w.s1);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the window: no hoisting of later blocks' loads
  }
}
// Dense layer on the row tile: wave w owns column tiles w and w + 8.
template <int NBLK>
__device__ __forceinline__ void dense_open(WWin& w, __amdgpu_buffer_rsrc_t rw, unsigned W, int N, int wave, int lane) {
  const int ntiles = (N + 15) >> 4;
  const int n = lane & 15;
  wopen<NBLK>(w, rw, wave < ntiles, W, N, min(wave * 16 + n, N - 1), W, N, min((wave + kW) * 16 + n, N - 1),
              wave + kW < ntiles, lane);
}
// epi(valid, n0, acc): acc[r] = element (row lane % 16, column n0 + r) of the layer's output, n0 = 16 tile + 4 (lane / 16);
// valid = n0 < N (columns n0 + r >= N of a ragged last quad hold zeros from the padded pack).
template <int NBLK, class Epi>
__device__ __forceinline__ void dense_run(const float* A, WWin& w, __amdgpu_buffer_rsrc_t rw, int N, int wave, int lane,
                                          Epi epi) {
  if (!w.act) return;
  const int c0 = wave * 16 + 4 * (lane >> 4), c1 = c0 + kW * 16;
  f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  wrun<NBLK>(acc0, acc1, A, A, w, rw, lane);
  epi(c0 < N, c0, acc0);
  if (w.two) epi(c1 < N, c1, acc1);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0): every global store of
// saved activations still in flight (an acknowledgement from L2 takes a microsecond under load) would be waited for at
// every layer boundary.  Nothing a workgroup stores to global memory in these kernels is read back by it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- quads.  Tile side: columns n0 .. n0+3 (n0 % 4 == 0) of tile row m are one 16-byte word.
__device__ __forceinline__ f32x4v ldq(const float* T, int n0, int m) {
  return *reinterpret_cast<const f32x4v*>(T + ((n0 >> 2) * kR + m) * 4);
}
__device__ __forceinline__ void stq(float* T, int n0, int m, const f32x4v& v) {
  *reinterpret_cast<f32x4v*>(T + ((n0 >> 2) * kR + m) * 4) = v;
}
// Global side: a quad of a row-major activation through a raw buffer (dword alignment is all a buffer access
// needs): voff = this lane's byte offset (32 bits, invariant over the step loop -- a 64-bit address per store site
// would be hoisted out of that loop and spilled), soff = the step's byte offset (scalar); nv = valid columns of the
// quad (>= 4: all).
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t arsrc(const float* p) {  // whole address space behind p
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0xfffffffcu, 0x00020000);
}
#ifndef RT_STORE_AUX
#define RT_STORE_AUX 0
#endif
__device__ __forceinline__ void bstq(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const f32x4v& v, int nv) {
#ifdef RT_NO_STORE  // ablation build (tools/build_variant.sh): results wrong, time meaningful
  if (v[0] != 123.456f) return;
#endif
  // The elements of a ragged quad are copied out BEFORE the branch and pinned in registers of their own: with the
  // extraction inside the else-branch the compiler (ROCm 7.2) stored element 0 three times (the other side of the
  // branch ends the vector's live range).
  float e0 = v[0], e1 = v[1], e2 = v[2];
  asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2));
  if (nv >= 4) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, voff, soff, RT_STORE_AUX);
  } else {
    if (nv > 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e0), r, voff, soff, RT_STORE_AUX);
    if (nv > 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e1), r, voff + 4u, soff, RT_STORE_AUX);
    if (nv > 2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e2), r, voff + 8u, soff, RT_STORE_AUX);
  }
}
__device__ __forceinline__ f32x4v bldq(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {  // a whole quad
  return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// bias quad: the bias vectors are packed behind the weights (N = 1 jobs: copied, zero-padded to 16), so a quad is one
// aligned 16-byte buffer load whose lane offset does not depend on the vector
__device__ __forceinline__ f32x4v ldbias(__amdgpu_buffer_rsrc_t rw, unsigned boff, int n0) {
  return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rw, 4u * (unsigned)n0, boff, 0));
}

}  // namespace repo
