// Probe for VERDICT r2 #3: what ONE layer exchange of a column-split, weight-stationary observe scan costs.
//
// The scan's in-step chain is (state, action) -> 200 -> GRU -> belief (200) -> posterior hidden (200) -> (mean, std).
// Column-split over W workgroups, every layer whose input is a full (rows x 200) activation needs an all-gather
// of per-workgroup column slices between two DEPENDENT layers.  This kernel runs only that exchange, `nex` times
// per step for `steps` steps, with no arithmetic in between:
//   publish : each workgroup stores its (rows x 200/W) slice with sc1 (write-through) 16-byte stores,
//             s_waitcnt vmcnt(0), workgroup barrier, one lane stores the epoch into the workgroup's flag (sc1)
//   gather  : wave 0 polls the W flags (one lane per flag, sc1 loads), workgroup barrier, all 256 threads load the
//             whole (rows x 200) activation with sc1 16-byte loads (>= 8 in flight per lane) into LDS, barrier
// (the hand-off form of MI355X_MICROARCH.md's table, row 1).  Every gathered value is checked (value = f(epoch,
// row, column)), so a stale read shows up as an error count, not as a fast number.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/scan_exchange tools/probe/scan_exchange.hip
//   tools/probe/bin/scan_exchange            (prints the table committed as profiles/r03_scan_exchange_probe.txt)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int COLS = 208;  // 200 padded to 13 x 16
constexpr int SC1 = 16;    // cache-policy bit of the raw buffer builtins on gfx940+: sc1

__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__device__ inline float expect(int epoch, int row, int col) { return (float)((epoch * 131 + row * 17 + col) & 0xffff); }

struct Args {
  float* act;        // [2][rows][COLS]
  unsigned* flags;   // [W] on 128-byte lines (32 words apart)
  unsigned* errors;  // [1]
  unsigned* spins;   // [1] polls that did not match (contention gauge)
  int rows, W, steps, nex, spin_limit;
};

template <int MODE>  // 0: sc1 stores + sc1 loads (no fences);  1: plain stores + release fence, acquire fence + plain loads
__global__ __launch_bounds__(256) void exchange_kernel(Args a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, w = blockIdx.x;
  const int cw = COLS / a.W;  // columns per workgroup (a.W divides 208: 13, 26, 16 -> 13 cols ... see main)
  const unsigned act_bytes = 2u * a.rows * COLS * 4u;
  const __amdgpu_buffer_rsrc_t ra = rsrc(a.act, act_bytes);
  unsigned nerr = 0, nspin = 0;
  const int total = a.steps * a.nex;
  const int nvec_row = COLS / 4;                 // 52 float4 per row
  const int nvec = a.rows * nvec_row;            // gathered vectors
  const int svec_row = cw / 4;                   // float4 per slice row
  const int svec = a.rows * svec_row;
  for (int e = 0; e < total; ++e) {
    const int buf = e & 1;
    // ---- publish this workgroup's slice
    for (int v = tid; v < svec; v += 256) {
      const int row = v / svec_row, c4 = v % svec_row;
      const int col = w * cw + 4 * c4;
      f32x4 val = {expect(e, row, col), expect(e, row, col + 1), expect(e, row, col + 2), expect(e, row, col + 3)};
      const unsigned off = ((unsigned)(buf * a.rows + row) * COLS + col) * 4u;
      if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, val), ra, off, 0, SC1);
      else *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.act) + off) = val;
    }
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.flags + 32 * w, (unsigned)(e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- gather: one lane per flag polls
    if (tid < a.W) {
      int n = 0;
      while (__hip_atomic_load(a.flags + 32 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(e + 1)) {
        __builtin_amdgcn_s_sleep(1);
        ++nspin;
        if (++n > a.spin_limit) {  // a missing peer must not hang the box
          nerr += 1u << 20;
          break;
        }
      }
    }
    if (MODE == 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int v0 = 0; v0 < nvec; v0 += 256 * 8) {
      f32x4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int v = v0 + j * 256 + tid;
        const unsigned off = v < nvec ? ((unsigned)buf * a.rows * COLS + (unsigned)v * 4u) * 4u : 0xfffffff0u;
        if (MODE == 0) r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, SC1));
        else r[j] = v < nvec ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.act) + off) : f32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int v = v0 + j * 256 + tid;
        if (v < nvec) {
          *reinterpret_cast<f32x4*>(lds + 4 * v) = r[j];
          const int row = v / nvec_row, col = 4 * (v % nvec_row);
          if (col < a.W * cw)
            for (int q = 0; q < 4; ++q) nerr += r[j][q] != expect(e, row, col + q);
        }
      }
    }
    __syncthreads();
  }
  if (nerr) atomicAdd(a.errors, nerr);
  if (nspin) atomicAdd(a.spins, nspin);
}

// background load: streams a buffer on every CU the probe leaves free (and beside it on the probe's own CUs)
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n, int passes) {
  for (int p = 0; p < passes; ++p)
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      f32x4 v = src[i];
      v[0] += 1.f;
      dst[i] = v;
    }
}

int main() {
  hipStream_t s, s2;
  CK(hipStreamCreate(&s));
  CK(hipStreamCreate(&s2));
  unsigned *flags, *errors, *spins;
  float* act;
  CK(hipMalloc(&flags, 64 * 32 * 4));
  CK(hipMalloc(&errors, 4));
  CK(hipMalloc(&spins, 4));
  CK(hipMalloc(&act, 2 * 64 * COLS * 4));
  const size_t nbg = (size_t)64 << 20;  // 1 GiB + 1 GiB
  f32x4 *bsrc, *bdst;
  CK(hipMalloc(&bsrc, nbg * 16));
  CK(hipMalloc(&bdst, nbg * 16));
  CK(hipMemset(bsrc, 0, nbg * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("# one layer exchange of a column-split scan: W workgroups publish (rows x 208/W) slices, all gather rows x 208 floats\n");
  printf("# steps=49; us/exchange = launch time / (49 * nex); 'loaded' = a 2 GiB streaming kernel on another stream beside it\n");
  printf("# %-28s %4s %5s %4s %10s %12s %8s %10s\n", "form", "W", "rows", "nex", "us/launch", "us/exchange", "errors", "spins/poll");
  for (int loaded = 0; loaded < 2; ++loaded)
    for (int mode = 0; mode < 2; ++mode)
      for (int W : {13, 26, 52})
        for (int rows : {16, 64})
          for (int nex : {2, 3}) {
            if (mode == 1 && (nex == 3 || W == 52)) continue;
            Args a{act, flags, errors, spins, rows, W, 49, nex, 1 << 22};
            const size_t ldsb = (size_t)rows * COLS * 4;
            float best = 1e30f, sum = 0.f;
            unsigned herr = 0, hspin = 0;
            const int reps = 12;
            for (int rep = 0; rep < reps + 2; ++rep) {
              CK(hipMemsetAsync(flags, 0, 64 * 32 * 4, s));
              CK(hipMemsetAsync(errors, 0, 4, s));
              CK(hipMemsetAsync(spins, 0, 4, s));
              CK(hipStreamSynchronize(s));
              CK(hipEventRecord(e0, s));
              if (mode == 0) hipLaunchKernelGGL(exchange_kernel<0>, dim3(W), dim3(256), ldsb, s, a);
              else hipLaunchKernelGGL(exchange_kernel<1>, dim3(W), dim3(256), ldsb, s, a);
              CK(hipEventRecord(e1, s));
              if (loaded) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s2, bsrc, bdst, nbg, 2);
              CK(hipEventSynchronize(e1));
              CK(hipDeviceSynchronize());
              float ms;
              CK(hipEventElapsedTime(&ms, e0, e1));
              if (rep >= 2) {
                sum += ms;
                best = ms < best ? ms : best;
              }
              unsigned he, hs;
              CK(hipMemcpy(&he, errors, 4, hipMemcpyDeviceToHost));
              CK(hipMemcpy(&hs, spins, 4, hipMemcpyDeviceToHost));
              herr += he;
              hspin = hs;
            }
            const float avg = sum / reps * 1e3f;
            char form[64];
            snprintf(form, sizeof form, "%s %s", mode == 0 ? "sc1 stores+loads" : "plain+release/acquire", loaded ? "loaded" : "idle");
            printf("  %-28s %4d %5d %4d %10.1f %12.2f %8u %10.2f\n", form, W, rows, nex, avg, avg / (49 * nex), herr,
                   (double)hspin / (49.0 * nex * W * W));
            fflush(stdout);
          }
  return 0;
}
