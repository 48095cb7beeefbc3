// Probe: what does the uconv inner loop (two 16-step v_mfma_f32_16x16x4_f32 chains sharing the A fragments, with the
// LDS read-add-write of the previous pair folded in) cost per MFMA, alone on a SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/uconv_chain.hip -o /tmp/uconv_chain && /tmp/uconv_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KST = 16, PLANE = 228;

template <int MODE>  // 0: MFMA only, same operands; 1: distinct operand registers; 2: + RMW spread; 3: + RMW clustered at the head
__global__ __launch_bounds__(256) void chain(float* out, const float* in, int iters) {
  __shared__ float planes[16 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += 256) planes[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a[KST], b0[KST], b1[KST];
#pragma unroll
  for (int s = 0; s < KST; ++s) {
    a[s] = in[(s * 64 + lane) % 1000];
    b0[s] = in[(s * 64 + lane + 7) % 1000];
    b1[s] = in[(s * 64 + lane + 13) % 1000];
  }
  char* pl = reinterpret_cast<char*>(planes);
  const int base = 4 * (wave * 16 * PLANE / 4 + (lane >> 4) * PLANE + (lane & 15));
  f32x4 pa0 = {0, 0, 0, 0}, pa1 = pa0;
  float po0[4] = {0, 0, 0, 0}, po1[4] = {0, 0, 0, 0};
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    float co0[4], co1[4];
    if (MODE == 3) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        *reinterpret_cast<float*>(pl + base + 4 * s * PLANE) = po0[s] + pa0[s];
        *reinterpret_cast<float*>(pl + base + 64 + 4 * s * PLANE) = po1[s] + pa1[s];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        co0[s] = *reinterpret_cast<const float*>(pl + base + 4 + 4 * s * PLANE);
        co1[s] = *reinterpret_cast<const float*>(pl + base + 68 + 4 * s * PLANE);
      }
    }
    if (MODE == 5) {
      float w0[4], w1[4];
#pragma unroll
      for (int s = 0; s < KST; ++s) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], c0, 0, 0, 0);
        if (s < 4) w0[s] = po0[s] + pa0[s];
        if (s == 4) *reinterpret_cast<f32x4*>(pl + 16 * (threadIdx.x)) = f32x4{w0[0], w0[1], w0[2], w0[3]};
        if (s == 6) {
          __builtin_amdgcn_wave_barrier();
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(pl + 16 * (threadIdx.x) + 32);
#pragma unroll
          for (int r = 0; r < 4; ++r) co0[r] = r0[r];
        }
        __builtin_amdgcn_sched_barrier(0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], c1, 0, 0, 0);
        if (s < 4) w1[s] = po1[s] + pa1[s];
        if (s == 4) *reinterpret_cast<f32x4*>(pl + 16 * (threadIdx.x) + 16384) = f32x4{w1[0], w1[1], w1[2], w1[3]};
        if (s == 6) {
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(pl + 16 * (threadIdx.x) + 16384 + 32);
#pragma unroll
          for (int r = 0; r < 4; ++r) co1[r] = r1[r];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      if (MODE == 0) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b1[0], c1, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], c1, 0, 0, 0);
      }
      if (MODE == 5) {  // as 4, but at most ONE filler instruction per MFMA gap (adds spread over 8 gaps)
        // handled below (the MFMAs of this step were already issued above, so fillers trail them by half a step)
      }
      if (MODE == 4) {  // rows of the accumulator contiguous in LDS: one 16-byte read / write per tile
        if (s == 0) {
          f32x4 w0 = {po0[0] + pa0[0], po0[1] + pa0[1], po0[2] + pa0[2], po0[3] + pa0[3]};
          *reinterpret_cast<f32x4*>(pl + 16 * (threadIdx.x)) = w0;
        }
        if (s == 1) {
          f32x4 w1 = {po1[0] + pa1[0], po1[1] + pa1[1], po1[2] + pa1[2], po1[3] + pa1[3]};
          *reinterpret_cast<f32x4*>(pl + 16 * (threadIdx.x) + 16384) = w1;
        }
        if (s == 4) __builtin_amdgcn_wave_barrier();
        if (s == 4) {
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(pl + 16 * (threadIdx.x) + 32);
#pragma unroll
          for (int r = 0; r < 4; ++r) co0[r] = r0[r];
        }
        if (s == 5) {
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(pl + 16 * (threadIdx.x) + 16384 + 32);
#pragma unroll
          for (int r = 0; r < 4; ++r) co1[r] = r1[r];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 2) {
        if (s < 4) {
          *reinterpret_cast<float*>(pl + base + 4 * s * PLANE) = po0[s] + pa0[s];
          *reinterpret_cast<float*>(pl + base + 64 + 4 * s * PLANE) = po1[s] + pa1[s];
        }
        if (s == 4) __builtin_amdgcn_wave_barrier();
        if (s >= 4 && s < 8) {
          co0[s - 4] = *reinterpret_cast<const float*>(pl + base + 4 + 4 * (s - 4) * PLANE);
          co1[s - 4] = *reinterpret_cast<const float*>(pl + base + 68 + 4 * (s - 4) * PLANE);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep each step's LDS work in its own MFMA gap
      }
    }
    pa0 = c0;
    pa1 = c1;
    if (MODE >= 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { po0[r] = co0[r]; po1[r] = co1[r]; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long t1 = clock64();
  float s = 0;
  for (int r = 0; r < 4; ++r) s += pa0[r] + pa1[r] + po0[r] + po1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 2.0f * KST);
}

int main() {
  float *d, *in;
  hipMalloc(&d, 1 << 22);
  hipMalloc(&in, 4096);
  hipMemset(in, 0, 4096);
  float h;
#define RUN(M, name, blocks)                                    \
  chain<M><<<blocks, 256>>>(d, in, 4000);                       \
  hipDeviceSynchronize();                                       \
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);                   \
  printf("%-70s %7.1f clk/mfma\n", name, h);
  RUN(0, "2 chains, same operand registers, 1 CU", 1)
  RUN(1, "2 chains, distinct A/B registers per step, 1 CU", 1)
  RUN(1, "2 chains, distinct A/B registers per step, 256 CUs", 256)
  RUN(2, "+ LDS read-add-write of the previous pair, one step per MFMA gap, 1 CU", 1)
  RUN(2, "+ LDS read-add-write ..., 256 CUs", 256)
  RUN(3, "+ LDS read-add-write clustered before the chain, 1 CU", 1)
  RUN(4, "+ LDS read-add-write as ONE 16-byte read + write per tile, 1 CU", 1)
  RUN(4, "+ LDS read-add-write as ONE 16-byte read + write per tile, 256 CUs", 256)
  RUN(5, "+ the same, at most one filler instruction per MFMA gap, 1 CU", 1)
  return 0;
}
