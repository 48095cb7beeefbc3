// Probe: an fp32-ACCURATE GEMM on the bf16 matrix pipe ("bf16x6").
//
// gfx950 has no TF32 and its fp32 MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md: 64 vs 1024 FLOP/clk/SIMD).
// A float splits EXACTLY into three bf16 (8 significand bits each: a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1),
// a3 = bf16(a - a1 - a2); the subtractions are exact in fp32), every bf16 x bf16 product is exact in the MFMA's fp32
// accumulate, and of the nine cross products the six with i + j <= 4 carry everything down to 2^-24 |a||b|:
//     a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2) + O(2^-24 |ab|)
// Six v_mfma_f32_32x32x16_bf16 per 16 k = 6/16 of the fp32 MFMA's time for the same k: a 2.67x higher matrix ceiling at
// fp32 accuracy.  This probe measures (a) the error against an fp64 reference on sampled outputs, next to the 3-term
// variant (a1b1 + a1b2 + a2b1: ~2^-16) and (b) the time of C[M][N] = sum_k A[m][k] * B[n][k] (both operands
// k-contiguous fp32 in memory, split ON THE FLY while staging to LDS) for the decoder's big product
// (2450 x 3200 x 1024) and the other dense shapes of the update.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/bgemm_probe tools/probe/bgemm_probe.hip
//   tools/probe/bin/bgemm_probe            (prints the table committed as profiles/r04_bgemm_probe.txt)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// two floats -> their three bf16 parts, packed pairwise (low half = first element)
__device__ __forceinline__ void split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  const float r0 = x0 - __builtin_bit_cast(float, p1 << 16), r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float s0 = r0 - __builtin_bit_cast(float, p2 << 16), s1 = r1 - __builtin_bit_cast(float, p2 & 0xffff0000u);
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

#ifndef SCHED
#define SCHED 4
#endif
constexpr int BM = 256, BN = 128, BK = 16, NT = 512;
constexpr int ROWB = 48;                       // bytes per LDS row: 16 bf16 + 16 B pad (conflict-free ds_read_b128)
constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
constexpr int BUF = 3 * (A_PLANE + B_PLANE);   // bytes per stage buffer

struct Args {
  const float *A, *B;
  float* C;
  int M, N, K, lda, ldb, ldc;
};

// TERMS = 6 (fp32-accurate) or 3
template <int TERMS>
__global__ __launch_bounds__(NT) void bgemm_nt_kernel(Args p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware tile order: the 8 XCDs each walk a contiguous range of tiles, column tiles of a row block adjacent
  const int gx = (p.N + BN - 1) / BN, gy = (p.M + BM - 1) / BM, total = gx * gy;
  const int Lx = blockIdx.x, q = total >> 3, r = total & 7, xc = Lx & 7;
  const int t = xc * q + min(xc, r) + (Lx >> 3);
  const int m0 = (t / gx) * BM, n0 = (t % gx) * BN;

  const __amdgpu_buffer_rsrc_t ra = rsrc(p.A, 4u * (unsigned)((p.M - 1) * p.lda + p.K));
  const __amdgpu_buffer_rsrc_t rb = rsrc(p.B, 4u * (unsigned)((p.N - 1) * p.ldb + p.K));
  // staging roles: A 1024 vectors (row = v / 4, k-quad = v % 4): 2 per thread; B 512: 1 per thread
  unsigned aoff[2], boff;
  int alds[2], blds;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int v = tid + j * NT, row = v >> 2, kq = v & 3;
    aoff[j] = (m0 + row < p.M) ? 4u * (unsigned)((m0 + row) * p.lda + 4 * kq) : 0x80000000u;
    alds[j] = row * ROWB + kq * 8;
  }
  {
    const int row = tid >> 2, kq = tid & 3;
    boff = (n0 + row < p.N) ? 4u * (unsigned)((n0 + row) * p.ldb + 4 * kq) : 0x80000000u;
    blds = 3 * A_PLANE + row * ROWB + kq * 8;
  }
  f32x4 ga[2], gb;
  // TAIL = false: k0 + BK <= K is known (the main loop): no masks, no branches -- the loop body stays ONE basic block
  auto gload = [&](int k0, auto tail) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = !TAIL || k0 + 4 * ((tid + j * NT) & 3) < p.K;
      ga[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? aoff[j] + 4u * k0 : 0x80000000u, 0, 0));
    }
    const bool ok = !TAIL || k0 + 4 * (tid & 3) < p.K;
    gb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? boff + 4u * k0 : 0x80000000u, 0, 0));
    if (TAIL) {  // elements of a quad that straddles K belong to the next row
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (k0 + 4 * ((tid + j * NT) & 3) + e >= p.K) ga[j][e] = 0.f;
        if (k0 + 4 * (tid & 3) + e >= p.K) gb[e] = 0.f;
      }
    }
  };
  auto stage = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned a1, a2, a3, b1, b2, b3;
      split3(ga[j][0], ga[j][1], a1, a2, a3);
      split3(ga[j][2], ga[j][3], b1, b2, b3);
      *reinterpret_cast<u32x2*>(buf + alds[j]) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(buf + A_PLANE + alds[j]) = u32x2{a2, b2};
      *reinterpret_cast<u32x2*>(buf + 2 * A_PLANE + alds[j]) = u32x2{a3, b3};
    }
    unsigned a1, a2, a3, b1, b2, b3;
    split3(gb[0], gb[1], a1, a2, a3);
    split3(gb[2], gb[3], b1, b2, b3);
    *reinterpret_cast<u32x2*>(buf + blds) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(buf + B_PLANE + blds) = u32x2{a2, b2};
    *reinterpret_cast<u32x2*>(buf + 2 * B_PLANE + blds) = u32x2{a3, b3};
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nst = (p.K + BK - 1) / BK;
  const std::true_type TAILY{};
  const std::false_type TAILN{};
  gload(0, TAILY);
  stage(lds);
  if (nst > 1) gload(BK, TAILY);
  __syncthreads();
  const int afrag = (wm * 64 + li) * ROWB + lh * 16, bfrag = 3 * A_PLANE + (wn * 64 + li) * ROWB + lh * 16;
  // one stage: fragments of `cur`, then -- BETWEEN its MFMAs (a wave issues in order: after its 24 MFMAs it would
  // otherwise run ~100 vector instructions with the matrix pipe idle) -- the split + LDS stores of the next stage and
  // the global loads of the one after
  auto body = [&](int s, auto more1, auto more2, auto tail) __attribute__((always_inline)) {
    char* cur = lds + (s & 1) * BUF;
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        fa[i][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * A_PLANE + afrag + i * 32 * ROWB);
        fb[i][pl] = *reinterpret_cast<const bf16x8*>(cur + pl * B_PLANE + bfrag + i * 32 * ROWB);
      }
    if (decltype(more1)::value) stage(lds + ((s + 1) & 1) * BUF);
    if (decltype(more2)::value) gload((s + 2) * BK, tail);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        if (TERMS == 6) {  // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
#if SCHED
    if (decltype(more1)::value) {
#pragma unroll
      for (int g = 0; g < 4 * TERMS; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, SCHED, 0);   // a few vector ALU instructions of the split
        if (g % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // an LDS store
      }
    }
#endif
    __syncthreads();
  };
  int s = 0;
  for (; s + 3 < nst; ++s) body(s, TAILY, TAILY, TAILN);   // stage s + 2 <= nst - 2: a full stage
  for (; s < nst; ++s) {
    if (s + 2 < nst) body(s, TAILY, TAILY, TAILY);
    else if (s + 1 < nst) body(s, TAILY, TAILN, TAILN);
    else body(s, TAILN, TAILN, TAILN);
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < p.M && n < p.N) p.C[(size_t)m * p.ldc + n] = acc[i][j][e];
      }
    }
}

template <int TERMS>
static float run(const Args& a, int iters) {
  const int gx = (a.N + BN - 1) / BN, gy = (a.M + BM - 1) / BM;
  CK(hipFuncSetAttribute((const void*)bgemm_nt_kernel<TERMS>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(bgemm_nt_kernel<TERMS>, dim3(gx * gy), dim3(NT), 2 * BUF, 0, a);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(bgemm_nt_kernel<TERMS>, dim3(gx * gy), dim3(NT), 2 * BUF, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main() {
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {
      {2450, 3200, 1024, "decoder conv1 forward   h1 = h0 @ W1           (vgemm today: 0.53-0.64 of the fp32 peak)"},
      {2450, 1024, 3200, "decoder conv1 dgrad     dh0 = d1 @ W1^T"},
      {1024, 3200, 2450, "decoder conv1 wgrad     dW1 = h0^T @ d1        (operands pre-transposed)"},
      {2450, 1024, 230, "decoder fc1 forward"},
      {2450, 200, 1224, "hoisted posterior embed (vgemm today: 0.185)"},
      {34300, 200, 230, "head layer 1 over the rollout rows"},
      {4096, 4096, 4096, "square 4096 (reference point)"},
  };
  printf("# bf16x6 / bf16x3 GEMM probe: C[M][N] = sum_k A[m][k] B[n][k], fp32 in / out, split on the fly, tile %dx%dx%d\n", BM, BN, BK);
  printf("# err = max |c - c64| / (sum_k |a||b|) over 4096 sampled outputs (the fp32 dot product's own bound is ~K * 6e-8 of that)\n");
  printf("# %-6s %-6s %-6s | %9s %9s %8s | %9s %9s %8s | %s\n", "M", "N", "K", "x6 us", "TF(fp32)", "err", "x3 us", "TF(fp32)", "err", "shape");
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    const int lda = (K + 3) & ~3, ldb = lda;
    std::vector<float> hA((size_t)M * lda), hB((size_t)N * ldb);
    srand(1234);
    auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0) * (1.0f + (rand() % 7 == 0 ? 30.f : 0.f)); };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    Args a{dA, dB, dC, M, N, K, lda, ldb, N};
    std::vector<float> hC((size_t)M * N);
    double err[2];
    float us[2];
    for (int v = 0; v < 2; ++v) {
      CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
      us[v] = 1e3f * (v == 0 ? run<6>(a, 20) : run<3>(a, 20));
      CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0;
      srand(77);
      for (int sidx = 0; sidx < 4096; ++sidx) {
        const int m = sidx < 8 ? (sidx & 1 ? M - 1 : 0) : rand() % M, n = sidx < 8 ? (sidx & 2 ? N - 1 : 0) : rand() % N;
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) {
          const double x = (double)hA[(size_t)m * lda + k] * (double)hB[(size_t)n * ldb + k];
          ref += x;
          mag += fabs(x);
        }
        const double e = fabs((double)hC[(size_t)m * N + n] - ref) / mag;
        if (!(e <= worst)) worst = e;  // NaN-proof
      }
      err[v] = worst;
    }
    const double gf = 2.0 * M * N * K;
    printf("  %-6d %-6d %-6d | %9.1f %9.1f %8.1e | %9.1f %9.1f %8.1e | %s\n", M, N, K, us[0], gf / us[0] * 1e-6, err[0], us[1],
           gf / us[1] * 1e-6, err[1], sh.what);
    CK(hipFree(dA));
    CK(hipFree(dB));
    CK(hipFree(dC));
  }
  return 0;
}
