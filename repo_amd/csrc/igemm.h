// LDS-tiled implicit GEMM on the fp32-input matrix cores of gfx950.
//
//   C[m][n] = sum_k A(m,k) * B(k,n)      m in [0,M)  n in [0,N)  k in [kbeg,kend)
//
// The operands are *functors*: an Op supplies element loaders a(m,k), b(k,n) and an
// epilogue store(m,n,acc), so the same tile engine serves the dense layers (NT/NN/TN),
// the strided convolutions (gather form), the transposed convolutions (parity-class
// form) and their weight gradients (split-K over images) without materialising im2col.
//
// Mapping to the hardware:
//  * one wave64 owns TM x TN accumulator tiles of 32x32, each fed by
//    v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate; A operand lane l
//    holds A[i=l&31][k=l>>5], B operand holds B[k=l>>5][j=l&31], C/D register r of
//    lane l is row (r&3)+8*(r>>2)+4*(l>>5), column l&31).
//  * a workgroup of WM x WN waves stages a BM x BK slice of A and a BK x BN slice of B
//    in LDS, laid out [k][m] / [k][n] with a row stride == 2 (mod 32) dwords so that both
//    the k-major and the m-major staging writes and the MFMA operand reads are free of
//    bank conflicts (ds_write_b32 / ds_read_b32 bank = dword address mod 32).
//  * two LDS buffers; global loads of slice t+1 are issued into registers before the
//    MFMAs of slice t and written to the other buffer afterwards: one barrier per slice.
//  * lanes run along n in the epilogue, so Ops are arranged with the memory-contiguous
//    output index on n (pixels for NCHW activations, features for row-major matrices).
#pragma once
#include "common.h"

namespace repo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM_, int WN_, int TM_, int TN_, int BK_ = 16>
struct TileCfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, BK = BK_;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static constexpr int NT = WM * WN * 64;
  static_assert(BK % 2 == 0, "BK must be even (mfma 32x32x2)");
  static_assert((BM * BK) % NT == 0 && (BN * BK) % NT == 0, "tile must divide over threads");
};

using T128x128 = TileCfg<2, 2, 2, 2>;
using T64x128 = TileCfg<2, 2, 1, 2>;
using T64x64 = TileCfg<2, 2, 1, 1>;
using T32x128 = TileCfg<1, 4, 1, 1>;
using T32x256 = TileCfg<1, 4, 1, 2>;
using T128x64 = TileCfg<2, 2, 2, 1>;
using T256x64 = TileCfg<4, 1, 2, 2>;

// Op concept:
//   __device__ void  init(int z);                 // per-blockIdx.z setup (class / split)
//   __device__ int   M() const, N() const;        // logical extents for this z
//   __device__ int   kbeg() const, kend() const;
//   __device__ float a(int m, int k) const;       // 0 <= m < M, kbeg <= k < kend
//   __device__ float b(int k, int n) const;
//   __device__ void  store(int m, int n, float v);
//   __device__ void  finish();                    // after the epilogue (block-level reductions)
//   static constexpr bool A_KMAJOR, B_KMAJOR;     // staging thread order: k fastest (source is
//                                                 // k-contiguous) or m/n fastest
template <class Op, class T>
__global__ __launch_bounds__(T::NT) void igemm_kernel(Op op) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NT = T::NT;
  constexpr int LDA = BM + 2, LDB = BN + 2;
  constexpr int A_PER = BM * BK / NT, B_PER = BN * BK / NT;
  __shared__ float lds[2 * BK * (LDA + LDB)];
  float* As = lds;
  float* Bs = lds + 2 * BK * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / T::WN, wn = wid % T::WN;
  const int li = lane & 31, lh = lane >> 5;

  op.init(blockIdx.z);
  const int M = op.M(), N = op.N();
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  if (m0 < M && n0 < N) {
    const int kbeg = op.kbeg(), kend = op.kend();

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[A_PER], rb[B_PER];

    auto gload = [&](int kt) {
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        const int e = tid + j * NT;
        const int kk = Op::A_KMAJOR ? (e % BK) : (e / BM);
        const int mm = Op::A_KMAJOR ? (e / BK) : (e % BM);
        const int m = m0 + mm, k = kt + kk;
        ra[j] = (m < M && k < kend) ? op.a(m, k) : 0.f;
      }
#pragma unroll
      for (int j = 0; j < B_PER; ++j) {
        const int e = tid + j * NT;
        const int kk = Op::B_KMAJOR ? (e % BK) : (e / BN);
        const int nn = Op::B_KMAJOR ? (e / BK) : (e % BN);
        const int n = n0 + nn, k = kt + kk;
        rb[j] = (n < N && k < kend) ? op.b(k, n) : 0.f;
      }
    };
    auto lstore = [&](int buf) {
      float* as = As + buf * BK * LDA;
      float* bs = Bs + buf * BK * LDB;
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        const int e = tid + j * NT;
        const int kk = Op::A_KMAJOR ? (e % BK) : (e / BM);
        const int mm = Op::A_KMAJOR ? (e / BK) : (e % BM);
        as[kk * LDA + mm] = ra[j];
      }
#pragma unroll
      for (int j = 0; j < B_PER; ++j) {
        const int e = tid + j * NT;
        const int kk = Op::B_KMAJOR ? (e % BK) : (e / BN);
        const int nn = Op::B_KMAJOR ? (e / BK) : (e % BN);
        bs[kk * LDB + nn] = rb[j];
      }
    };

    int buf = 0;
    if (kbeg < kend) {
      gload(kbeg);
      lstore(0);
    }
    __syncthreads();
    for (int kt = kbeg; kt < kend; kt += BK) {
      const bool more = kt + BK < kend;
      if (more) gload(kt + BK);
      const float* as = As + buf * BK * LDA + wm * (T::TM * 32) + li;
      const float* bs = Bs + buf * BK * LDB + wn * (T::TN * 32) + li;
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        float av[T::TM], bv[T::TN];
#pragma unroll
        for (int i = 0; i < T::TM; ++i) av[i] = as[(ks * 2 + lh) * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) bv[j] = bs[(ks * 2 + lh) * LDB + j * 32];
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
          for (int j = 0; j < T::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
      if (more) lstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }

#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        const int n = n0 + (wn * T::TN + j) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * T::TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M && n < N) op.store(m, n, acc[i][j][r]);
        }
      }
  }
  op.finish();
}

template <class T, class Op>
inline int launch_igemm(const Op& op, long M, long N, int Z, hipStream_t s) {
  if (M <= 0 || N <= 0 || Z <= 0) return REPO_OK;
  const long gx = (N + T::BN - 1) / T::BN, gy = (M + T::BM - 1) / T::BM;
  if (gx > 2147483647L || gy > 65535 || Z > 65535) return REPO_E_SHAPE;
  dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)Z);
  hipLaunchKernelGGL((igemm_kernel<Op, T>), grid, dim3(T::NT), 0, s, op);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
