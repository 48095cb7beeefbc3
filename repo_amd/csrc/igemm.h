// LDS-tiled implicit GEMM on the fp32-input matrix cores of gfx950.
//
//   C[m][n] = sum_k A(m,k) * B(k,n)      m in [0,M)  n in [0,N)  k in [kbeg,kend)
//
// The operands are *functors*: an Op supplies element loaders and an epilogue, so the same
// tile engine serves the dense layers (NT/NN/TN), the strided convolutions (gather form),
// the transposed convolutions (parity-class form) and their weight gradients (split-K over
// images) without materialising im2col.
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
//
// Address generation is SEPARABLE: every operand element address is rowctx(m or n) (+)
// colctx(k).  The engine hoists whichever context is fixed for a thread out of the K loop
// (an m/n-major operand: the thread's single m/n, decoded once; a k-major operand: its
// A_PER/B_PER rows, decoded once, plus ONE k decode per slice) and, when all lanes of a
// wave share k, computes the k context on the scalar unit.  Loads are unconditional from a
// clamped in-range address followed by a select: a predicated load makes hipcc branch
// around every element and drain vmcnt per element (measured 2-5x slower).
#pragma once
#include <type_traits>

#include "common.h"

namespace repo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DREPO_IGEMM_STAMPS (tools/probe/igemm_stamps.hip only): per-phase cycle totals of the K loop,
// summed over waves: [0] load issue  [1] MFMA phase  [2] LDS write (incl. wait for the loads)
// [3] barrier  [4] epilogue  [5] prologue  [6] waves  [7] slices
#ifdef REPO_IGEMM_STAMPS
static __device__ unsigned long long g_igemm_stamps[8];
#define REPO_STAMP_DECL unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_readcyclecounter(); unsigned st_n = 0;
#define REPO_STAMP(i)                                            \
  do {                                                           \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    st_[i] += now_ - st_t;                                       \
    st_t = now_;                                                 \
  } while (0)
#else
#define REPO_STAMP_DECL
#define REPO_STAMP(i)
#endif

template <int WM_, int WN_, int TM_, int TN_, int BK_ = 16, int SETS_ = 2>
struct TileCfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, BK = BK_;
  // register staging sets: 2 = loads issued two slices ahead (launches with <= 1 workgroup per CU),
  // 1 = one slice ahead, 32-64 fewer VGPRs (compute-bound launches that rely on occupancy instead)
  static constexpr int SETS = SETS_;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static constexpr int NT = WM * WN * 64;
  static_assert(BK % 2 == 0, "BK must be even (mfma 32x32x2)");
  static_assert((BM * BK) % NT == 0 && (BN * BK) % NT == 0, "tile must divide over threads");
  static_assert(NT % BM == 0 && NT % BN == 0 && NT % BK == 0, "one row context per thread");
};

using T128x128 = TileCfg<2, 2, 2, 2>;
using T128x128s1 = TileCfg<2, 2, 2, 2, 16, 1>;
using T64x128s1 = TileCfg<2, 2, 1, 2, 16, 1>;
using T64x128 = TileCfg<2, 2, 1, 2>;
using T64x64 = TileCfg<2, 2, 1, 1>;
using T32x128 = TileCfg<1, 4, 1, 1>;
using T32x256 = TileCfg<1, 4, 1, 2>;
using T128x64 = TileCfg<2, 2, 2, 1>;
using T64x64k32 = TileCfg<2, 2, 1, 1, 32>;
using T32x64k32 = TileCfg<1, 2, 1, 1, 32>;
using T32x128k32 = TileCfg<1, 4, 1, 1, 32>;
using T64x128k32 = TileCfg<2, 2, 1, 2, 32>;

// Op concept:
//   __device__ void  init(int z);                 // per-blockIdx.z setup (class / split)
//   __device__ int   M() const, N() const;        // logical extents for this z
//   __device__ int   kbeg() const, kend() const;
//   types AM, AK, BN, BK (trivially copyable contexts)
//   __device__ AM a_m(int m) const;  AK a_k(int k) const;  float a(const AM&, const AK&) const;
//   __device__ BN b_n(int n) const;  BK b_k(int k) const;  float b(const BK&, const BN&) const;
//        indices handed to a_m/a_k/b_n/b_k are always in range; a()/b() must be branch-free
//   __device__ void  store_col(int mb, int n, const f32x16& acc, int M);  // rows mb+(r&3)+8*(r>>2) < M
//   __device__ void  finish();                    // after the epilogue (block-level reductions)
//   static constexpr bool A_KMAJOR, B_KMAJOR;     // staging thread order: k fastest (source is
//                                                 // k-contiguous) or m/n fastest
template <int ROWS, int NT>
__device__ __forceinline__ int uniform_div(int tid) {
  // tid / ROWS; wave-uniform (hence scalar) when a wave of 64 lanes cannot straddle two values
  if (ROWS % 64 == 0) return __builtin_amdgcn_readfirstlane(tid / ROWS);
  return tid / ROWS;
}

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

    // ---- per-thread staging roles and hoisted row contexts
    // k-major: kk = tid % BK fixed, rows r_j = tid / BK + j * (NT / BK)
    // m-major: row = tid % BM fixed, kk_j = tid / BM + j * (NT / BM)
    constexpr int A_NCTX = Op::A_KMAJOR ? A_PER : 1;
    constexpr int B_NCTX = Op::B_KMAJOR ? B_PER : 1;
    typename Op::AM am[A_NCTX];
    typename Op::BN bn[B_NCTX];
    unsigned a_ok = 0, b_ok = 0;  // bit j: row of element j is inside the matrix
    const int a_kk0 = Op::A_KMAJOR ? (tid % BK) : uniform_div<BM, NT>(tid);
    const int b_kk0 = Op::B_KMAJOR ? (tid % BK) : uniform_div<BN, NT>(tid);
#pragma unroll
    for (int j = 0; j < A_NCTX; ++j) {
      const int mm = Op::A_KMAJOR ? (tid / BK + j * (NT / BK)) : (tid % BM);
      const int m = m0 + mm;
      am[j] = op.a_m(min(m, M - 1));
      a_ok |= (m < M ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < B_NCTX; ++j) {
      const int nn = Op::B_KMAJOR ? (tid / BK + j * (NT / BK)) : (tid % BN);
      const int n = n0 + nn;
      bn[j] = op.b_n(min(n, N - 1));
      b_ok |= (n < N ? 1u : 0u) << j;
    }

    // Two register staging sets: while slice t is being multiplied out of LDS, slice t+1 sits
    // in one set (in flight or landed) and the loads of slice t+2 are issued into the other,
    // so a global load has two MFMA phases (>= 1000 cycles) to land before its LDS write.
    // Needed where a launch has <= 1 workgroup per CU (the 2450-row dense layers of the
    // imagination rollout: one wave per SIMD, nothing else to hide HBM/L2 latency behind).
    // Raw loaded values + validity bitmasks per set.  The zero-fill select is applied when the
    // slice is written to LDS, NOT at load time: a select on the loaded value right after the
    // load makes hipcc wait for the load immediately and serialises the whole pipeline.
    float ra[2][A_PER], rb[2][B_PER];
    unsigned amask[2] = {0u, 0u}, bmask[2] = {0u, 0u};

    // CK = false: the whole slice [kt, kt+BK) is inside [kbeg, kend) -> no k checks at all
    auto gload = [&](int kt, auto set_c, auto check_k) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      constexpr bool CK = decltype(check_k)::value;
      if (Op::A_KMAJOR) {
        const int k = kt + a_kk0;
        const typename Op::AK ak = op.a_k(CK ? min(k, kend - 1) : k);
        amask[S] = (!CK || k < kend) ? a_ok : 0u;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) ra[S][j] = op.a(am[j], ak);
      } else {
        unsigned mk = 0;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
          const int k = kt + a_kk0 + j * (NT / BM);
          ra[S][j] = op.a(am[0], op.a_k(CK ? min(k, kend - 1) : k));
          mk |= ((!CK || k < kend) ? 1u : 0u) << j;
        }
        amask[S] = (a_ok & 1u) ? mk : 0u;
      }
      if (Op::B_KMAJOR) {
        const int k = kt + b_kk0;
        const typename Op::BK bk = op.b_k(CK ? min(k, kend - 1) : k);
        bmask[S] = (!CK || k < kend) ? b_ok : 0u;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) rb[S][j] = op.b(bk, bn[j]);
      } else {
        unsigned mk = 0;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
          const int k = kt + b_kk0 + j * (NT / BN);
          rb[S][j] = op.b(op.b_k(CK ? min(k, kend - 1) : k), bn[0]);
          mk |= ((!CK || k < kend) ? 1u : 0u) << j;
        }
        bmask[S] = (b_ok & 1u) ? mk : 0u;
      }
    };
    auto gload_any = [&](int kt, auto set_c) __attribute__((always_inline)) {
      if (kt + BK <= kend) gload(kt, set_c, std::false_type{});
      else gload(kt, set_c, std::true_type{});
    };
    // interior workgroups (whole tile inside M x N, K a multiple of BK) skip the zero-fill selects
    const bool interior = (m0 + BM <= M) && (n0 + BN <= N) && ((kend - kbeg) % BK == 0);
    auto lstore = [&](int buf, auto set_c) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      float* as = As + buf * BK * LDA;
      float* bs = Bs + buf * BK * LDB;
      if (interior) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
          const int kk = Op::A_KMAJOR ? (tid % BK) : (tid / BM + j * (NT / BM));
          const int mm = Op::A_KMAJOR ? (tid / BK + j * (NT / BK)) : (tid % BM);
          as[kk * LDA + mm] = ra[S][j];
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
          const int kk = Op::B_KMAJOR ? (tid % BK) : (tid / BN + j * (NT / BN));
          const int nn = Op::B_KMAJOR ? (tid / BK + j * (NT / BK)) : (tid % BN);
          bs[kk * LDB + nn] = rb[S][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
          const int kk = Op::A_KMAJOR ? (tid % BK) : (tid / BM + j * (NT / BM));
          const int mm = Op::A_KMAJOR ? (tid / BK + j * (NT / BK)) : (tid % BM);
          as[kk * LDA + mm] = ((amask[S] >> j) & 1u) ? ra[S][j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
          const int kk = Op::B_KMAJOR ? (tid % BK) : (tid / BN + j * (NT / BN));
          const int nn = Op::B_KMAJOR ? (tid / BK + j * (NT / BK)) : (tid % BN);
          bs[kk * LDB + nn] = ((bmask[S] >> j) & 1u) ? rb[S][j] : 0.f;
        }
      }
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
      const float* as = As + buf * BK * LDA + wm * (T::TM * 32) + li;
      const float* bs = Bs + buf * BK * LDB + wn * (T::TN * 32) + li;
      // All operand fragments of the slice are read from LDS up front, then the MFMA chain runs
      // with counted lgkmcnt waits: with one wave per SIMD (small launches) a read->wait->MFMA
      // sequence per k-step exposes the ~130-cycle LDS latency on every 64-cycle MFMA.
      float av[BK / 2][T::TM], bv[BK / 2][T::TN];
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i) av[ks][i] = as[(ks * 2 + lh) * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) bv[ks][j] = bs[(ks * 2 + lh) * LDB + j * 32];
      }
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks)
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
          for (int j = 0; j < T::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][i], bv[ks][j], acc[i][j], 0, 0, 0);
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nt = (kend - kbeg + BK - 1) / BK;
    // Loads and LDS writes are issued UNCONDITIONALLY (slice index clamped to the last slice):
    // hipcc's s_waitcnt insertion is static, so a load that is skipped on some path makes every
    // later counted wait collapse to vmcnt(0) and drains the prefetch.  Re-loading the last
    // slice once or twice at the tail is cheaper than losing the overlap everywhere.
    auto slice_k = [&](int t) __attribute__((always_inline)) { return kbeg + min(t, nt - 1) * BK; };
    REPO_STAMP_DECL
    if (nt > 0 && T::SETS == 1) {
      gload_any(slice_k(0), S0{});
      lstore(0, S0{});
      __syncthreads();
      REPO_STAMP(5);
      int buf = 0;
      for (int t = 0; t < nt; ++t) {
        gload_any(slice_k(t + 1), S0{});
        __builtin_amdgcn_sched_barrier(0);
        REPO_STAMP(0);
        compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        REPO_STAMP(1);
        lstore(buf ^ 1, S0{});
        REPO_STAMP(2);
        __syncthreads();
        REPO_STAMP(3);
        buf ^= 1;
      }
    } else if (nt > 0) {
      gload_any(slice_k(0), S0{});
      lstore(0, S0{});
      gload_any(slice_k(1), S1{});
      __syncthreads();
      REPO_STAMP(5);
      for (int t = 0; t < nt; t += 2) {
        // even slice t: LDS buffer 0; slice t+1 is in set 1; issue slice t+2 into set 0
        gload_any(slice_k(t + 2), S0{});
        // coarse fences: keep the LDS-write selects of the landed set BEHIND the MFMA phase and
        // the new loads AHEAD of it (hipcc otherwise hoists the selects above the load issue)
        __builtin_amdgcn_sched_barrier(0);
        REPO_STAMP(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        REPO_STAMP(1);
        lstore(1, S1{});
        REPO_STAMP(2);
        __syncthreads();
        REPO_STAMP(3);
        if (t + 1 >= nt) break;
        // odd slice t+1: LDS buffer 1; slice t+2 is in set 0; issue slice t+3 into set 1
        gload_any(slice_k(t + 3), S1{});
        __builtin_amdgcn_sched_barrier(0);
        REPO_STAMP(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        REPO_STAMP(1);
        lstore(0, S0{});
        REPO_STAMP(2);
        __syncthreads();
        REPO_STAMP(3);
      }
    }

    // Epilogue: lane (li, lh) of tile (i, j) holds column n and the 16 rows
    // mb + (r & 3) + 8 * (r >> 2), r = 0..15.  The Op gets the whole column at once so that
    // everything that depends on n only (bias, pixel decode, base pointers) is computed once,
    // not 16 times: for the small dense layers the per-element epilogue used to cost more
    // than the K loop.
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        const int n = n0 + (wn * T::TN + j) * 32 + li;
        const int mb = m0 + (wm * T::TM + i) * 32 + 4 * lh;
        if (n < N && mb < M) op.store_col(mb, n, acc[i][j], M);
      }
#ifdef REPO_IGEMM_STAMPS
    REPO_STAMP(4);
    if (lane == 0 && blockIdx.x % 16 == 0) {
      for (int i = 0; i < 6; ++i) atomicAdd(&g_igemm_stamps[i], st_[i]);
      atomicAdd(&g_igemm_stamps[6], 1ull);
      atomicAdd(&g_igemm_stamps[7], (unsigned long long)nt);
    }
#endif
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
