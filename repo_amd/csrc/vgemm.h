// Dense-operand tile GEMM on the fp32 matrix cores of gfx950, staged with VECTOR buffer loads.
//
//   C[m][n] = sum_k A(m,k) * B(k,n)      m in [0,M)  n in [0,N)  k in [kbeg,kend)
//
// Same tile / LDS / MFMA mapping as igemm.h (wave64 owns TM x TN tiles of 32x32 fed by
// v_mfma_f32_32x32x2_f32, LDS slices laid out [k][m] / [k][n], two LDS buffers, one barrier per
// slice), but the operands are plain 2-D arrays and every staging load moves VW consecutive floats
// per lane (buffer_load_dwordx4 / dwordx2) along the operand's memory-contiguous axis.
//
// Why: in-kernel stamps (tools/probe/igemm_stamps.hip) show the dword-granular igemm loop is bound
// by the CU's vector-memory front end, not by the matrix pipe: a wave-level dword load costs the
// texture addresser ~12 cycles when its 64 lanes are contiguous and 30-80 when they touch 4-10
// segments, so the 16 loads per thread per slice keep the four waves of a workgroup ~800-3900 cycles
// in the "issue" phase against 1000-2000 cycles of MFMA work.  One dwordx4 load replaces four.
//
// Buffer addressing (raw buffer, stride 0, num_records = bytes of the operand) also gives
//  * 32-bit per-lane offsets off an SGPR base (no 64-bit address arithmetic per load), and
//  * hardware zero-fill: a slice row beyond kend gets offset 0x80000000 (out of range -> 0.0), and a
//    vector that runs past the end of the array reads zeros instead of faulting.
// Rows m >= M / columns n >= N of a tile read real (finite) neighbouring data and are never stored.
#pragma once
#include <type_traits>

#include "igemm.h"

namespace repo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int VW>
struct VecLoad;
template <>
struct VecLoad<4> {
  typedef f32x4 type;
  static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  }
};
template <>
struct VecLoad<2> {
  typedef f32x2 type;
  static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
  }
};

constexpr unsigned kOobOffset = 0x80000000u;  // >= num_records of any operand (all are < 2 GiB)

// A dense operand: element (row r, col c) at p[r * ld + c], c contiguous.
struct Dense2D {
  const float* p;
  unsigned bytes;
  int ld;
};

// Op concept (dense):
//   static constexpr bool A_VK;   // A's contiguous axis is k  (A.p[m*ld + k]); else m (A.p[k*ld + m])
//   static constexpr bool B_VK;   // B's contiguous axis is k  (B.p[n*ld + k]); else n (B.p[k*ld + n])
//   static constexpr int  VW;     // 4 or 2: divides every ld, kbeg and kend-kbeg of a k-contiguous
//                                 // operand; base pointers aligned to 4*VW bytes
//   Dense2D A, B;  void init(int z);  int M(), N(), kbeg(), kend();
//   template <class V> void fix_b(V& v, int n0) const;   // hook on a loaded B vector (n-contiguous)
//   void store_col(int mb, int n, const f32x16& acc, int M);  void finish();
// An Op may own the mapping flat tile id -> (tile x, tile y, z): void decode(int t, int& bx, int& by, int& bz) const.
template <class Op, class = void>
struct HasTileDecode : std::false_type {};
template <class Op>
struct HasTileDecode<Op, std::void_t<decltype(&Op::decode)>> : std::true_type {};

template <class Op, class T>
__global__ __launch_bounds__(T::NT) void vgemm_kernel(Op op) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NT = T::NT, VW = Op::VW;
  constexpr bool AVK = Op::A_VK, BVK = Op::B_VK;
  // k-vector operands are written to LDS as VW ds_write_b32 by lanes (kv, row): row stride == 2 (mod 8)
  // keeps the 32 lanes of a half-wave on 32 banks; m/n-vector operands are written as one
  // ds_write_b128 / b64 per lane: row stride a multiple of VW.  MFMA fragment reads are conflict-free
  // for any stride (lanes l and l+32 never conflict on ds_read_b32).
  constexpr int LDA = AVK ? BM + 2 : BM + 4, LDB = BVK ? BN + 2 : BN + 4;
  constexpr int A_VROW = AVK ? BK / VW : BM / VW;  // vectors per tile row (k-vectors per m / m-vectors per k)
  constexpr int B_VROW = BVK ? BK / VW : BN / VW;
  constexpr int A_NV = BM * BK / VW, B_NV = BN * BK / VW;  // vectors per slice
  constexpr int A_PER = (A_NV + NT - 1) / NT, B_PER = (B_NV + NT - 1) / NT;
  static_assert(A_NV % NT == 0 || A_NV < NT, "A tile must divide over threads");
  static_assert(B_NV % NT == 0 || B_NV < NT, "B tile must divide over threads");
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (LDA + LDB)];
  float* As = lds;
  float* Bs = lds + 2 * BK * LDA;
  typedef typename VecLoad<VW>::type vec_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / T::WN, wn = wid % T::WN;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: dispatch slot L (dealt round-robin to the 8 XCDs) -> tile id T such that each
  // XCD walks a contiguous range of (x, y, z) tiles, x fastest: the column tiles of one row block (which
  // re-read the same A rows) and all tiles of one split-K slab share an L2 instead of each XCD fetching
  // the operand from HBM again.
  int bx, by, bz;
  {
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int q = total >> 3, r = total & 7, xc = L & 7;
    const int t = xc * q + min(xc, r) + (L >> 3);
    if constexpr (HasTileDecode<Op>::value) {
      op.decode(t, bx, by, bz);  // a dense 1-D grid over several problems (grouped weight gradients)
    } else {
      bx = t % gx;
      by = (t / gx) % gy;
      bz = t / (gx * gy);
    }
  }
  op.init(bz);
  const int M = op.M(), N = op.N();
  const int n0 = bx * BN, m0 = by * BM;
  if (m0 < M && n0 < N) {
    const int kbeg = op.kbeg(), kend = op.kend();
    const __amdgpu_buffer_rsrc_t ra_ = make_rsrc(op.A.p, op.A.bytes), rb_ = make_rsrc(op.B.p, op.B.bytes);

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging roles: vector j of this thread is tile vector v = tid + j*NT; v % VROW runs along
    // the contiguous axis.  Byte offsets at k = kbeg, advanced by a constant per slice.
    const bool a_act = (A_NV >= NT) || tid < A_NV, b_act = (B_NV >= NT) || tid < B_NV;
    unsigned aoff[A_PER], boff[B_PER];
    int a_kl[A_PER], b_kl[B_PER];  // slice-local k of the vector's first element
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      const int v = tid + j * NT, c = v % A_VROW, r = v / A_VROW;
      if (AVK) {  // r = m (clamped: rows >= M are never stored), c = k-vector
        a_kl[j] = c * VW;
        aoff[j] = 4u * (unsigned)(min(m0 + r, M - 1) * op.A.ld + kbeg + c * VW);
      } else {  // r = k, c = m-vector
        a_kl[j] = r;
        aoff[j] = 4u * (unsigned)((kbeg + r) * op.A.ld + m0 + c * VW);
      }
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int v = tid + j * NT, c = v % B_VROW, r = v / B_VROW;
      if (BVK) {
        b_kl[j] = c * VW;
        boff[j] = 4u * (unsigned)(min(n0 + r, N - 1) * op.B.ld + kbeg + c * VW);
      } else {
        b_kl[j] = r;
        boff[j] = 4u * (unsigned)((kbeg + r) * op.B.ld + n0 + c * VW);
      }
    }
    const unsigned a_step = 4u * (unsigned)(AVK ? BK : BK * op.A.ld);
    const unsigned b_step = 4u * (unsigned)(BVK ? BK : BK * op.B.ld);
    int b_n0[B_PER];  // first column of an n-vector (for fix_b)
#pragma unroll
    for (int j = 0; j < B_PER; ++j) b_n0[j] = BVK ? 0 : n0 + ((tid + j * NT) % B_VROW) * VW;

    vec_t ra[2][A_PER], rb[2][B_PER];

    // slice index t (0-based from kbeg); CK: the slice may run past kend -> those vectors read zeros
    auto gload = [&](int t, auto set_c, auto check_k) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      constexpr bool CK = decltype(check_k)::value;
      const int krem = kend - kbeg - t * BK;  // valid k in this slice (>= BK when !CK)
      if (a_act) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
          unsigned o = aoff[j] + (unsigned)t * a_step;
          if (CK) o = a_kl[j] < krem ? o : kOobOffset;
          ra[S][j] = VecLoad<VW>::load(ra_, o);
        }
      }
      if (b_act) {
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
          unsigned o = boff[j] + (unsigned)t * b_step;
          if (CK) o = b_kl[j] < krem ? o : kOobOffset;
          rb[S][j] = VecLoad<VW>::load(rb_, o);
        }
      }
    };
    const int nt = (kend - kbeg + BK - 1) / BK;
    const bool k_ragged = (kend - kbeg) % BK != 0;
    auto gload_any = [&](int t, auto set_c) __attribute__((always_inline)) {
      t = min(t, nt - 1);  // unconditional loads (see igemm.h): the tail re-reads the last slice
      if (k_ragged && t == nt - 1) gload(t, set_c, std::true_type{});
      else gload(t, set_c, std::false_type{});
    };
    auto lstore = [&](int buf, auto set_c) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      float* as = As + buf * BK * LDA;
      float* bs = Bs + buf * BK * LDB;
      if (a_act) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
          const int v = tid + j * NT, c = v % A_VROW, r = v / A_VROW;
          if (AVK) {
#pragma unroll
            for (int i = 0; i < VW; ++i) as[(c * VW + i) * LDA + r] = ra[S][j][i];
          } else {
            *reinterpret_cast<vec_t*>(as + r * LDA + c * VW) = ra[S][j];
          }
        }
      }
      if (b_act) {
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
          const int v = tid + j * NT, c = v % B_VROW, r = v / B_VROW;
          if (BVK) {
#pragma unroll
            for (int i = 0; i < VW; ++i) bs[(c * VW + i) * LDB + r] = rb[S][j][i];
          } else {
            vec_t x = rb[S][j];
            op.fix_b(x, b_n0[j]);
            *reinterpret_cast<vec_t*>(bs + r * LDB + c * VW) = x;
          }
        }
      }
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
      const float* as = As + buf * BK * LDA + wm * (T::TM * 32) + li;
      const float* bs = Bs + buf * BK * LDB + wn * (T::TN * 32) + li;
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
    REPO_STAMP_DECL
    if (nt > 0 && T::SETS == 1) {
      gload_any(0, S0{});
      lstore(0, S0{});
      __syncthreads();
      REPO_STAMP(5);
      int buf = 0;
      for (int t = 0; t < nt; ++t) {
        gload_any(t + 1, S0{});
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
      gload_any(0, S0{});
      lstore(0, S0{});
      gload_any(1, S1{});
      __syncthreads();
      REPO_STAMP(5);
      for (int t = 0; t < nt; t += 2) {
        gload_any(t + 2, S0{});
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
        gload_any(t + 3, S1{});
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
inline int launch_vgemm_flat(const Op& op, long blocks, hipStream_t s) {
  if (blocks <= 0) return REPO_OK;
  if (blocks > 2147483647L) return REPO_E_SHAPE;
  hipLaunchKernelGGL((vgemm_kernel<Op, T>), dim3((unsigned)blocks), dim3(T::NT), 0, s, op);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

template <class T, class Op>
inline int launch_vgemm(const Op& op, long M, long N, int Z, hipStream_t s) {
  if (M <= 0 || N <= 0 || Z <= 0) return REPO_OK;
  const long gx = (N + T::BN - 1) / T::BN, gy = (M + T::BM - 1) / T::BM;
  if (gx > 2147483647L || gy > 65535 || Z > 65535) return REPO_E_SHAPE;
  dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)Z);
  hipLaunchKernelGGL((vgemm_kernel<Op, T>), grid, dim3(T::NT), 0, s, op);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
