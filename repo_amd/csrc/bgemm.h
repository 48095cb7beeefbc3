// fp32-ACCURATE dense products on the bf16 matrix pipe of gfx950 ("bf16x6").
//
// gfx950 has no TF32, and its fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate (64 against 1024
// FLOP/clk/SIMD, MI355X_MICROARCH.md).  A float splits EXACTLY into three bf16 (8 significand bits each):
//     a = a1 + a2 + a3,   a1 = bf16(a),  a2 = bf16(a - a1),  a3 = bf16(a - a1 - a2)     (the subtractions are exact)
// every bf16 x bf16 product is exact in the MFMA's fp32 accumulation, and of the nine cross products the six with
// i + j <= 4 carry everything down to 2^-24 |a||b|:
//     a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2) + O(2^-24 |ab|)
// Six v_mfma_f32_32x32x16_bf16 per 16 k take 6/16 of the fp32 MFMA's time for the same k: a 2.67x higher matrix
// ceiling at the SAME accuracy (measured against fp64: tools/probe/bgemm_probe.hip -> profiles/r04_bgemm_probe.txt,
// tests/test_ops_gpu.py::test_bgemm_*: the error of this engine is at or below the fp32-MFMA engine's on the same
// operands; both are dominated by the fp32 accumulation over K).  Inputs, outputs and accumulation stay fp32: the
// split is a way of feeding the multiplier array, not a reduced-precision format -- nothing is rounded to bf16.
// Non-finite inputs: inf - inf in the split gives NaN where fp32 arithmetic would give inf (activations are finite).
//
//   C[m][n] (+)= epi( sum_k A(m,k) B(k,n) + bias[n / bias_div] )
// Operands are plain fp32 2-D arrays, each either k-contiguous (A[m][k] / B[n][k]) or m/n-contiguous (A[k][m] /
// B[k][n]); they are split ON THE FLY while being staged: 16-byte global loads -> registers -> three bf16 planes in
// LDS, rows of BK bf16 (+16 B pad: the fragment reads, one ds_read_b128 per lane and plane = 8 consecutive k, are
// conflict-free).  An m/n-contiguous operand is transposed on the way: a thread owns a 4(k) x 4(m) micro-tile (four
// 16-byte loads) and writes four rows of 4 bf16.  Two LDS buffers, ONE barrier per BK; the global loads run TWO stages
// ahead of their use (two register sets: one workgroup of 8 waves per CU has nothing else to hide them behind).
// 8 waves as 4 (m) x 2 (n); the wave's TM x TN tiles of 32 x 32 share fragments (6 MFMAs per tile and 16 k).
#pragma once
#include <type_traits>

#include "vgemm.h"

namespace repo {

typedef float bg_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bg_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned bg_u32x2 __attribute__((ext_vector_type(2)));

// two floats -> their three bf16 parts, packed pairwise (low half = first element)
__device__ __forceinline__ void bg_split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
#ifdef BG_NO_SPLIT   // ablation build (tools/build_variant.sh; profiles/r05_split_ablation.txt): no split arithmetic in ANY
  p1 = __builtin_bit_cast(unsigned, x0), p2 = __builtin_bit_cast(unsigned, x1), p3 = p1 ^ p2;   // kernel -- results wrong, time meaningful
  return;
#endif
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(bg_f32x2{x0, x1}, bg_bf16x2));
  const float r0 = x0 - __builtin_bit_cast(float, p1 << 16), r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(bg_f32x2{r0, r1}, bg_bf16x2));
  const float s0 = r0 - __builtin_bit_cast(float, p2 << 16), s1 = r1 - __builtin_bit_cast(float, p2 & 0xffff0000u);
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(bg_f32x2{s0, s1}, bg_bf16x2));
}

struct BgArgs {
  Dense2D A, B;  // A: [M][K] if A_KC else [K][M];  B: [N][K] if B_KC else [K][N]
  const float* bias;
  const float* aux;
  float* C;
  int ldc, ldaux, bias_div;
  int M, N, K;
  int epi, accumulate;
};

template <int BM_, int BN_, int BK_>
struct BgTile {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, NT = 512;
  static constexpr int TM = BM / 128, TN = BN / 64;   // 4 x 2 waves
  static constexpr int ROWB = 2 * BK + 16;            // bytes per LDS row
  static constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
  static constexpr int BUF = 3 * (A_PLANE + B_PLANE);
  // behind the two stage buffers: where threads without a staging unit write (no branch in the loop body)
  static constexpr int DUMP = 4 * ROWB + 3 * (A_PLANE > B_PLANE ? A_PLANE : B_PLANE);
  static constexpr int LDS_BYTES = 2 * BUF + DUMP;
  static_assert(BK == 16 || BK == 32, "one or two MFMA k-steps per stage");
  static_assert(BM % 128 == 0 && BN % 64 == 0, "tile / wave grid mismatch");
};

// staging of ONE operand tile (ROWS x BK): per thread NV 16-byte vectors
template <int ROWS, int BK, bool KC>
struct BgStage {
  // k-contiguous: vector v = (row = v / (BK/4), k-quad = v % (BK/4))
  // row-contiguous: micro-tile u = (k-quad = u / (ROWS/4), row-quad = u % (ROWS/4)), 4 vectors (k .. k+3) each
  static constexpr int UNITS = ROWS * BK / (KC ? 4 : 16);
  static constexpr int PER = (UNITS + 511) / 512;          // units per thread
  static constexpr int NV = KC ? PER : 4 * PER;            // vectors per thread
};

template <class T, bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) void bgemm_kernel(BgArgs p) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, TM = T::TM, TN = T::TN, ROWB = T::ROWB;
  constexpr int A_PLANE = T::A_PLANE, B_PLANE = T::B_PLANE, BUF = T::BUF;
  typedef BgStage<BM, BK, A_KC> SA;
  typedef BgStage<BN, BK, B_KC> SB;
  extern __shared__ __attribute__((aligned(16))) char bg_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware tile order (vgemm.h): every XCD walks a contiguous range of tiles, the column tiles of a row block first
  const int gx = (p.N + BN - 1) / BN, gy = (p.M + BM - 1) / BM, total = gx * gy;
  const int q = total >> 3, r = total & 7, xc = blockIdx.x & 7;
  const int t = xc * q + min(xc, r) + (int)(blockIdx.x >> 3);
  const int m0 = (t / gx) * BM, n0 = (t % gx) * BN;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A.p, p.A.bytes), rb = make_rsrc(p.B.p, p.B.bytes);

  // ---- staging roles (byte offsets at k = 0; kOobOffset = out of range -> zeros).  A thread without a unit keeps
  // loading zeros and writes them to the dump slot behind the buffers: the loop body has no branch
  unsigned aoff[SA::PER], boff[SB::PER];
  int alds[SA::PER], blds[SB::PER], akq[SA::PER], bkq[SB::PER];
  bool aact[SA::PER], bact[SB::PER];
#pragma unroll
  for (int j = 0; j < SA::PER; ++j) {
    const int u = tid + j * 512;
    const bool act = aact[j] = (SA::UNITS % 512 == 0) || u < SA::UNITS;
    if (A_KC) {
      const int row = u / (BK / 4), kq = u % (BK / 4);
      akq[j] = 4 * kq;
      aoff[j] = (act && m0 + row < p.M) ? 4u * (unsigned)((m0 + row) * p.A.ld + 4 * kq) : kOobOffset;
      alds[j] = row * ROWB + kq * 8;
    } else {
      const int kq = u / (BM / 4), rq = u % (BM / 4);
      akq[j] = 4 * kq;
      // rows past M inside a quad read the neighbouring (finite) data and are never stored; a quad wholly past M or
      // past the end of the array reads zeros
      aoff[j] = (act && m0 + 4 * rq < p.M) ? 4u * (unsigned)(4 * kq * p.A.ld + m0 + 4 * rq) : kOobOffset;
      alds[j] = 4 * rq * ROWB + kq * 8;
    }
  }
#pragma unroll
  for (int j = 0; j < SB::PER; ++j) {
    const int u = tid + j * 512;
    const bool act = bact[j] = (SB::UNITS % 512 == 0) || u < SB::UNITS;
    if (B_KC) {
      const int row = u / (BK / 4), kq = u % (BK / 4);
      bkq[j] = 4 * kq;
      boff[j] = (act && n0 + row < p.N) ? 4u * (unsigned)((n0 + row) * p.B.ld + 4 * kq) : kOobOffset;
      blds[j] = 3 * A_PLANE + row * ROWB + kq * 8;
    } else {
      const int kq = u / (BN / 4), rq = u % (BN / 4);
      bkq[j] = 4 * kq;
      boff[j] = (act && n0 + 4 * rq < p.N) ? 4u * (unsigned)(4 * kq * p.B.ld + n0 + 4 * rq) : kOobOffset;
      blds[j] = 3 * A_PLANE + 4 * rq * ROWB + kq * 8;
    }
  }
  char* const dump = bg_lds + 2 * BUF;

  f32x4 ga[2][SA::NV], gb[2][SB::NV];  // two register sets: stage parity
  // TAIL = false: k0 + BK <= K is known (the main loop): no masks, no selects on k
  auto gload = [&](auto setc, int k0, auto tailc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    constexpr bool TAIL = decltype(tailc)::value;
#pragma unroll
    for (int j = 0; j < SA::PER; ++j) {
      if (A_KC) {
        const int k = k0 + akq[j];
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            ra, (!TAIL || k < p.K) ? aoff[j] + 4u * k0 : kOobOffset, 0, 0));
        if (TAIL) {  // a quad that straddles K holds elements of the next row
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = k + e < p.K ? v[e] : 0.f;
        }
        ga[set][j] = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k0 + akq[j] + e;
          ga[set][4 * j + e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              ra, (!TAIL || k < p.K) ? aoff[j] + 4u * (unsigned)((k0 + e) * p.A.ld) : kOobOffset, 0, 0));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < SB::PER; ++j) {
      if (B_KC) {
        const int k = k0 + bkq[j];
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rb, (!TAIL || k < p.K) ? boff[j] + 4u * k0 : kOobOffset, 0, 0));
        if (TAIL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = k + e < p.K ? v[e] : 0.f;
        }
        gb[set][j] = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k0 + bkq[j] + e;
          gb[set][4 * j + e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              rb, (!TAIL || k < p.K) ? boff[j] + 4u * (unsigned)((k0 + e) * p.B.ld) : kOobOffset, 0, 0));
        }
      }
    }
  };
  // one unit -> LDS: k-contiguous: a vector is 4 k of one row; row-contiguous: vectors e = 0..3 are k + e of 4 rows
  auto put4 = [&](char* base, int plane_bytes, float x0, float x1, float x2, float x3) __attribute__((always_inline)) {
    unsigned a1, a2, a3, b1, b2, b3;
    bg_split3(x0, x1, a1, a2, a3);
    bg_split3(x2, x3, b1, b2, b3);
    *reinterpret_cast<bg_u32x2*>(base) = bg_u32x2{a1, b1};
    *reinterpret_cast<bg_u32x2*>(base + plane_bytes) = bg_u32x2{a2, b2};
    *reinterpret_cast<bg_u32x2*>(base + 2 * plane_bytes) = bg_u32x2{a3, b3};
  };
  auto stage = [&](auto setc, char* buf) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < SA::PER; ++j) {
      char* dst = (SA::UNITS % 512 == 0 || aact[j]) ? buf + alds[j] : dump;
      if (A_KC) {
        put4(dst, A_PLANE, ga[set][j][0], ga[set][j][1], ga[set][j][2], ga[set][j][3]);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          put4(dst + rr * ROWB, A_PLANE, ga[set][4 * j][rr], ga[set][4 * j + 1][rr], ga[set][4 * j + 2][rr],
               ga[set][4 * j + 3][rr]);
      }
    }
#pragma unroll
    for (int j = 0; j < SB::PER; ++j) {
      char* dst = (SB::UNITS % 512 == 0 || bact[j]) ? buf + blds[j] : dump;
      if (B_KC) {
        put4(dst, B_PLANE, gb[set][j][0], gb[set][j][1], gb[set][j][2], gb[set][j][3]);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          put4(dst + rr * ROWB, B_PLANE, gb[set][4 * j][rr], gb[set][4 * j + 1][rr], gb[set][4 * j + 2][rr],
               gb[set][4 * j + 3][rr]);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int afrag = (wm * (TM * 32) + li) * ROWB + lh * 16;
  const int bfrag = 3 * A_PLANE + (wn * (TN * 32) + li) * ROWB + lh * 16;
  const std::integral_constant<int, 0> S0{};
  const std::integral_constant<int, 1> S1{};
  const std::true_type YES{};
  const std::false_type NO{};

  // One stage of parity P: the fragments of buffer P, then its MFMAs -- and, in the SAME basic block so that the
  // scheduler places them between the MFMAs (a wave issues in order: behind its MFMAs it would otherwise run ~100
  // vector instructions with the matrix pipe idle), the split + LDS stores of stage s + 1 (register set 1 - P ->
  // buffer 1 - P) and the global loads of stage s + 2 (-> register set P, free since the previous barrier)
  auto body = [&](int s, auto pc, auto more1, auto more2, auto tailc) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    const char* cur = bg_lds + P * BUF;
    bg_bf16x8 fa[BK / 16][TM][3], fb[BK / 16][TN][3];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[kk][i][pl] = *reinterpret_cast<const bg_bf16x8*>(cur + pl * A_PLANE + afrag + i * 32 * ROWB + kk * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[kk][j][pl] = *reinterpret_cast<const bg_bf16x8*>(cur + pl * B_PLANE + bfrag + j * 32 * ROWB + kk * 32);
      }
    if (decltype(more1)::value) stage(std::integral_constant<int, 1 - P>{}, bg_lds + (1 - P) * BUF);
    if (decltype(more2)::value) gload(pc, (s + 2) * BK, tailc);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];  // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i][1], fb[kk][j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i][0], fb[kk][j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i][2], fb[kk][j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i][0], fb[kk][j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i][1], fb[kk][j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i][0], fb[kk][j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    __syncthreads();
  };

  // ---- pipeline: LDS buffer and register set of stage s have parity s & 1; loads run two stages ahead
  const int nst = (p.K + BK - 1) / BK;
  gload(S0, 0, YES);
  stage(S0, bg_lds);
  if (nst > 1) gload(S1, BK, YES);
  __syncthreads();
  int s = 0;
  for (; s + 4 < nst; s += 2) {  // stages s + 2, s + 3 <= nst - 2: full stages
    body(s, S0, YES, YES, NO);
    body(s + 1, S1, YES, YES, NO);
  }
  for (; s < nst; s += 2) {
    if (s + 2 < nst) body(s, S0, YES, YES, YES);
    else if (s + 1 < nst) body(s, S0, YES, NO, NO);
    else body(s, S0, NO, NO, NO);
    if (s + 1 >= nst) break;
    if (s + 3 < nst) body(s + 1, S1, YES, YES, YES);
    else if (s + 2 < nst) body(s + 1, S1, YES, NO, NO);
    else body(s + 1, S1, NO, NO, NO);
  }

  // ---- epilogue.  C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + li;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[p.bias_div > 1 ? n / p.bias_div : n] : 0.f;   // bias_div < 0: one bias per output column
      const int mb = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m < p.M) {
          float v = acc[i][j][e] + bv;
          if (p.epi == REPO_EPI_ELU) v = elu(v);
          else if (p.epi == REPO_EPI_RELU) v = fmaxf(v, 0.f);
          else if (p.epi == REPO_EPI_MUL_DELU) v *= elu_grad_from_out(p.aux[(size_t)m * p.ldaux + n]);
          else if (p.epi == REPO_EPI_MUL_DRELU) v = p.aux[(size_t)m * p.ldaux + n] > 0.f ? v : 0.f;
          else if (p.epi == REPO_EPI_FILM_RELU) {   // row m's FiLM table: [scale (C) | shift (C)], C = ldaux / 2, channel n / bias_div
            const int ch = p.bias_div == 1 ? n : n / (p.bias_div < 0 ? -p.bias_div : p.bias_div);
            v = fmaxf(fmaf(p.aux[(size_t)m * p.ldaux + ch], v, p.aux[(size_t)m * p.ldaux + (p.ldaux >> 1) + ch]), 0.f);
          }
          float* c = p.C + (size_t)m * p.ldc + n;
          if (p.accumulate) v += *c;
          *c = v;
        }
      }
    }
}

typedef BgTile<256, 128, 16> BgBig;
typedef BgTile<128, 128, 32> BgMid;

// Shapes this engine takes (everything else stays on the fp32-MFMA tile engines): big products whose tiles fill the
// chip, 16-byte aligned operands with leading dimensions that keep every staged vector aligned.
#ifndef BG_MIN_TILES
#define BG_MIN_TILES 150   // (A/B builds: tools/build_variant.sh; 100 -- the 128 x 128 stack's fc at 1568 rows makes 104 tiles -- measured neutral: 12.94-12.99 against 12.90-13.02 ms per update, round 6)
#endif
inline bool bgemm_ok(int64_t M, int64_t N, int64_t K, bool a_kc, int64_t lda, bool b_kc, int64_t ldb, const void* A,
                     const void* B) {
  if (M < 512 || N < 512 || K < 128) return false;
  if (((M + 127) / 128) * ((N + 127) / 128) < BG_MIN_TILES) return false;
  if (lda % 4 || ldb % 4 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return false;
  if (!a_kc && M % 4) return false;  // row quads must not straddle the end of a row of A[k][m]
  if (!b_kc && N % 4) return false;
  return true;
}

template <class T, bool A_KC, bool B_KC>
static int launch_bgemm(const BgArgs& a, hipStream_t s) {
  const int gx = (a.N + T::BN - 1) / T::BN, gy = (a.M + T::BM - 1) / T::BM;
  hipError_t he = hipFuncSetAttribute((const void*)bgemm_kernel<T, A_KC, B_KC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      T::LDS_BYTES);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL((bgemm_kernel<T, A_KC, B_KC>), dim3((unsigned)(gx * gy)), dim3(512), T::LDS_BYTES, s, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

template <bool A_KC, bool B_KC>
static int bgemm_dispatch(const BgArgs& a, hipStream_t s) {
  // 256 x 128 tiles when they alone fill the chip twice over, else 128 x 128 (K = 32 per stage)
  const long big = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
  if (big >= 400) return launch_bgemm<BgBig, A_KC, B_KC>(a, s);
  return launch_bgemm<BgMid, A_KC, B_KC>(a, s);
}

}  // namespace repo
