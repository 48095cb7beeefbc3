// Weight gradients of the dense heads on the bf16 matrix pipe ("bf16x6", bgemm.h):
//   dW[n][k] = sum_m dY[m][n] X[m][k],  ~200 x ~231 outputs (column K = the bias gradient: X's ones column), tens of
//   thousands of rows m -- the K dimension of the product is the ROW index of both operands.
//
// wgrad_direct.h feeds v_mfma_f32_32x32x2_f32 straight from global memory (the fp32 MFMA takes a row-major operand element
// by element) and is bound by that instruction: 0.63 of the fp32 pipe, 150 us for the actor trunk's four layers.  The
// bf16 MFMA wants 8 consecutive m per lane -- a transposition of both operands.  gfx950's ds_read_b64_tr_b16 does it on
// the way out of LDS (twgrad.h): both operands are staged AS THEY LIE, rows of bf16 in three planes (the exact split of
// bgemm.h), and a fragment of either is two transposing reads -- lane 4q+p of a 16-lane group names row q's address,
// columns 4p .. 4p+3, and receives column (lane & 15) of the four rows.  The row pitch is an odd number of 32 B: the 8
// rows a half-wave reads (the 32 m of a block are dealt to the lane groups so that they are consecutive) hit every bank
// once; every address is (lane base) + immediate.
//   * workgroup = (job, row range, half of X's column tiles: 8 + 7 of 15, 7 + 6 of 13): 13 n-tiles x its k-tiles of 16 x 16;
//     both halves stage dY;
//   * 8 waves, SPECIALISED as in twgrad.h: waves 0-3 multiply (2 k-tiles each, B fragments held for the block, A fragments
//     streamed tile by tile one step ahead: 12 MFMAs per 6 reads), waves 4-7 stage (global -> split -> the other of two
//     LDS buffers, loads two blocks ahead); one LDS-only barrier per 32 rows;
//   * slab[z][n][K+1] as wgrad_direct.h's (slab_reduce_group_kernel adds the row ranges in fixed order).
// Reference: autograd's weight gradients of the nn.Linear stacks in models/actor_critic.py:10-60 / models/decoder.py
// (RewardModel), dreamer.py:357-373.
#pragma once
#include "twgrad.h"
#include "wgrad_direct.h"

namespace repo {

constexpr int kWtNT = 13, kWtKT = 8;               // n-tiles; k-tiles per column half
constexpr int kWtPY = 32 * kWtNT;                  // 416 B: 208 bf16, 13 x 32 B
constexpr int kWtPX = 32 * (kWtKT + 1);            // 288 B: 128 bf16 + 32 B pad, 9 x 32 B
constexpr int kWtYPlane = 32 * kWtPY, kWtXPlane = 32 * kWtPX;
constexpr int kWtBuf = 3 * (kWtYPlane + kWtXPlane);
constexpr int kWtLds = 2 * kWtBuf;
constexpr int kWtYQ = kWtPY / 8, kWtXQ = 16 * kWtKT / 4;   // column quads per row: 52, 32
constexpr int kWtItems = 32 * (kWtYQ + kWtXQ);             // 2688 per block
constexpr int kWtPer = (kWtItems + 255) / 256;             // 11 per staging thread

inline bool wgrad_tr_ok(int64_t rows, int64_t N, int64_t K, int64_t lddy, int64_t ldx) {
  return rows >= 4096 && N > 16 * (kWtNT - 1) && N <= 16 * kWtNT && N % 4 == 0 && lddy % 4 == 0 && K >= 128 &&
         K + 1 <= 16 * (2 * kWtKT - 1) && rows * lddy < kMaxBufElems && rows * ldx < kMaxBufElems;
}

__global__ __launch_bounds__(512) void wgrad_tr_kernel(WdJobs g) {
  // the two column halves of a (job, row range) read the same dY: consecutive workgroups go to consecutive XCDs (8 L2s), so
  // the pair is dealt 8 apart -- same XCD, same L2 (430 -> ~260 MB of HBM reads per launch, which was the bound: 4.9 TB/s)
  const int L = blockIdx.x;
  const int half = (L >> 3) & 1, idx = (L >> 4) * 8 + (L & 7);
  const int jobi = idx / g.splits, z = idx - jobi * g.splits;
  if (jobi >= g.njobs) return;
  WdJob q = g.job[0];
#pragma unroll
  for (int i = 1; i < kWdMaxJobs; ++i)
    if (i == jobi) q = g.job[i];  // constant indices only: the table stays in scalar registers
  extern __shared__ __attribute__((aligned(16))) char wt_lds[];
  const int tid = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rbeg = z * q.rps, rend = min(q.rows, rbeg + q.rps);
  const int nblk = (rend - rbeg + 31) / 32;
  const int N = q.N, K = q.K, K1 = q.K + 1;
  const int nkt_all = (K1 + 15) / 16, nkt_lo = (nkt_all + 1) / 2;   // 13 -> 7 + 6, 15 -> 8 + 7 k-tiles
  const int kt0 = half * nkt_lo;                      // first k-tile of this half
  const int nkt = half ? nkt_all - nkt_lo : nkt_lo;   // its k-tiles

  if (wid < 4) {
    // =================================================================== the multiplying waves
    __builtin_amdgcn_s_setprio(2);
    const int lane = tid & 63, lg = lane >> 4;
    // lane 4q+p of group lg: row 16 (lg >> 1) + 4 (lg & 1) + q (+ 8 for the second read), columns 4p .. 4p+3 of a tile
    const int rowb = 16 * (lg >> 1) + 4 * (lg & 1) + ((lane & 15) >> 2);
    const int ybase = rowb * kWtPY + 8 * (lane & 3), xbase = rowb * kWtPX + 8 * (lane & 3) + 64 * wid;   // wave: k-tiles 2 wid, 2 wid + 1
    f32x4 acc[kWtNT][2];
#pragma unroll
    for (int i = 0; i < kWtNT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the last wave of the upper half may own no tile; with an odd tile count the last live wave's second tile is the other
    // half's first (or a column range past K): it is multiplied all the same -- no branch in the loop -- and stored only
    // where k <= K, where both halves write the same bits
    const bool live = 2 * wid < nkt;

    auto frag = [&](const char* base, int pitch8, int off) __attribute__((always_inline)) {
      const tw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(TW_LDS(base + off));
      const tw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(TW_LDS(base + off + pitch8));
      return __builtin_bit_cast(bg_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    lds_barrier();   // block 0 is staged
    for (int b = 0; b < nblk; ++b) {
      const char* Yl = wt_lds + (b & 1) * kWtBuf;
      const char* Xl = Yl + 3 * kWtYPlane;
      if (live) {
        bg_bf16x8 fb[2][3], fa[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fb[j][pl] = frag(Xl + pl * kWtXPlane, 8 * kWtPX, xbase + 32 * j);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[0][pl] = frag(Yl + pl * kWtYPlane, 8 * kWtPY, ybase);
#pragma unroll
        for (int i = 0; i < kWtNT; ++i) {
          if (i + 1 < kWtNT) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[(i + 1) & 1][pl] = frag(Yl + pl * kWtYPlane, 8 * kWtPY, ybase + 32 * (i + 1));
          }
          __builtin_amdgcn_sched_barrier(0);
          constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};   // smallest terms first
#pragma unroll
          for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][PA[pr]], fb[j][PB[pr]], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      lds_barrier();
    }
    // ---- slab[z][n][k]: lane & 15 = k inside the tile, accumulator register r = row n = 16 i + 4 lg + r
    if (live) {
      float* sl = q.slab + (size_t)z * N * K1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = 16 * (kt0 + 2 * wid + j) + (lane & 15);
        if (k < K1) {
#pragma unroll
          for (int i = 0; i < kWtNT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int n = 16 * i + 4 * lg + r;
              if (n < N) sl[(size_t)n * K1 + k] = acc[i][j][r];
            }
        }
      }
    }
  } else {
    // =================================================================== the staging waves
    // dY items: v = ptid + 256 i -> (row v / 52, column quad v % 52), 7 per thread; X items: v -> (row v / 32, quad v % 32
    // of this half), 4 per thread.  Each kind has its own loop: the descriptor of a load must be wave-uniform
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(q.dY, 4u * (unsigned)((q.rows - 1) * q.lddy + N));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(q.X, 4u * (unsigned)((q.rows - 1) * q.ldx + K));
    constexpr int YPER = (32 * kWtYQ + 255) / 256, XPER = 32 * kWtXQ / 256;
    static_assert(YPER + XPER == kWtPer && 32 * kWtXQ % 256 == 0, "staging items");
    const int ptid = tid - 256;
    f32x4 rvy[2][YPER], rvx[2][XPER];
    // a dead item's offset gets the top bit (out of range: zeros) by arithmetic: a select becomes branches around the loads
    auto deadbit = [](int ok_minus_1_minus_x) { return (unsigned)(ok_minus_1_minus_x >> 31) & kOobOffset; };
    auto gload = [&](auto sc, int b) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value;
      const int r0 = rbeg + 32 * b;
      const unsigned dead_blk = b < nblk ? 0u : kOobOffset;
#pragma unroll
      for (int i = 0; i < YPER; ++i) {
        const int v = ptid + 256 * i, m = v / kWtYQ, col = 4 * (v - m * kWtYQ);
        const unsigned dead = deadbit(rend - 1 - (r0 + m)) | deadbit(N - 1 - col) | deadbit(32 * kWtYQ - 1 - v) | dead_blk;
        rvy[S][i] = VecLoad<4>::load(ry, (4u * (unsigned)((r0 + m) * q.lddy + col)) | dead);
      }
#pragma unroll
      for (int i = 0; i < XPER; ++i) {
        const int v = ptid + 256 * i, m = v / kWtXQ, col = 16 * kt0 + 4 * (v - m * kWtXQ);
        // a quad that straddles the end of X's row is loaded from K - 4 and rotated by the store (never past the buffer)
        const int cl = min(col, K - 4);
        const unsigned dead = deadbit(rend - 1 - (r0 + m)) | deadbit(K - 1 - col) | dead_blk;
        rvx[S][i] = VecLoad<4>::load(rx, (4u * (unsigned)((r0 + m) * q.ldx + cl)) | dead);
      }
    };
    auto put = [&](char* dst, int plane, float x0, float x1, float x2, float x3) __attribute__((always_inline)) {
      unsigned a1, a2, a3, b1, b2, b3;
      tw_split3(x0, x1, a1, a2, a3);
      tw_split3(x2, x3, b1, b2, b3);
      *reinterpret_cast<bg_u32x2*>(dst) = bg_u32x2{a1, b1};
      *reinterpret_cast<bg_u32x2*>(dst + plane) = bg_u32x2{a2, b2};
      *reinterpret_cast<bg_u32x2*>(dst + 2 * plane) = bg_u32x2{a3, b3};
    };
    auto lstore = [&](auto sc, int b) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value;
      char* Yl = wt_lds + (b & 1) * kWtBuf;
      char* Xl = Yl + 3 * kWtYPlane;
      const int r0 = rbeg + 32 * b;
#pragma unroll
      for (int i = 0; i < YPER; ++i) {
        const int v = ptid + 256 * i, m = v / kWtYQ, cq = v - m * kWtYQ;
        if (32 * kWtYQ % 256 == 0 || v < 32 * kWtYQ)
          put(Yl + m * kWtPY + 8 * cq, kWtYPlane, rvy[S][i][0], rvy[S][i][1], rvy[S][i][2], rvy[S][i][3]);
      }
#pragma unroll
      for (int i = 0; i < XPER; ++i) {
        const int v = ptid + 256 * i, m = v / kWtXQ, cq = v - m * kWtXQ, col = 16 * kt0 + 4 * cq;
        const int sh = col - min(col, K - 4);   // > 0: the quad was loaded from K - 4, element e sits at e + sh
        const float one = r0 + m < rend ? 1.f : 0.f;   // the ones column (the bias gradient), inside the row range only
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int sidx = e + sh;
          float t = rvx[S][i][e];
          t = sidx == 1 ? rvx[S][i][1] : t;
          t = sidx == 2 ? rvx[S][i][2] : t;
          t = sidx == 3 ? rvx[S][i][3] : t;
          x[e] = col + e < K ? t : (col + e == K ? one : 0.f);
        }
        put(Xl + m * kWtPX + 8 * cq, kWtXPlane, x[0], x[1], x[2], x[3]);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // block b is multiplied from buffer b & 1 while block b + 1 (register set (b + 1) & 1) is split into the other buffer
    // and block b + 3's loads are issued into the set that just emptied
    gload(I0{}, 0);
    gload(I1{}, 1);
    lstore(I0{}, 0);
    gload(I0{}, 2);
    lds_barrier();
    for (int b = 0; b < nblk; b += 2) {
      if (b + 1 < nblk) lstore(I1{}, b + 1);
      gload(I1{}, b + 3);
      lds_barrier();
      if (b + 1 < nblk) {
        if (b + 2 < nblk) lstore(I0{}, b + 2);
        gload(I0{}, b + 4);
        lds_barrier();
      }
    }
  }
}

inline int launch_wgrad_tr(const WdJobs& dj, int splits, hipStream_t s) {
  hipError_t e = hipFuncSetAttribute((const void*)wgrad_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWtLds);
  if (e != hipSuccess) return (int)e;
  WdJobs t = dj;
  t.splits = splits;
  hipLaunchKernelGGL(wgrad_tr_kernel, dim3(16u * (unsigned)((splits * dj.njobs + 7) / 8)), dim3(512), kWtLds, s, t);
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
