// Weight gradients of the dense heads, dW[n][k] = sum_m dY[m][n] X[m][k] with ~200 x ~230 outputs and tens of
// thousands of rows m, straight from global memory to the matrix cores.
//
// Both operands are row-major over m, and v_mfma_f32_32x32x2_f32 wants A[i][kk] with lane = (i, kk) and B[kk][j]
// with lane = (j, kk): with i = n, j = k and kk = the row pair, a lane's operand is ONE element of row m of dY / X
// and the 32 lanes of a half-wave read 32 consecutive floats of that row -- no LDS staging, no transpose, no
// barrier.  The tile engine (vgemm.h) pads the 200 x 231 product to 256 x 256 on 64 x 64 tiles (30-39 % of its
// MFMAs are padding), re-reads both operands four times and runs at 0.33 of peak; here a workgroup of 8 waves owns
// the WHOLE output for a range of rows (wave w: the 32 k-columns 32w .. 32w+31 against all 7 n-tiles = 224 rows of
// the output, 112 accumulator registers), reads every operand element once, and the n-tiles are cut so that one
// 16-byte load of dY feeds four of them: tile t < 4 holds n = 4i + t (i = the MFMA row), tiles 4, 5 hold
// n = 128 + 2i + (t - 4) (one 8-byte load), tile 6 holds n = 192 + i.  The MFMA does not care which n a row stands
// for; the epilogue writes each accumulator row to its n.  Column k = K of the product is the bias gradient (B = 1).
// Split-K over row ranges into slabs [split][N][K+1], reduced in fixed order by slab_reduce_group_kernel: bit-
// reproducible like every other gradient of the update.
//
// Reference: autograd's weight gradients of the nn.Linear stacks in models/actor_critic.py:10-60 / models/decoder.py
// (RewardModel), dreamer.py:357-373.
#pragma once
#include "vgemm.h"

namespace repo {

constexpr int kWdMaxJobs = 8;
constexpr int kWdMaxSplits = 128;  // row ranges (workgroups) per job: 256 / jobs, so that one launch fills the 256 CUs once
struct WdJob {
  const float* dY;
  const float* X;
  float* slab;
  int rows, N, K, lddy, ldx, rps;
};
struct WdJobs {
  WdJob job[kWdMaxJobs];
  int njobs;
  int splits;   // wgrad_tr.h: row ranges per job (its grid is one-dimensional)
};

// what the kernel's n-tile cut and load widths assume
inline bool wgrad_direct_ok(int64_t rows, int64_t N, int64_t K, int64_t lddy, int64_t ldx) {
  return rows >= 4096 && N > 192 && N <= 224 && N % 4 == 0 && lddy % 4 == 0 && K >= 1 && K + 1 <= 256 &&
         rows * lddy < kMaxBufElems && rows * ldx < kMaxBufElems;
}

__global__ __launch_bounds__(512) void wgrad_direct_kernel(WdJobs g) {
  WdJob q = g.job[0];
#pragma unroll
  for (int i = 1; i < kWdMaxJobs; ++i)
    if (i == (int)blockIdx.y) q = g.job[i];  // constant indices only: the table stays in scalar registers
  const int z = blockIdx.x;
  const int rbeg = z * q.rps, rend = min(q.rows, rbeg + q.rps);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int N = q.N, K = q.K, K1 = q.K + 1;
  if (32 * w >= K1) return;  // K = 200: seven k-tiles, the eighth wave has none
  // rows >= rend are out of the descriptors' range: they load zeros (an odd row count's last pair, nothing else)
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(q.dY, 4u * (unsigned)((rend - 1) * q.lddy + N));
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(q.X, 4u * (unsigned)((rend - 1) * q.ldx + K));
  const unsigned yrow = 4u * (unsigned)((rbeg + lh) * q.lddy), xrow = 4u * (unsigned)((rbeg + lh) * q.ldx);
  const int kcol = 32 * w + li;
  unsigned o4 = yrow + 16u * (unsigned)li;                                                   // n = 4 li .. + 3 < 128 <= N
  unsigned o2 = (128 + 2 * li < N) ? yrow + 4u * (unsigned)(128 + 2 * li) : kOobOffset;      // n = 128 + 2 li, + 1
  unsigned o1 = (192 + li < N) ? yrow + 4u * (unsigned)(192 + li) : kOobOffset;              // n = 192 + li
  unsigned ob = (kcol < K) ? xrow + 4u * (unsigned)kcol : kOobOffset;
  const bool ones = kcol == K;
  const unsigned ystep = 8u * (unsigned)q.lddy, xstep = 8u * (unsigned)q.ldx;  // two rows per k-step

  f32x16 acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr int PD = 4;  // k-steps in flight
  f32x4 a4[PD];
  f32x2 a2[PD];
  float a1[PD], b1[PD];
  auto issue = [&](int slot) __attribute__((always_inline)) {
    a4[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, o4, 0, 0));
    a2[slot] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ry, o2, 0, 0));
    a1[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, o1, 0, 0));
    b1[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ob, 0, 0));
    o4 += ystep;
    o2 += (o2 >= kOobOffset) ? 0u : ystep;
    o1 += (o1 >= kOobOffset) ? 0u : ystep;
    ob += (ob >= kOobOffset) ? 0u : xstep;
  };
  const int nsteps = (rend - rbeg + 1) / 2;
#pragma unroll
  for (int s = 0; s < PD; ++s) issue(s);  // steps past the range read out-of-range rows: zeros
  for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const f32x4 va4 = a4[u];
      const f32x2 va2 = a2[u];
      const float va1 = a1[u];
      // the ones column: 1 for every row INSIDE the range (beyond it A is zero, so the product vanishes anyway)
      const float vb = ones ? 1.f : b1[u];
      issue(u);  // refill this slot with step s0 + u + PD
      if (s0 + u < nsteps) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(va4[t], vb, acc[t], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(va2[0], vb, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(va2[1], vb, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, vb, acc[6], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the window: no hoisting of later steps' loads
    }
  }
  // ---- slab[z][n][k], k on the lane (128 contiguous bytes per half-wave and accumulator row)
  if (kcol < K1) {
    float* sl = q.slab + (size_t)z * N * K1 + kcol;
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int n = t < 4 ? 4 * i + t : (t < 6 ? 128 + 2 * i + (t - 4) : 192 + i);
        if (n < N) sl[(size_t)n * K1] = acc[t][r];
      }
  }
}

}  // namespace repo
