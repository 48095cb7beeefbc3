// Conv weight gradient at fp32 accuracy on the bf16 matrix pipe ("bf16x6", bgemm.h): dconv_wgrad_kernel (dconv.h) with
// its MFMA loop on v_mfma_f32_32x32x16_bf16.
//
//   dw[cs][cb][ky][kx] = sum_{img,sy,sx} small[img][cs][sy][sx] * big[img][cb][2sy+ky][2sx+kx]
//   M = cs (A = small), N = (cb,ky,kx) (B = big), K = (img, sy, sx), split over image groups into slabs (unchanged)
//
// The chunking, staging of `big`, split-K slabs and the epilogue are dconv_wgrad_kernel's.  What changes:
//   * K is still the FLAT pixel index kf = sy*WS + sx of a band, now walked 16 at a time: lane half h takes
//     kf = 16b + 8h .. + 7 (one MFMA k-block), so an A fragment is 8 CONSECUTIVE floats of a row of the band copy;
//   * A (`small`, contiguous along k) is split while it is stored: three bf16 planes with rows of AP elements (AP = 8 mod 16:
//     conflict-free ds_read_b128), a fragment = one aligned 16-byte read per plane, no vector arithmetic at use;
//   * B (`big`) keeps its fp32 LDS image (its fragment is a stride-2 gather with row wraps: eight ds_read_b32 with
//     immediate offsets -- pixel kf sits at 2*sy*BRP + 2*sx, and the half h = 1 lanes are 16 + w*(2*BRP - 2*WS) floats on,
//     w = row wraps between kf and kf + 8, a compile-time property of the slot: two per-lane bases) and is split IN
//     REGISTERS at use: 44 vector instructions per 8 elements, shared by the wave's TM m-tiles -- with 64 x 64 wave
//     tiles that is 88 vector instructions beside 24 MFMAs of 32 cycles;
//   * phantom k (the last block of a band: kf >= ROWS*WS) are forced to zero on the B side (select, never multiply:
//     the reads land in stale or unwritten LDS).
#pragma once
#include "bgemm.h"
#include "dconv.h"

namespace repo {

constexpr int bw_pitch(int len) {  // >= len, == 8 (mod 16) bf16 elements
  int p = (len + 15) & ~15;
  return p + 8;
}

template <class G, class T>
__global__ __launch_bounds__(T::NT) void bconv_wgrad_kernel(WgradArgs p) {
  constexpr int BM = T::BM, BN = T::BN, TM = T::TM, TN = T::TN, NT = T::NT, GI = T::GI, RB = T::RB;
  constexpr int NW = G::CB * G::KK;
  constexpr int NB = (G::HS + RB - 1) / RB;
  constexpr int BR = 2 * RB + G::KS - 2;
  constexpr int BRP = (G::WB % 64 == 0) ? G::WB + 4 : G::WB;
  constexpr int ALEN = RB * G::WS, BLEN = BR * BRP;
  constexpr int NBLK_MAX = (ALEN + 15) / 16;
  constexpr int AP = bw_pitch(16 * NBLK_MAX);              // bf16 elements per (image, cs) row
  constexpr int APLANE = GI * BM * AP * 2;                 // bytes of one A plane
  constexpr int BP = pitch4(BLEN);
  constexpr int CBT = cmin(G::CB, (BN + G::KK - 2) / G::KK + 1);
  constexpr int A_VPC = (ALEN + 3) / 4, B_VPC = (BR * G::WB + 3) / 4;
  constexpr int A_NV = GI * BM * A_VPC, B_NV = GI * CBT * B_VPC;
  constexpr int A_PER = (A_NV + NT - 1) / NT, B_PER = (B_NV + NT - 1) / NT;
  constexpr int WSTEP = 2 * BRP - 2 * G::WS;               // extra B offset per row wrap
  constexpr int WMIN = 8 / G::WS;                          // row wraps between kf and kf + 8: WMIN or WMIN + 1
  constexpr int LDS_BYTES = 3 * APLANE + (GI * CBT * BP + 64) * 4;
  __shared__ __attribute__((aligned(16))) char lds[cmax(LDS_BYTES, NT * 4)];
  char* Al = lds;
  float* Bl = reinterpret_cast<float*>(lds + 3 * APLANE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / T::WN, wn = wid % T::WN;
  const int li = lane & 31, lh = lane >> 5;

  int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
  {
    const int tpz = gridDim.x * gridDim.y, gz = gridDim.z;
    const int L = bx + gridDim.x * (by + gridDim.y * z);
    const int full = (gz / 8) * 8 * tpz;
    if (L < full) {
      const int grp = L / (8 * tpz), r = L % (8 * tpz);
      const int t = r / 8;
      z = grp * 8 + r % 8;
      bx = t % (int)gridDim.x;
      by = t / (int)gridDim.x;
    }
  }
  const int n0 = bx * BN, m0 = by * BM;
  const int img_beg = z * p.imgs_per_split, img_end = min(p.nimg, img_beg + p.imgs_per_split);
  const int cbf = n0 / G::KK;
  const int ngrp = (img_end - img_beg + GI - 1) / GI;
  const int nch = ngrp * NB;

  const __amdgpu_buffer_rsrc_t rsm = make_rsrc(p.small, p.small_bytes), rbg = make_rsrc(p.big, p.big_bytes);

  int aoff[A_PER], boff[B_PER], ag[A_PER], bgi[B_PER];
  int alds[A_PER], blds[B_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int v = tid + j * NT, e4 = v % A_VPC, rm = v / A_VPC;
    const int m = rm % BM, g = rm / BM;
    const bool act = g < GI && m0 + m < G::CS;
    ag[j] = act ? g : 1 << 20;
    aoff[j] = (g * G::CS + m0 + m) * G::PS + e4 * 4;
    alds[j] = ((g * BM + m) * AP + e4 * 4) * 2;   // bytes inside a plane
  }
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int v = tid + j * NT, e4 = v % B_VPC, rc = v / B_VPC;
    const int c = rc % CBT, g = rc / CBT;
    const bool act = g < GI && cbf + c < G::CB;
    bgi[j] = act ? g : 1 << 20;
    boff[j] = (g * G::CB + cbf + c) * G::PB + e4 * 4;
    blds[j] = (g * CBT + c) * BP + (BRP == G::WB ? e4 * 4 : (e4 * 4 / G::WB) * BRP + e4 * 4 % G::WB);
  }

  // ---- per-lane fragment bases.  A: byte offset of (row m, k = 8 lh) inside a plane; B: two bases per n-tile (float
  // offsets) for the two possible numbers of row wraps between a slot's lower and upper half
  int abase[TM], bb[TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i) abase[i] = (((wm * TM + i) * 32 + li) * AP + 8 * lh) * 2;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = min(n0 + (wn * TN + j) * 32 + li, NW - 1);
    const int cb = n / G::KK, r = n % G::KK;
    const int b0 = (cb - cbf) * BP + (r / G::KS) * BRP + r % G::KS;
    bb[j][0] = b0 + lh * (16 + WMIN * WSTEP);
    bb[j][1] = b0 + lh * (16 + (WMIN + 1) * WSTEP);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float dbsum = 0.f;

  f32x4 rav[A_PER], rbv[B_PER];
  int ld_img0 = 0, ld_abias = 0, ld_bbias = 0;
  auto gload = [&](int c) __attribute__((always_inline)) {
    const int grp = c / NB, band = c % NB;
    const int r0 = band * RB;
    ld_img0 = img_beg + grp * GI;
    ld_abias = ld_img0 * G::CS * G::PS + r0 * G::WS;
    ld_bbias = ld_img0 * G::CB * G::PB + 2 * r0 * G::WB;
#pragma unroll
    for (int j = 0; j < A_PER; ++j)
      rav[j] = VecLoad<4>::load(rsm, ld_img0 + ag[j] < img_end ? 4u * (unsigned)(aoff[j] + ld_abias) : kOobOffset);
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      rbv[j] = VecLoad<4>::load(rbg, ld_img0 + bgi[j] < img_end ? 4u * (unsigned)(boff[j] + ld_bbias) : kOobOffset);
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j)
      if ((A_NV % NT == 0) || tid + j * NT < A_NV) {
        unsigned a1, a2, a3, b1, b2, b3;
        bg_split3(rav[j][0], rav[j][1], a1, a2, a3);
        bg_split3(rav[j][2], rav[j][3], b1, b2, b3);
        *reinterpret_cast<bg_u32x2*>(Al + alds[j]) = bg_u32x2{a1, b1};
        *reinterpret_cast<bg_u32x2*>(Al + APLANE + alds[j]) = bg_u32x2{a2, b2};
        *reinterpret_cast<bg_u32x2*>(Al + 2 * APLANE + alds[j]) = bg_u32x2{a3, b3};
      }
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      if ((B_NV % NT == 0) || tid + j * NT < B_NV) *reinterpret_cast<f32x4*>(Bl + blds[j]) = rbv[j];
  };

  constexpr int RL = G::HS - (NB - 1) * RB;
  auto compute = [&](auto rows_tag, int c) __attribute__((always_inline)) {
    constexpr int ROWS = decltype(rows_tag)::value;
    constexpr int NK = ROWS * G::WS, NBLK = (NK + 15) / 16;
    const int band = c % NB;
    const int R = min(RB, G::HS - band * RB);
#pragma unroll
    for (int g = 0; g < GI; ++g) {
      const char* ar = Al + g * BM * AP * 2;
      const float* br = Bl + g * CBT * BP;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        bg_bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            fa[i][q] = *reinterpret_cast<const bg_bf16x8*>(ar + q * APLANE + abase[i] + 32 * b);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int klo = 16 * b + e, khi = klo + 8;
            const int off = 2 * (klo / G::WS) * BRP + 2 * (klo % G::WS);
            const int w = khi / G::WS - klo / G::WS - WMIN;   // 0 or 1 (compile time)
            const bool vlo = klo < NK, vhi = khi < NK;
            float x = 0.f;
            if (vlo) x = br[bb[j][w] + off];
            if (vlo && !vhi) x = lh ? 0.f : x;   // the upper half of this slot is a phantom k
            v[e] = x;
          }
          unsigned pl[3][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) bg_split3(v[2 * e], v[2 * e + 1], pl[0][e], pl[1][e], pl[2][e]);
#pragma unroll
          for (int q = 0; q < 3; ++q) fb[j][q] = __builtin_bit_cast(bg_bf16x8, u32x4s{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            f32x16 cacc = acc[i][j];  // smallest terms first
            cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], cacc, 0, 0, 0);
            cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], cacc, 0, 0, 0);
            cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], cacc, 0, 0, 0);
            cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], cacc, 0, 0, 0);
            cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], cacc, 0, 0, 0);
            cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], cacc, 0, 0, 0);
            acc[i][j] = cacc;
          }
        // one block's fragments at a time: without the fence the scheduler hoists the later blocks' LDS reads and
        // splits up here and the kernel drops to one wave per SIMD
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (p.want_db && bx == 0) {  // bias gradient of `small`: row sums of the A chunk (a = a1 + a2 + a3 exactly)
      constexpr int PARTS = NT / BM;
      const int m = tid % BM, part = tid / BM;
      if (part < PARTS) {
        for (int g = 0; g < GI; ++g)
          for (int e = part; e < R * G::WS; e += PARTS) {
            const int o = ((g * BM + m) * AP + e) * 2;
            float s = 0.f;
#pragma unroll
            for (int q = 2; q >= 0; --q)
              s += __builtin_bit_cast(float, (unsigned)(*reinterpret_cast<const unsigned short*>(Al + q * APLANE + o)) << 16);
            dbsum += s;
          }
      }
    }
  };

  if (nch > 0) {
    gload(0);
    // rows of the A planes beyond a band's pixels are read by the last k-block: finite values are all they must hold
    for (int i = tid; i < 3 * APLANE / 16; i += NT) reinterpret_cast<f32x4*>(Al)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      lstore();
      __syncthreads();
      gload(min(c + 1, nch - 1));
      if (RL == RB || c % NB != NB - 1) compute(std::integral_constant<int, RB>{}, c);
      else compute(std::integral_constant<int, RL>{}, c);
      __syncthreads();
    }
  }

  // ---- epilogue: slab[z][m][n]
  constexpr int LDS_ = NW + 1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + li;
      const int mb = m0 + (wm * TM + i) * 32 + 4 * lh;
      if (n < NW && mb < G::CS) {
        float* cdst = p.slab + ((size_t)z * G::CS + mb) * LDS_ + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dm = (r & 3) + 8 * (r >> 2);
          if (mb + dm < G::CS) cdst[dm * LDS_] = acc[i][j][r];
        }
      }
    }
  if (p.want_db && bx == 0) {
    constexpr int PARTS = NT / BM;
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();
    red[tid] = dbsum;
    __syncthreads();
    if (tid < BM && m0 + tid < G::CS) {
      float s = 0.f;
      for (int q = 0; q < PARTS; ++q) s += red[q * BM + tid];
      p.slab[((size_t)z * G::CS + m0 + tid) * LDS_ + NW] = s;
    }
  }
}

template <class G, class T>
inline int launch_bconv_wgrad(const WgradArgs& a, int splits, hipStream_t s) {
  constexpr int NW = G::CB * G::KK;
  dim3 grid((NW + T::BN - 1) / T::BN, (G::CS + T::BM - 1) / T::BM, (unsigned)splits);
  hipLaunchKernelGGL((bconv_wgrad_kernel<G, T>), grid, dim3(T::NT), 0, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
