// Stride-2 "down" convolutions at fp32 accuracy on the bf16 matrix pipe ("bf16x6", see bgemm.h for the arithmetic).
//
//   small[img][cs][sy][sx] = sum_{cb,ky,kx} big[img][cb][2sy+ky][2sx+kx] w[cs][cb][ky][kx]
//   M = CS (weights: A operand), N = pixels (img, sy, sx), K = (cb, ky, kx)
// The frame of dconv.h's direct kernel is kept -- the workgroup copies the RAW input rows its output pixels need into
// LDS once per channel chunk, with 16-byte loads of contiguous memory, and reads MFMA fragments straight from that
// patch at  base(pixel of this lane) + offset(k), the offset a compile-time immediate of the unrolled K loop; the same
// epilogue (dconv_down_epilogue) -- what changes is the number format of the two LDS images:
//   * the patch is split while it is stored: three bf16 planes (6 bytes per element instead of 4).  A v_mfma_f32_32x32x16_bf16
//     B fragment is 8 consecutive k of one pixel; k is ordered so that they are whole ROWS of taps of one channel: a
//     lane half h takes channel 2g + h of a channel pair, its 8 k are slots q = 8b .. 8b+7 of that channel's
//     (ky, kx) taps in row-major order, rows padded to an even length (k5: one slot whose weight is zero) and the
//     channel to a multiple of 8 slots (k6: 36 -> 40, k5: 30 -> 32, k4: 16): two taps (kx even, kx + 1) are one aligned
//     dword of the bf16 plane (the big rows' pitch must be even), so a fragment is 4 ds_read_b32 per plane with
//     immediate offsets and NO vector arithmetic -- the split was paid once per staged element, not per use;
//   * the weights arrive pre-split and fragment-ready from a pack (bconv_pack_kernel, one launch per call: they
//     change once per optimiser step): per (M tile, channel chunk) [plane][block][half][m][8 bf16], so staging is a
//     linear 16-byte copy and an A fragment one conflict-free ds_read_b128 per plane.
// Six MFMAs per (tile, 16 k); ONE LDS buffer, two barriers per channel chunk (the next chunk's global loads are in
// flight during the MFMAs; 2-3 workgroups per CU cover the barriers).
#pragma once
#include "bgemm.h"
#include "dconv.h"

namespace repo {

template <class G>
struct BGeo {
  static constexpr int KSE = G::KS + (G::KS & 1);   // slots per tap row (even)
  static constexpr int TAPS = G::KS * KSE;          // slots per channel before padding
  static constexpr int KKE = (TAPS + 7) & ~7;       // padded to whole 8-k fragments
  static constexpr int BPG = KKE / 8;               // MFMA k-blocks (16 k = 8 per lane half) per channel PAIR
  // dword-aligned tap pairs need an even row pitch: an odd-width plane (31 x 31, 13 x 13) is stored in LDS with its
  // rows padded to a multiple of FOUR elements (WBE: 32, 16) and staged by LDS quads (QROW in the kernel)
  static constexpr bool ODD = (G::WB & 1) != 0;
  static constexpr int WBE = ODD ? ((G::WB + 3) & ~3) : G::WB;
  // an odd channel count (the encoder's 3-channel first layer) is padded to a channel PAIR: the phantom channel's patch
  // is staged as zeros (out-of-range loads) against zero weights in the pack
  static constexpr int CBP = G::CB + (G::CB & 1);
};

// BM x BN tile, CK channels per chunk (even), WM x WN waves
template <int BM_, int BN_, int CK_, int WM_, int WN_>
struct BTile {
  static constexpr int BM = BM_, BN = BN_, CK = CK_, WM = WM_, WN = WN_;
  static constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  static constexpr int NT = WM * WN * 64;
  static_assert(TM >= 1 && TN >= 1 && BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && CK % 2 == 0, "tile / wave grid mismatch");
};

template <class G, class T>
struct BPack {
  static constexpr int NBLK = (T::CK / 2) * BGeo<G>::BPG;            // k-blocks per chunk
  static constexpr int NSL = BGeo<G>::CBP / T::CK;                    // chunks
  static constexpr int MT = (G::CS + T::BM - 1) / T::BM;              // M tiles
  static constexpr int WPLANE = NBLK * 2 * T::BM * 16;                // bytes of one plane of one chunk
  static constexpr int CHUNK_BYTES = 3 * WPLANE;
  static constexpr size_t BYTES = (size_t)MT * NSL * CHUNK_BYTES;
  // uint8 frames (below): the folded bias vector sits behind the weight planes
  static constexpr size_t BYTES_U8 = BYTES + (size_t)((G::CS * 4 + 255) & ~255);
  static_assert(BGeo<G>::CBP % T::CK == 0, "channel chunk must divide the (pair-padded) channel count");
};

struct BPackArgs {
  const float* w;   // [CS][CB][KS][KS]
  char* wp;
  const float* bias;  // U8 only (nullable)
};

// one thread per (M tile, chunk, block, half, m): the 8 k of one A fragment, all three planes.
// U8 (the frames of the encoder's first layer stay uint8 in HBM; reference: x = (v / 255) * 2 - 1, common/utils.py:79):
//   sum_k w_k x_k + b  =  sum_k (2 w_k / 255) v_k  +  (b - sum_k w_k)
// -- the kernel multiplies the RAW bytes v (exact in ONE bf16: no split of the activation, three products instead of six)
// with weights scaled here, and its bias is the folded vector written behind the planes.
template <class G, class T, bool U8 = false>
__global__ __launch_bounds__(256) void bconv_pack_kernel(BPackArgs p) {
  typedef BGeo<G> BG;
  typedef BPack<G, T> P;
  const int total = P::MT * P::NSL * P::NBLK * 2 * T::BM;
  if (U8) {
    float* fb = reinterpret_cast<float*>(p.wp + P::BYTES);
    for (int cs = blockIdx.x * 256 + threadIdx.x; cs < G::CS; cs += gridDim.x * 256) {
      double s = 0.0;
      for (int k = 0; k < G::CB * G::KK; ++k) s += (double)p.w[(size_t)cs * G::CB * G::KK + k];
      fb[cs] = (float)((double)(p.bias ? p.bias[cs] : 0.f) - s);
    }
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int r = i;
    const int m = r % T::BM;
    r /= T::BM;
    const int h = r & 1;
    r >>= 1;
    const int blk = r % P::NBLK;
    r /= P::NBLK;
    const int t = r % P::NSL, mt = r / P::NSL;
    const int g = blk / BG::BPG, b = blk % BG::BPG;
    const int cs = mt * T::BM + m, cb = t * T::CK + 2 * g + h;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = 8 * b + j, ky = q / BG::KSE, kx = q % BG::KSE;
      v[j] = (q < BG::TAPS && kx < G::KS && cs < G::CS && cb < G::CB) ? p.w[((size_t)cs * G::CB + cb) * G::KK + ky * G::KS + kx] : 0.f;
      if (U8) v[j] *= 2.f / 255.f;
    }
    unsigned pl[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bg_split3(v[2 * e], v[2 * e + 1], pl[0][e], pl[1][e], pl[2][e]);
    char* dst = p.wp + ((size_t)mt * P::NSL + t) * P::CHUNK_BYTES + ((blk * 2 + h) * T::BM + m) * 16;
#pragma unroll
    for (int q3 = 0; q3 < 3; ++q3)
      *reinterpret_cast<u32x4s*>(dst + q3 * P::WPLANE) = u32x4s{pl[q3][0], pl[q3][1], pl[q3][2], pl[q3][3]};
  }
}

template <class G, class T, class BigT = float>
__global__ __launch_bounds__(T::NT) void bconv_down_kernel(DownArgs p) {
  typedef BGeo<G> BG;
  typedef BPack<G, T> P;
  constexpr bool U8 = std::is_same<BigT, uint8_t>::value;   // raw bytes: ONE patch plane (a byte is exact in bf16)
  constexpr int NPL = U8 ? 1 : 3;
  static_assert(BG::CBP == G::CB || P::NSL == 1, "a padded channel count: one chunk (the phantom channel is chunk-local)");
  static_assert(!U8 || !BG::ODD, "uint8 frames: even row pitch");
  constexpr bool ODD = BG::ODD;
  constexpr int WBE = BG::WBE;
  constexpr int BM = T::BM, BN = T::BN, CK = T::CK, NT = T::NT, TM = T::TM, TN = T::TN;
  constexpr int NBLK = P::NBLK, NSL = P::NSL, WPLANE = P::WPLANE;
  // ---- patch geometry (dconv_down_kernel): a tile is BN consecutive pixels of (img, sy, sx); image i of the tile
  // needs input rows [2*f_i, 2*l_i + KS) -- full width, hence ONE contiguous span per (image, channel)
  constexpr int ROWS_FULL = 2 * (G::HS - 1) + G::KS;
  constexpr int LENFULL = (ROWS_FULL * G::WB + 3) & ~3;
  constexpr int NIMG_MAX = (BN - 1) / G::PS + 2;
  constexpr int PLMAX = cmin(NIMG_MAX * LENFULL,
                             ((2 * (BN / G::WS + 2) + NIMG_MAX * (G::KS - 2)) * G::WB + 4 * NIMG_MAX + 3) & ~3);
  constexpr int PLV = PLMAX / 4;
  // the LDS image of a span (pitch WBE): its own lengths where the pitch is padded
  constexpr int LENFULL_E = ODD ? ROWS_FULL * WBE : LENFULL;
  constexpr int PLMAX_E = ODD ? cmin(NIMG_MAX * LENFULL_E, ((2 * (BN / G::WS + 2) + NIMG_MAX * (G::KS - 2)) * WBE + 3) & ~3)
                              : PLMAX;
  constexpr int PPLANE = (CK * PLMAX_E * 2 + 15) & ~15;      // bytes of one patch plane
  // QROW (round 5): an odd-width plane (pitch padded to a multiple of 4: 31 -> 32, the encoder's conv2; 13 -> 16, the
  // decoder's conv2 data gradient: 221 -> 197 us) is staged by LDS quads instead of memory quads: a unit = four consecutive slots of one LDS row = four consecutive elements of the
  // memory row at a DWORD-aligned (not 16-byte-aligned) address -- one 16-byte buffer load, one aligned 8-byte store per
  // plane, exactly the even-pitch path (round 4 placed every element on its own, twelve 2-byte stores per unit: enc2's
  // forward was bound by them, 312 vs 309 us on the fp32 kernel).  A row's last unit spills into the pad slots.
  constexpr bool QROW = ODD;
  static_assert(!ODD || WBE % 4 == 0, "padded pitch: whole quads per row");
  constexpr int PLV_E = PLMAX_E / 4;
  constexpr int W_NV = 3 * NBLK * 2 * BM;                    // 16-byte weight vectors per chunk
  constexpr int W_PER = (W_NV + NT - 1) / NT, P_PER = (CK * (QROW ? PLV_E : PLV) + NT - 1) / NT;
  constexpr int EPI_FLOATS = (NT / 64) * 32 * 36 + (NT / 64) * TM * 32;
  constexpr int LDS_BYTES = cmax(3 * WPLANE + NPL * PPLANE + 64, 4 * EPI_FLOATS);
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char* Wl = lds;
  char* Pl = lds + 3 * WPLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / T::WN, wn = wid % T::WN;
  const int li = lane & 31, lh = lane >> 5;

  const int Ntot = p.nimg * G::PS;
  const int n0 = xcd_tile(blockIdx.x, gridDim.x) * BN, m0 = blockIdx.y * BM;
  const int nlast = min(n0 + BN, Ntot) - 1;
  const int ia = n0 / G::PS, ib = nlast / G::PS;
  const int fa = (n0 % G::PS) / G::WS, lb = (nlast % G::PS) / G::WS;
  const int la_ = (ia == ib) ? lb : G::HS - 1;
  const int lenA = ((2 * (la_ - fa) + G::KS) * G::WB + 3) & ~3;
  auto span_start = [&](int i) __attribute__((always_inline)) { return i == ia ? 0 : lenA + (i - ia - 1) * LENFULL; };
  const int PL = span_start(ib) + (ib == ia ? lenA : ((2 * lb + G::KS) * G::WB + 3) & ~3);
  const int lenA_e = (2 * (la_ - fa) + G::KS) * WBE;
  auto span_start_e = [&](int i) __attribute__((always_inline)) {
    return ODD ? (i == ia ? 0 : lenA_e + (i - ia - 1) * LENFULL_E) : span_start(i);
  };

  const __amdgpu_buffer_rsrc_t rbig = make_rsrc(p.big, p.big_bytes), rw = make_rsrc(p.w, p.w_bytes);

  // ---- staging roles
  unsigned poff[P_PER];
  int plds[P_PER];
#pragma unroll
  for (int j = 0; j < P_PER; ++j) {
    if (QROW) {
      const int v = tid + j * NT, c = v / PLV_E, qe = (v % PLV_E) * 4;   // LDS slot offset inside the channel's image
      int i, rel;
      if (qe < lenA_e) {
        i = ia;
        rel = qe;
      } else {
        i = ia + 1 + (qe - lenA_e) / LENFULL_E;
        rel = (qe - lenA_e) % LENFULL_E;
      }
      const int rows_i = (i == ia) ? 2 * (la_ - fa) + G::KS : (i == ib ? 2 * lb + G::KS : ROWS_FULL);
      const int r = rel / WBE, col = rel % WBE;
      const bool act = c < CK && i <= ib && r < rows_i;
      const int f = (i == ia) ? fa : 0;
      poff[j] = act ? 4u * (unsigned)((i * G::CB + c) * G::PB + (2 * f + r) * G::WB + col) : kOobOffset;
      plds[j] = act ? 2 * (c * PLMAX_E + span_start_e(i) + rel) : -1;
      continue;
    }
    const int v = tid + j * NT, c = v / PLV, q = (v % PLV) * 4;
    const bool act = c < CK && q < PL;
    int i, rel;
    if (q < lenA) {
      i = ia;
      rel = q;
    } else {
      i = ia + 1 + (q - lenA) / LENFULL;
      rel = (q - lenA) % LENFULL;
    }
    const int f = (i == ia) ? fa : 0;
    // (a phantom channel -- odd CB, single chunk -- loads zeros and stores them: its fragment reads meet zero weights)
    poff[j] = (act && c < G::CB) ? (unsigned)sizeof(BigT) * (unsigned)((i * G::CB + c) * G::PB + 2 * f * G::WB + rel) : kOobOffset;
    plds[j] = act ? 2 * (c * PLMAX_E + q) : -1;   // bytes inside a plane
  }
  constexpr unsigned P_STEP = (unsigned)sizeof(BigT) * CK * G::PB;
  const unsigned wbase = (unsigned)(blockIdx.y * NSL) * (unsigned)P::CHUNK_BYTES;

  // ---- per-lane byte bases of the B fragments: the lane's pixel, its half's channel of the pair
  int bbh[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = min(n0 + (wn * TN + j) * 32 + li, Ntot - 1);
    const int i = n / G::PS, pix = n % G::PS;
    const int f = (i == ia) ? fa : 0;
    bbh[j] = 2 * (span_start_e(i) + 2 * (pix / G::WS - f) * WBE + 2 * (pix % G::WS) + lh * PLMAX_E);
  }
  const int abase = (lh * BM + wm * (TM * 32) + li) * 16;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 rwv[W_PER];
  typename Patch4<BigT>::raw_t rpv[P_PER];
  auto gload = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < W_PER; ++j) {
      const int v = tid + j * NT;
      rwv[j] = VecLoad<4>::load(rw, ((W_NV % NT == 0) || v < W_NV) ? wbase + (unsigned)t * P::CHUNK_BYTES + 16u * v : kOobOffset);
    }
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
      rpv[j] = Patch4<BigT>::load(rbig, poff[j] == kOobOffset ? kOobOffset : poff[j] + (unsigned)t * P_STEP);
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < W_PER; ++j) {
      const int v = tid + j * NT;
      if ((W_NV % NT == 0) || v < W_NV) *reinterpret_cast<f32x4*>(Wl + 16 * v) = rwv[j];
    }
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
      if (plds[j] >= 0) {
        if constexpr (U8) {
          // four bytes -> four bf16 (exact): float(v) has at most 8 significant bits, its upper half IS the bf16
          const unsigned v = rpv[j];
          const unsigned f0 = __builtin_bit_cast(unsigned, (float)(v & 0xffu)), f1 = __builtin_bit_cast(unsigned, (float)((v >> 8) & 0xffu));
          const unsigned f2 = __builtin_bit_cast(unsigned, (float)((v >> 16) & 0xffu)), f3 = __builtin_bit_cast(unsigned, (float)(v >> 24));
          *reinterpret_cast<bg_u32x2*>(Pl + plds[j]) = bg_u32x2{(f0 >> 16) | (f1 & 0xffff0000u), (f2 >> 16) | (f3 & 0xffff0000u)};
        } else {
          unsigned a1, a2, a3, b1, b2, b3;
          bg_split3(rpv[j][0], rpv[j][1], a1, a2, a3);
          bg_split3(rpv[j][2], rpv[j][3], b1, b2, b3);
          *reinterpret_cast<bg_u32x2*>(Pl + plds[j]) = bg_u32x2{a1, b1};
          *reinterpret_cast<bg_u32x2*>(Pl + PPLANE + plds[j]) = bg_u32x2{a2, b2};
          *reinterpret_cast<bg_u32x2*>(Pl + 2 * PPLANE + plds[j]) = bg_u32x2{a3, b3};
        }
      }
  };
  // byte offset (inside a plane, from the half's base) of the dword that holds slots q, q + 1 of channel pair g
  auto qoff = [](int g, int q) constexpr {
    const int ky = q / BG::KSE, kx = q % BG::KSE;
    return (q < BG::TAPS && kx < G::KS) ? 2 * (2 * g * PLMAX_E + ky * WBE + kx) : 2 * (2 * g * PLMAX_E);
  };
  auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int g = blk / BG::BPG, b = blk % BG::BPG;
      bg_bf16x8 fa[TM][3], fb[TN][NPL];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i][pl] = *reinterpret_cast<const bg_bf16x8*>(Wl + pl * WPLANE + blk * (2 * BM * 16) + abase + i * 32 * 16);
#pragma unroll
        for (int j = 0; j < (pl < NPL ? TN : 0); ++j) {
          const char* pb = Pl + pl * PPLANE + bbh[j];
          u32x4s d;
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) d[pp] = *reinterpret_cast<const unsigned*>(pb + qoff(g, 8 * b + 2 * pp));
          fb[j][pl] = __builtin_bit_cast(bg_bf16x8, d);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];  // smallest terms first
          if constexpr (U8) {   // the byte plane is exact: w = w1 + w2 + w3 against it
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
          } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][NPL > 1 ? 1 : 0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][NPL > 2 ? 2 : 0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][NPL > 1 ? 1 : 0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
          }
          acc[i][j] = c;
        }
    }
  };

  gload(0);
  if (ODD) {
    // the pad column of every row is read (slot kx = KS of an odd kernel, against a zero weight): it must hold a
    // finite value, and no store below ever touches it
    for (int i = tid; i < NPL * PPLANE / 16; i += NT) reinterpret_cast<f32x4*>(Pl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
  }
  lstore();
  __syncthreads();
  for (int t = 0; t < NSL; ++t) {
    if (t + 1 < NSL) gload(t + 1);
#ifndef BC_NO_COMPUTE   // ablation builds (profiles/r05_bconv_ablation.txt): results wrong, time meaningful
    compute();
#endif
    __syncthreads();
    if (t + 1 < NSL) {
#ifndef BC_NO_STAGE
      lstore();
#endif
      __syncthreads();
    }
  }
  dconv_down_epilogue<G, T>(p, reinterpret_cast<float*>(lds), acc, n0, m0, Ntot);
}

template <class G, class T, class BigT = float>
inline int launch_bconv_down(const DownArgs& a, const float* w, char* pack, hipStream_t s) {
  typedef BPack<G, T> P;
  constexpr bool U8 = std::is_same<BigT, uint8_t>::value;
  const int total = P::MT * P::NSL * P::NBLK * 2 * T::BM;
  hipLaunchKernelGGL((bconv_pack_kernel<G, T, U8>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, BPackArgs{w, pack, a.bias});
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  DownArgs b = a;
  b.w = reinterpret_cast<const float*>(pack);
  b.w_bytes = (unsigned)P::BYTES;
  if (U8) b.bias = reinterpret_cast<const float*>(pack + P::BYTES);   // the folded bias: b - sum w (pack kernel)
  const long gx = ((long)a.nimg * G::PS + T::BN - 1) / T::BN, gy = P::MT;
  hipLaunchKernelGGL((bconv_down_kernel<G, T, BigT>), dim3((unsigned)gx, (unsigned)gy), dim3(T::NT), 0, s, b);
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
