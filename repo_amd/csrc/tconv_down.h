// Decoder conv3's data gradient (the stride-2 "down" convolution 32 ch @ 30 x 30 -> 64 ch @ 13 x 13, k = 6) on the bf16
// matrix pipe ("bf16x6", bgemm.h) with waves that ONLY stage beside waves that ONLY multiply.
//
//   small[img][cs][sy][sx] = sum_{cb,ky,kx} big[img][cb][2sy+ky][2sx+kx] w[cs][cb][ky][kx]
//   M = cs (A = weights), N = the 169 pixels of ONE image (12 tiles of <= 16), K = (8 channels of a chunk) x (4 taps) per MFMA
//
// bconv.h stages, then multiplies, all waves in turn: a fifth of its time is staging that nothing hides
// (profiles/r05_bconv_ablation.txt).  Here, as in twgrad.h / tconv_up.h:
//   * `big` is staged CHANNEL-INNERMOST -- per image and 8-channel chunk three bf16 planes [row][column parity][x/2][8 ch],
//     16 B per pixel, the odd columns 256 B behind the even ones (44 KB; two buffers) -- so the B fragment of
//     v_mfma_f32_16x16x32_bf16 (column = pixel, a lane group's 8 k = the chunk's 8 channels at ONE tap) is one
//     ds_read_b128 per plane at
//        base(pixel) + (ky * row + (kx & 1) * 256 + (kx >> 1) * 16 B)(tap of the lane's group),
//     stride 2 and the tap in the address, no gather, no vector arithmetic per use;
//   * conflict-free by construction (ds_read_b128 is served in four groups of sixteen NON-contiguous lanes, each holding
//     all sixteen pixels of a tile and two neighbouring lane groups): neighbouring lane groups take the taps (ky, 2j) and
//     (ky, 2j+1) (256 B apart: the same banks), and lane ln of a tile takes a pixel of bank slot ln = (14 sy + sx) mod 16,
//     one per tile in row order -- twelve tiles of at most sixteen pixels (sixteen CONSECUTIVE pixels of 13-pixel rows never
//     fill the sixteen slots);
//   * the weights arrive pre-split and fragment-ready from a pack (tconv_down_pack_kernel): per (chunk, 4 taps)
//     [plane][16-row tile][lane group = tap][row][8 ch] = 12 KB (a tile's A fragment = 1 KB at 16 B x lane), streamed from
//     L2 (all workgroups read the same 442 KB: profiles/r05_l2_weight_stream.txt) through an LDS ring of two 3-step groups
//     by the staging waves;
//   * a PERSISTENT workgroup per CU takes images b, b + grid, ...; 8 waves: waves 0-3 multiply -- wave = (32 cs) x (6 of
//     the 12 pixel tiles), B fragments requested two (step, tile) items ahead -- waves 4-7 stage the next chunk's patch and
//     the next group's weights; one LDS-only barrier per 3 MFMA steps (12 taps x 8 channels);
//   * epilogue by the multiplying waves while the staging waves are already on the next image; the ReLU operand is
//     requested a chunk before the image completes, the stores are not waited for.
// Versions, ablations and counters: profiles/r05_tconv_down.txt.
// Reference: autograd's input gradient of nn.ConvTranspose2d(64, 32, 6, stride 2) (models/decoder.py:43-47).
#pragma once
#include "bgemm.h"
#include "dconv.h"
#include "twgrad.h"

namespace repo {

// KS: kernel (even), WB: width = height of `big`, WS: of `small`; GST: MFMA steps per weight group (= per barrier); NTL: pixel
// tiles per multiplying wave (2 NTL tiles of <= 16 pixels must hold every bank slot's pixels: see the kernel)
template <int KS_, int WB_, int WS_, int GST_, int NTL_>
struct TcdGeoT {
  static constexpr int CB = 32, CS = 64, WB = WB_, HB = WB_, WS = WS_, PS = WS_ * WS_, PB = WB_ * WB_, KS = KS_, KK = KS_ * KS_;
  static_assert(KS % 2 == 0 && WS == (WB - KS) / 2 + 1 && HB <= 32, "even kernel, stride 2, at most 32 rows (the staging items)");
  static constexpr int CC = 8;                         // channels per chunk
  static constexpr int NCH = CB / CC;                  // chunks per image
  static constexpr int NST = KK / 4;                   // MFMA steps (4 taps x 8 channels) per chunk
  static constexpr int GST = GST_, NGR = NST / GST;    // steps per weight group (= per barrier), groups per chunk
  static constexpr int XH = (WB + 1) / 2, PIXB = 16;
  // a staged row: [column parity][x/2][8 ch], the odd columns 256 B behind the even ones -- the taps (ky, 2j) and (ky, 2j+1)
  // of a lane-group pair then read the SAME banks, which is what ds_read_b128's lane groups (half of one lane group of
  // 16, half of its neighbour: MI355X_MICROARCH.md, LDS) need to stay conflict-free
  static constexpr int PO = 256, ROWB = PO + XH * PIXB;
  static_assert(XH * PIXB <= PO, "a column-parity plane of a row fits 256 B");
  // the lane group's tap at step s: pair 2 s + (kg >> 1) of the KS * KS / 2 (ky, j) pairs, column 2 j + (kg & 1)
  static constexpr int KP = KS / 2;
  static constexpr int tap_of(int s, int kg) { return ((2 * s + (kg >> 1)) / KP) * KS + 2 * ((2 * s + (kg >> 1)) % KP) + (kg & 1); }
  // a pixel's 16-byte slot in the 256-byte bank row: (SLOTSTEP sy + sx) mod 16
  static constexpr int SLOTSTEP = (2 * ROWB / PIXB) % 16;
  static constexpr int PPLANE = HB * ROWB, PBUF = 3 * PPLANE;
  static constexpr int WPL = CS * 4 * 16;              // one plane of one step: [16-row tile][lane group][row][8 ch]
  static constexpr int WSTEP = 3 * WPL, WGRP = GST * WSTEP;
  static constexpr int LDS_BYTES = 2 * PBUF + 2 * WGRP;
  static constexpr int NTL = NTL_;                     // pixel tiles per multiplying wave
  static constexpr int W_PER = WGRP / 16 / 256;        // 16-byte vectors per staging thread and group
  static constexpr int QPR = (WB + 3) / 4, P_PER = 2;  // pixel quads per row; items per staging thread
  static_assert(WGRP % (16 * 256) == 0 && QPR == 8 && NST % GST == 0 && NCH % 2 == 0 && NGR <= 3 && NGR >= P_PER, "staging shares");
  static constexpr size_t PACK_BYTES = (size_t)NCH * NST * WSTEP;
};
using TcdGeo = TcdGeoT<6, 30, 13, 3, 6>;     // decoder conv3's data gradient
using TcdGeoE2 = TcdGeoT<4, 31, 14, 2, 7>;   // encoder conv2's forward
constexpr size_t kTcdPackBytes = TcdGeo::PACK_BYTES;

// one thread per (chunk, step, cs, lane group): the 8 channels of one A fragment's lane, all three planes
template <class T>
__global__ __launch_bounds__(256) void tconv_down_pack_kernel(const float* w, char* pack) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T::NCH * T::NST * T::CS * 4) return;
  const int kg = i & 3, cs = (i >> 2) % T::CS, st = (i >> 2) / T::CS;   // st = chunk * NST + step
  const int c = st / T::NST, tap = T::tap_of(st % T::NST, kg);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = w[((size_t)cs * T::CB + T::CC * c + j) * T::KK + tap];
  unsigned pl[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bg_split3(v[2 * e], v[2 * e + 1], pl[0][e], pl[1][e], pl[2][e]);
  // per 16-row tile [lane group][row][8 ch]: the tile's A fragment is 1 KB read at 16 B x lane (conflict-free for ds_read_b128)
  char* dst = pack + (size_t)st * T::WSTEP + ((cs >> 4) * 64 + kg * 16 + (cs & 15)) * 16;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4s*>(dst + q * T::WPL) = u32x4s{pl[q][0], pl[q][1], pl[q][2], pl[q][3]};
}

// KEPI: REPO_EPI_NONE, REPO_EPI_MUL_DRELU (aux = the fp32 activation whose ReLU the gradient passes through) or
// REPO_EPI_RELU (+ bias; + the output's channel-quad mask when p.cmask: a lane's four accumulator rows ARE one channel quad)
template <class T, int KEPI>
__global__ __launch_bounds__(512) void tconv_down_kernel(DownArgs p) {
  extern __shared__ __attribute__((aligned(16))) char td_lds[];
  char* const Pb = td_lds;
  char* const Wr = td_lds + 2 * T::PBUF;
  const int tid = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nloc = (p.nimg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // images of this workgroup
  if (nloc <= 0) return;
  const int nchunks = nloc * T::NCH;

  if (wid < 4) {
    // =================================================================== the multiplying waves
    __builtin_amdgcn_s_setprio(2);   // (409.6 / 415.0 against 418.2 / 425.8 us at priority 0, same box)
    const int lane = tid & 63;
    const int mh = wid & 1, nh = wid >> 1;
    const int ln = lane & 15, kg = lane >> 4;
    // Which pixel a lane holds in a tile.  A pixel's 16-byte slot in the 256-byte bank row is (2 sy ROWB / 16 + sx) mod 16
    // (decoder conv3: (14 sy + sx) mod 16): sixteen CONSECUTIVE pixels never fill the sixteen slots (a row wrap jumps by two),
    // and every one of ds_read_b128's four lane groups would take a second cycle.  Lane ln therefore takes the pixels of slot
    // ln, one per tile in row order (decoder conv3: 9-12 pixels per slot, twelve tiles, 6 per wave; encoder conv2, rows 512 B
    // apart: slot = sx, 14 pixels in each of 14 slots, fourteen tiles; a lane without a pixel in a tile re-reads its first).
    int pxs[T::NTL], pfirst = 0;
#pragma unroll
    for (int t = 0; t < T::NTL; ++t) pxs[t] = -1;
    {
      int cnt = 0;
#pragma unroll
      for (int sy = T::WS - 1; sy >= 0; --sy) {   // (descending: pfirst ends as the lowest row's)
        const int sx = (ln - T::SLOTSTEP * sy) & 15;
        if (sx < T::WS) pfirst = T::WS * sy + sx;
      }
#pragma unroll
      for (int sy = 0; sy < T::WS; ++sy) {
        const int sx = (ln - T::SLOTSTEP * sy) & 15;
        const int idx = cnt - T::NTL * nh;
#pragma unroll
        for (int t = 0; t < T::NTL; ++t)
          if (sx < T::WS && idx == t) pxs[t] = T::WS * sy + sx;
        cnt += sx < T::WS ? 1 : 0;
      }
    }
    int nbase[T::NTL];
#pragma unroll
    for (int t = 0; t < T::NTL; ++t) {
      const int px = pxs[t] < 0 ? pfirst : pxs[t];
      const int sy = px / T::WS, sx = px - sy * T::WS;
      nbase[t] = 2 * sy * T::ROWB + sx * T::PIXB;
    }
    // per step: the lane group's tap
    int toff[T::NST];
#pragma unroll
    for (int s = 0; s < T::NST; ++s) {
      const int pr = 2 * s + (kg >> 1), ky = pr / T::KP, kx = 2 * (pr - T::KP * ky) + (kg & 1);
      toff[s] = ky * T::ROWB + (kx & 1) * T::PO + (kx >> 1) * T::PIXB;
    }
    const int abase = 2 * mh * 1024 + lane * 16;

    f32x4 acc[2][T::NTL];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < T::NTL; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out, 4u * (unsigned)p.nimg * T::CS * T::PS);
    const __amdgpu_buffer_rsrc_t raux = make_rsrc(KEPI == REPO_EPI_MUL_DRELU ? p.aux : p.out, 4u * (unsigned)p.nimg * T::CS * T::PS);

    // the epilogue's byte offsets (image 0): cs = 32 mh + 16 m + 4 (lane >> 4) + r, the lane's pixel of the tile; a lane
    // without one is out of the descriptors' range
    unsigned eoff[T::NTL];
#pragma unroll
    for (int t = 0; t < T::NTL; ++t) {
      eoff[t] = (4u * (unsigned)((32 * mh + 4 * kg) * T::PS + max(pxs[t], 0))) | ((unsigned)(pxs[t] >> 31) & kOobOffset);
    }
    float av[2][T::NTL][4];   // REPO_EPI_MUL_DRELU: the ReLU operand, requested a chunk before the image completes
    float bv[2][4];           // REPO_EPI_RELU: the bias of the lane's channels
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[m][r] = (KEPI == REPO_EPI_RELU && p.bias) ? p.bias[32 * mh + 16 * m + 4 * kg + r] : 0.f;

    lds_barrier();   // chunk 0 and weight group 0 are staged
    int ph = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
      const char* Pl = Pb + (cc & 1) * T::PBUF;
      const unsigned ibase = 4u * (unsigned)(((int)blockIdx.x + (cc / T::NCH) * (int)gridDim.x) * T::CS * T::PS);
      if (KEPI == REPO_EPI_MUL_DRELU && (cc & (T::NCH - 1)) == T::NCH - 1) {
#pragma unroll
        for (int t = 0; t < T::NTL; ++t)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              av[m][t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(raux, eoff[t] + ibase + 4u * (unsigned)((16 * m + r) * T::PS), 0, 0));
      }
      bg_bf16x8 fa[2][2][3], fb[3][3];
#pragma unroll
      for (int g = 0; g < T::NGR; ++g) {
        const char* Wl = Wr + (ph & 1) * T::WGRP;
        // one item = one (step, tile): its 12 MFMAs (two row tiles x six products) run while the next item's three reads
        // (and, at a step's first tile, the next step's six weight reads) are in flight
        auto load_a = [&](bg_bf16x8(&fa)[2][3], int k) __attribute__((always_inline)) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q)
              fa[m][q] = *reinterpret_cast<const bg_bf16x8*>(Wl + k * T::WSTEP + q * T::WPL + abase + m * 1024);
        };
        auto load_b = [&](bg_bf16x8(&fb)[3], int s, int t) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < 3; ++q) fb[q] = *reinterpret_cast<const bg_bf16x8*>(Pl + q * T::PPLANE + nbase[t] + toff[s]);
        };
        // (the patch does not change inside a chunk: the first two items' B fragments of groups 1, 2 were requested in
        // front of the barrier that ended the group before)
        load_a(fa[0], 0);
        if (g == 0) {
          load_b(fb[0], 0, 0);
          load_b(fb[1], 0, 1);
        }
        constexpr int NIT = T::GST * T::NTL;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int k = it / T::NTL, t = it % T::NTL;
          if (it + 2 < NIT) load_b(fb[(it + 2) % 3], g * T::GST + (it + 2) / T::NTL, (it + 2) % T::NTL);
          if (t == 0 && k + 1 < T::GST) load_a(fa[(k + 1) & 1], k + 1);
          __builtin_amdgcn_sched_barrier(0);
          // per accumulator the smallest terms first; the two row tiles interleaved
          constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
          for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#ifdef TCD_NO_MFMA
              acc[m][t][0] += __builtin_bit_cast(float, (int)fa[k & 1][m][PA[pr]][0] + (int)fb[it % 3][PB[pr]][0]);
#else
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[k & 1][m][PA[pr]], fb[it % 3][PB[pr]], acc[m][t], 0, 0, 0);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        if (g + 1 < T::NGR) {
          load_b(fb[0], (g + 1) * T::GST, 0);
          load_b(fb[1], (g + 1) * T::GST, 1);
        }
        lds_barrier();
        ++ph;
      }
#ifndef TCD_NO_EPI
      if ((cc & (T::NCH - 1)) == T::NCH - 1)
#else
      if (cc == nchunks - 1)
#endif
      {
        // ---- the image is complete
        const int img = (int)blockIdx.x + (cc / T::NCH) * (int)gridDim.x;
#pragma unroll
        for (int t = 0; t < T::NTL; ++t)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            unsigned bits = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x = acc[m][t][r];
              if (KEPI == REPO_EPI_MUL_DRELU) x = av[m][t][r] > 0.f ? x : 0.f;
              if (KEPI == REPO_EPI_RELU) {
                x = fmaxf(x + bv[m][r], 0.f);
                bits |= (x > 0.f ? 1u : 0u) << r;
              }
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rout, eoff[t] + ibase + 4u * (unsigned)((16 * m + r) * T::PS), 0, 0);
              acc[m][t][r] = 0.f;
            }
            // the channel-quad mask of the output (repo_hip.h REPO_EPI_MUL_CMASK): byte (image, quad, pixel), bit = channel & 3
            if (KEPI == REPO_EPI_RELU && p.cmask && pxs[t] >= 0)
              p.cmask[((size_t)img * (T::CS / 4) + 8 * mh + 4 * m + kg) * T::PS + pxs[t]] = (unsigned char)bits;
          }
      }
    }
  } else {
    // =================================================================== the staging waves
    const int t_ = tid - 256;
    const __amdgpu_buffer_rsrc_t rbg = make_rsrc(p.big, p.big_bytes), rw = make_rsrc(p.w, (unsigned)T::PACK_BYTES);
    // TWO register sets each (group gi in set gi & 1, chunk cc's patch in set cc & 1): a weight group is requested two
    // phases before it is stored, a chunk's patch a whole chunk before (with two phases per chunk -- encoder conv2 -- one
    // set would put a load and its store either side of ONE barrier; on decoder conv3 the second set measured neutral)
    f32x4 rwv[2][T::W_PER], rpv[2][T::P_PER][4];
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // weight group gi (of this workgroup's sequence): the pack is walked linearly, once per image
    auto wload = [&](auto sc, int gi) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value;
      const unsigned base = (unsigned)(gi % (T::NCH * T::NGR)) * (unsigned)T::WGRP;
#pragma unroll
      for (int i = 0; i < T::W_PER; ++i) rwv[S][i] = VecLoad<4>::load(rw, base + 16u * (unsigned)(t_ + 256 * i));
    };
    auto wstore = [&](auto sc, int gi) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value;
      char* Wl = Wr + (gi & 1) * T::WGRP;
#pragma unroll
      for (int i = 0; i < T::W_PER; ++i) *reinterpret_cast<f32x4*>(Wl + 16 * (t_ + 256 * i)) = rwv[S][i];
    };
    // patch item i of chunk cc: v = t_ + 256 i -> channel quad v & 1, pixel quad q, row; a ds_write_b64 lane group (16
    // consecutive lanes) is [row & 1][q & 3][channel quad]: 8 B at 32 q + 8 cq of two rows 496 = 112 (mod 128) B apart --
    // every one of the 32 store banks once (decoder conv3; rows of 512 B: two-way).  Pixel quad q starts at column 4 q; on an
    // even width the last quad ends WITH the row (it re-stages two pixels), on an odd one it runs one column past the row:
    // that element lands in a slot no tap reads (x/2 = 15 of the odd plane)
    auto x0_of = [](int q) { return T::WB % 2 == 0 ? min(4 * q, T::WB - 4) : 4 * q; };
    auto pload = [&](auto sc, int cc) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value;
      const int img = (int)blockIdx.x + (cc / T::NCH) * (int)gridDim.x, c = cc & (T::NCH - 1);
      const unsigned dead_c = (unsigned)((nchunks - 1 - cc) >> 31);
#pragma unroll
      for (int i = 0; i < T::P_PER; ++i) {
        const int v = t_ + 256 * i, cq = v & 1, q = ((v >> 1) & 3) | (((v >> 4) & 1) << 2), row = ((v >> 3) & 1) | ((v >> 5) << 1);
        const unsigned dead = ((unsigned)((T::HB - 1 - row) >> 31) | dead_c) & kOobOffset;
        const int x0 = x0_of(q);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
          rpv[S][i][ch] = VecLoad<4>::load(rbg, (4u * (unsigned)((img * T::CB + T::CC * c + 4 * cq + ch) * T::PB + row * T::WB + x0)) | dead);
      }
    };
    auto pstore = [&](auto sc, int cc, int i) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value;
      char* Pl = Pb + (cc & 1) * T::PBUF;
      const int v = t_ + 256 * i, cq = v & 1, q = ((v >> 1) & 3) | (((v >> 4) & 1) << 2), row = ((v >> 3) & 1) | ((v >> 5) << 1);
      if (row < T::HB) {
        const int x0 = x0_of(q);   // even: pixel x0 + e has column parity e & 1, x/2 = x0/2 + (e >> 1)
        char* base = Pl + row * T::ROWB + (x0 >> 1) * T::PIXB + cq * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned a1, a2, a3, b1, b2, b3;
          tw_split3(rpv[S][i][0][e], rpv[S][i][1][e], a1, a2, a3);
          tw_split3(rpv[S][i][2][e], rpv[S][i][3][e], b1, b2, b3);
          char* dst = base + (e & 1) * T::PO + (e >> 1) * T::PIXB;
          *reinterpret_cast<bg_u32x2*>(dst) = bg_u32x2{a1, b1};
          *reinterpret_cast<bg_u32x2*>(dst + T::PPLANE) = bg_u32x2{a2, b2};
          *reinterpret_cast<bg_u32x2*>(dst + 2 * T::PPLANE) = bg_u32x2{a3, b3};
        }
      }
    };

    // prologue: chunk 0's patch and weight group 0 in LDS; groups 1, 2 and the patches of chunks 1, 2 in registers
    pload(I0{}, 0);
    wload(I0{}, 0);
    pstore(I0{}, 0, 0);
    pstore(I0{}, 0, 1);
    wstore(I0{}, 0);
    wload(I1{}, 1);
    wload(I0{}, 2);
    pload(I1{}, 1);
    pload(I0{}, 2);
    lds_barrier();
    // phase g of chunk cc (its group gi = NGR cc + g is being multiplied): group gi + 1 goes to the other ring slot (free since
    // the last barrier) and its registers take group gi + 3; chunk cc + 1's patch goes to the other patch buffer (free since
    // chunk cc - 1 ended), one item per phase, and its registers take chunk cc + 3
    auto phase = [&](auto pc, auto gc, int cc) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value, g = decltype(gc)::value;   // P = cc & 1
      constexpr int SW = (T::NGR * P + g + 1) & 1, SP = (P + 1) & 1;    // (gi + 1) & 1;  (cc + 1) & 1
      const int gi = T::NGR * cc + g;
#ifndef TCD_NO_STAGE   // ablation builds (tools/build_variant.sh): results wrong, time meaningful
      wstore(std::integral_constant<int, SW>{}, gi + 1);
      wload(std::integral_constant<int, SW>{}, gi + 3);
      if constexpr (g < T::P_PER) pstore(std::integral_constant<int, SP>{}, cc + 1, g);
      if constexpr (g == T::P_PER - 1) pload(std::integral_constant<int, SP>{}, cc + 3);
#endif
      lds_barrier();
    };
    auto chunk = [&](auto pc, int cc) __attribute__((always_inline)) {
      phase(pc, std::integral_constant<int, 0>{}, cc);
      if constexpr (T::NGR > 1) phase(pc, std::integral_constant<int, 1>{}, cc);
      if constexpr (T::NGR > 2) phase(pc, std::integral_constant<int, 2>{}, cc);
    };
    for (int cc = 0; cc < nchunks; cc += 2) {   // (NCH is even: whole pairs)
      chunk(I0{}, cc);
      chunk(I1{}, cc + 1);
    }
  }
}

template <class T, int KEPI>
inline int launch_tconv_down_k(const DownArgs& a, int grid, hipStream_t s) {
  static_assert(T::LDS_BYTES <= 160 * 1024, "tconv_down: LDS");
  hipError_t e = hipFuncSetAttribute((const void*)tconv_down_kernel<T, KEPI>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((tconv_down_kernel<T, KEPI>), dim3((unsigned)grid), dim3(512), T::LDS_BYTES, s, a);
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// pack (>= T::PACK_BYTES) is written here: the weights change once per optimiser step, the pack is one small launch per call.
// T = TcdGeo: a.epi NONE / MUL_DRELU, no bias;  T = TcdGeoE2: a.epi RELU (+ a.bias, + a.cmask)
template <class T>
inline int launch_tconv_down(const DownArgs& a, const float* w, char* pack, hipStream_t s) {
  hipLaunchKernelGGL(tconv_down_pack_kernel<T>, dim3((T::NCH * T::NST * T::CS * 4 + 255) / 256), dim3(256), 0, s, w, pack);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  DownArgs b = a;
  b.w = reinterpret_cast<const float*>(pack);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  const int grid = a.nimg < cus ? a.nimg : cus;
  if (a.epi == REPO_EPI_RELU) return launch_tconv_down_k<T, REPO_EPI_RELU>(b, grid, s);
  return a.epi == REPO_EPI_MUL_DRELU ? launch_tconv_down_k<T, REPO_EPI_MUL_DRELU>(b, grid, s) : launch_tconv_down_k<T, REPO_EPI_NONE>(b, grid, s);
}

}  // namespace repo
