// Scatter-form transposed convolution (decoder forward / encoder data-gradient) on the fp32 matrix cores.
//
//   big[img][cb][2iy+ky][2ix+kx] += sum_cs small[img][cs][iy][ix] * w[cs][cb][ky][kx]
//
// dconv_up_kernel (dconv.h) computes this OUTPUT-stationary: a lane owns an output pixel of a parity
// class and walks the J x J taps that can reach it, reading a zero halo where a tap falls off the input
// plane.  For the small planes of this model the halo is a large share of the executed MFMAs (13->30:
// x1.33, 5->13: x1.96, 2->6: x2.25).  Here the product is INPUT-stationary and exact: for one tap (ky,kx)
//   C_tap[cb][pixel] = sum_cs w[cs][cb][ky][kx] * small[cs][pixel]
// is a dense GEMM over the real input pixels only, and its result is ADDED into the output at the
// tap's shift.  Output pixels of different parity (y&1, x&1) are reached by disjoint taps, so
//   * each of the 4 waves of a workgroup owns ONE parity class: its 16-channel class plane lives in LDS
//     ([class][cb][cy][cx], 4 * 16 * ~NYM*NXM floats) and is updated with plain read-add-write -- no
//     atomics (bit-reproducible), no barrier in the main loop, no two waves ever touch the same word;
//   * the input pixels of the workgroup's images are the MFMA's N dimension (16 per v_mfma_f32_16x16x4_f32)
//     and stay in REGISTERS as B fragments for all taps (K/4 registers per 16 pixels), loaded straight
//     from global memory (4 x 64-byte segments per wave-load, zero-filled past the last image);
//   * the weights of (16 output channels, class, tap) are the A fragments: K/4 registers, streamed one
//     tap ahead from a fragment-ready pack (one coalesced 256-byte wave-load per register);
//   * a lane's LDS address is  base(pixel, class, lane quarter) + immediate(tap shift, accumulator row),
//     the old values are requested BEFORE the tap's MFMA chain and added / written back after it.
// The epilogue adds the bias, applies ReLU (decoder) or the ReLU mask of the layer input (encoder
// data-gradient) and writes the 16 output planes of each image as contiguous dwords.
//
// Executed / useful MFMAs = N-tile padding only (GI images * PS pixels rounded up to 16): 1.00-1.06.
//
// Reference: nn.ConvTranspose2d forward in VisualObservationModel (models/decoder.py:43-47) and autograd's
// input gradient of nn.Conv2d in VisualEncoder (models/encoder.py:35-38).
#pragma once
#include "dconv.h"

namespace repo {

typedef float f32x4acc __attribute__((ext_vector_type(4)));

template <class G, int GI_, int NC_>
struct SConf {
  static constexpr int GI = GI_;                 // images per workgroup
  static constexpr int NC = NC_;                 // N tiles whose B fragments are resident at a time
  static constexpr int KST = G::CS / 4;          // k-steps of v_mfma_f32_16x16x4_f32
  static constexpr int NGRP = G::CB / 16;        // 16-channel output groups
  static constexpr int NPX = GI * G::PS;
  static constexpr int NT = (NPX + 15) / 16;
  static constexpr int J = (G::KS + 1) / 2;
  // weight-fragment buffers, indexed by tap number mod NB: consecutive valid taps of a class (t, then t+1 or t+2)
  // must differ, and the last tap must not sit in buffer 0 (the next chunk's first tap is fetched there)
  static constexpr int NB = ((J * J - 1) % 3 != 0) ? 3 : 2;
  static constexpr int NYM = (G::HB + 1) / 2, NXM = (G::WB + 1) / 2;
  // class planes: [class][channel quad q = 0..3][cy * NXM + cx][4 channels]: the 4 accumulator rows of a lane
  // (channels 4q..4q+3 of its pixel) are ONE 16-byte word, so the read-add-write of a tile is one ds_read_b128 +
  // one ds_write_b128 per lane (the 16 lanes of a quarter touch 16 consecutive words: conflict-free)
  static constexpr int PLANE = NYM * NXM;         // pixels per class plane
  static constexpr int IMG_LDS = 4 * 4 * PLANE * 4;  // floats per image
  static constexpr int LDS_FLOATS = GI * IMG_LDS;
  // lanes without a real pixel (N-tile padding, images past the end) compute exact zeros; they add them into a
  // scratch strip behind the planes instead of being predicated off (no EXEC juggling between MFMAs).
  static constexpr int DUMMY_FLOATS = 4 * (64 + (J - 1) * (NXM + 1) + 4);
  static constexpr int LDS_TOTAL_FLOATS = LDS_FLOATS + DUMMY_FLOATS;
  static constexpr size_t PACK_FLOATS = (size_t)NGRP * 4 * J * J * KST * 64;
  static_assert(G::CS % 16 == 0 && G::CB % 16 == 0, "channel counts must fit the 16x16x4 MFMA, 4 k-steps per load");
};

// ---- fragment-ready weight pack: Wp[grp][cls][tap][s/4][lane][s%4] = w[4s + lane/16][16 grp + lane%16][py+2ty][px+2tx]
//      (a lane fetches the A fragments of four consecutive k-steps with one 16-byte load)
struct UPackArgs {
  const float* w;
  float* wp;
};
template <class G>
__global__ __launch_bounds__(256) void uconv_pack_kernel(UPackArgs p) {
  constexpr int J = (G::KS + 1) / 2, KST = G::CS / 4;
  const int total = (G::CB / 16) * 4 * J * J * KST * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int s4 = i & 3, lane = (i >> 2) & 63;
    int r = i >> 8;
    const int s = 4 * (r % (KST / 4)) + s4;
    r /= KST / 4;
    const int tap = r % (J * J);
    r /= J * J;
    const int cls = r & 3, grp = r >> 2;
    const int ky = (cls >> 1) + 2 * (tap / J), kx = (cls & 1) + 2 * (tap % J);
    const int cs = 4 * s + (lane >> 4), cb = 16 * grp + (lane & 15);
    p.wp[i] = (ky < G::KS && kx < G::KS) ? p.w[((size_t)cs * G::CB + cb) * G::KK + ky * G::KS + kx] : 0.f;
  }
}

struct UScatArgs {
  const float* small;
  const float* wp;  // pack
  const float* bias;
  const float* aux;
  float* out;
  int nimg, epi;
  unsigned small_bytes, wp_bytes;
};

// A chunk = up to NC consecutive N tiles (16 pixels each) of one workgroup tile; wave-uniform.
struct UChunk {
  int img0, grp, tile0;
  bool valid;
};

// B fragments of N tile `j` of chunk d: lane (pixel lp, k quarter lq) holds small[img][4s + lq][pixel],
// s = 0..KST-1; lanes past the last real pixel / image read zeros.
template <class G, class C>
__device__ __forceinline__ void uconv_load_b(const UScatArgs& p, const UChunk& d, int j, int lane, float (&b)[C::KST]) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.small, p.small_bytes);
  const int lp = lane & 15, lq = lane >> 4;
  const int q = (d.tile0 + j) * 16 + lp;  // flattened (image, pixel) of this lane
  const int il = q / G::PS, pix = q % G::PS;
  const bool ok = q < C::NPX && d.img0 + il < p.nimg;
  const unsigned off = 4u * (unsigned)(((d.img0 + il) * G::CS + lq) * G::PS + pix);
  // the k-step part of the address (c = 16 PS s bytes) is split into the instruction's 12-bit immediate (c % 4096,
  // folded from the constant add below) and a scalar offset that takes only a few distinct values (multiples of
  // 4096): neither a VGPR nor a live SGPR per load
  const unsigned base = ok ? off : kOobOffset;
#pragma unroll
  for (int s = 0; s < C::KST; ++s) {
    constexpr unsigned STEP = 4u * 4 * G::PS;
    b[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, base + (STEP * s) % 4096u, (STEP * s) / 4096u * 4096u, 0));
  }
}

// All taps of one parity class over one chunk of NTL (compile-time) N tiles, for one compute wave.
//
// Tiles run in pairs (two independent accumulator chains per A fragment).  The read-add-write of a pair is
// software-pipelined INTO the next pair's MFMA chain, in program order (the wave issues in order, so this is
// what lets the LDS / VALU work run under the matrix pipe):
//     k-steps 0..3 of pair P : add + write back row r = s of the PREVIOUS pair (its chain finished >= 64 cycles ago)
//     k-steps 4..7 of pair P : request the old values of P's own row r = s - 4 (needed when P's chain is done)
// LDS executes a wave's operations in order, so P's reads always see the previous pair's writes, whether the two
// alias (consecutive taps do) or not.  The wave never waits for global memory either: a tap's weight fragments are
// requested one tap ahead (NB rotating buffers indexed by the tap number), and during the chunk's LAST tap the
// NEXT chunk's B fragments are requested into the registers of each pair as soon as its chain has been issued
// (and the next chunk's first weight fragments into buffer 0).
template <class G, class C, int NTL, int NEXT_NTL>
__device__ __forceinline__ void uconv_chunk(const UScatArgs& p, char* pl, int cls, int lane, const UChunk& d,
                                            const UChunk& nx, int dummy_ofs, float (&bfr)[C::NC][C::KST],
                                            float (&afr)[C::NB][C::KST]) {
  constexpr int KST = C::KST, J = C::J, NXM = C::NXM, PLANE = C::PLANE, NB = C::NB;
  static_assert(KST >= 8, "the pipelined write-back needs 8 k-steps per chain");
  constexpr int NP = (NTL + 1) / 2;  // pairs (the last may hold one tile)
  const int py = cls >> 1, px = cls & 1;
  const int lp = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wp, p.wp_bytes);
  // A fragments of (grp, cls): tap t, k-steps 4u..4u+3 at ((((grp*4 + cls)*J*J + t)*KST/4 + u)*64 + lane)*4.  Only
  // the lane part lives in a VGPR; the rest is the instruction's immediate / SCALAR offset (a VGPR address per load
  // would cost J*J*K/16 registers: the compiler hoists them all)
  const unsigned a_lane = 16u * (unsigned)lane;
  constexpr unsigned A_TAP = 4u * KST * 64;
  auto a_sbase = [&](int grp) { return 4u * (unsigned)((((grp * 4 + cls) * J * J) * KST) * 64); };
  auto load_a = [&](float (&a)[KST], unsigned sbase, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < KST / 4; ++u) {
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, a_lane + 1024u * (u % 4), sbase + (unsigned)t * A_TAP + 4096u * (u / 4), 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) a[4 * u + e] = v[e];
    }
  };
  // taps of this class: ky = py + 2 ty < KS, kx = px + 2 tx < KS (all J*J for even kernels)
  auto tap_ok = [&](int t) { return G::KS % 2 == 0 || (py + 2 * (t / J) < G::KS && px + 2 * (t % J) < G::KS); };
  int last_t = 0;
#pragma unroll
  for (int t = 0; t < J * J; ++t)
    if (tap_ok(t)) last_t = t;

  int lbase[NTL];  // byte offset (from the plane set) of this lane's (pixel, class, quarter) word
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int q = (d.tile0 + j) * 16 + lp;
    const int il = q / G::PS, pix = q % G::PS;
    const bool ok = q < C::NPX && d.img0 + il < p.nimg;
    const int iy = pix / G::WS, ix = pix % G::WS;
    lbase[j] = ok ? 16 * ((il * 16 + cls * 4 + lq) * PLANE + iy * NXM + ix) : dummy_ofs + 16 * lane;
  }

  // previous pair (being written back) and current pair
  f32x4acc pa0 = {0.f, 0.f, 0.f, 0.f}, pa1 = pa0, po0 = pa0, po1 = pa0;
  int pb0 = 0, pb1 = 0;  // byte addresses incl. the tap shift

#pragma unroll
  for (int t = 0; t < J * J; ++t) {
    if (!tap_ok(t)) continue;  // wave-uniform (odd kernels: the classes' tap sets differ)
    const bool last = t == last_t;
    // weights one tap ahead: the next valid tap is t+1 or t+2 (tap sets are products of prefixes of 0..J-1)
    if (t + 1 < J * J && tap_ok(t + 1)) load_a(afr[(t + 1) % NB], a_sbase(d.grp), t + 1);
    else if (t + 2 < J * J && tap_ok(t + 2)) load_a(afr[(t + 2) % NB], a_sbase(d.grp), t + 2);
    else if (last && nx.valid) load_a(afr[0], a_sbase(nx.grp), 0);
    const int shift = 16 * ((t / J) * NXM + (t % J));  // bytes
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int j0 = 2 * pr, j1 = (2 * pr + 1 < NTL) ? 2 * pr + 1 : 2 * pr;
      const bool two = 2 * pr + 1 < NTL;
      const bool have_prev = !(t == 0 && pr == 0);
      const bool ptwo = pr > 0 ? true : (NTL % 2 == 0);  // did the previous pair hold two tiles? (compile time)
      const int cb0 = lbase[j0] + shift, cb1 = lbase[j1] + shift;
      f32x4acc ca0 = {0.f, 0.f, 0.f, 0.f}, ca1 = ca0;
      f32x4acc co0 = ca0, co1 = ca0;
#pragma unroll
      for (int s = 0; s < KST; ++s) {
        ca0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[t % NB][s], bfr[j0][s], ca0, 0, 0, 0);
        if (two) ca1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[t % NB][s], bfr[j1][s], ca1, 0, 0, 0);
        if (have_prev) {
          if (s == 1) *reinterpret_cast<f32x4acc*>(pl + pb0) = po0 + pa0;
          if (s == 2 && ptwo) *reinterpret_cast<f32x4acc*>(pl + pb1) = po1 + pa1;
        }
        // The write-back above and the reads below are ordered by the LDS queue (in order per wave), but lane A's
        // write and lane B's read of the same word look independent to the compiler (same base register, different
        // constant offsets): without this fence it may hoist the reads above the writes.
        if (s == 4) __builtin_amdgcn_wave_barrier();
        if (s == 4) co0 = *reinterpret_cast<const f32x4acc*>(pl + cb0);
        if (s == 5 && two) co1 = *reinterpret_cast<const f32x4acc*>(pl + cb1);
      }
      pa0 = ca0;
      pa1 = ca1;
      po0 = co0;
      po1 = co1;
      pb0 = cb0;
      pb1 = cb1;
      if (last && nx.valid) {  // this pair's B registers are free: fetch the next chunk's tiles into them
        if (2 * pr < NEXT_NTL) uconv_load_b<G, C>(p, nx, 2 * pr, lane, bfr[2 * pr]);
        if (2 * pr + 1 < NEXT_NTL) uconv_load_b<G, C>(p, nx, 2 * pr + 1, lane, bfr[(2 * pr + 1) % C::NC]);
      }
      // the unrolled body holds hundreds of independent loads; without this fence the scheduler hoists the
      // later taps' weight loads and LDS reads up here and runs out of registers
      __builtin_amdgcn_sched_barrier(0);
    }
    if (last && nx.valid) {
#pragma unroll
      for (int j = 2 * NP; j < NEXT_NTL; ++j) uconv_load_b<G, C>(p, nx, j, lane, bfr[j % C::NC]);
    }
  }
  // flush the last pair
  *reinterpret_cast<f32x4acc*>(pl + pb0) = po0 + pa0;
  if (NTL % 2 == 0) *reinterpret_cast<f32x4acc*>(pl + pb1) = po1 + pa1;
  __builtin_amdgcn_wave_barrier();
}

// Drain the finished tile: bias / activation / mask, stores.  Wave q owns channel quad q (channels 4q..4q+3 of the
// tile's 16).  A work item is FOUR consecutive pixels f0..f0+3 (f0 % 4 == 0) of an image's FLAT output plane (rows
// are contiguous, so a quad may run over a row end; only a plane's last quad can be ragged): four 16-byte LDS reads
// (one per pixel, each from the pixel's own class plane), a 4 x 4 register transpose, then ONE 16-byte store (and
// one 16-byte load of the ReLU operand) per channel.  With one pixel per item the drain issued four dword stores +
// four dword loads per pixel and was 26-27 % of the encoder's data gradients (ablation, round 3): the store issue,
// not the bytes, set its pace.  Items are issued in batches (all LDS reads and mask loads before the first store).
template <class G, class C>
__device__ __forceinline__ void uconv_drain(const UScatArgs& p, float* planes, int grp, int img0, int q, int lane) {
  constexpr int GI = C::GI, NXM = C::NXM, PLANE = C::PLANE;
  constexpr int PB = G::PB, WB = G::WB;
  constexpr int NQ = (PB + 3) / 4;       // pixel quads per image plane (the last may be ragged)
  constexpr int NI = GI * NQ;            // items of this wave
  constexpr int UB = 4;
  const int cb = grp * 16 + 4 * q;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = p.bias[cb + i];
  }
  const size_t out_elems = (size_t)p.nimg * G::CB * PB;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out, (unsigned)(out_elems * 4));
  const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux ? p.aux : p.out, (unsigned)(out_elems * 4));
  for (int v0 = lane; v0 < NI; v0 += 64 * UB) {
    f32x4acc w[UB][4], msk[UB][4];
    unsigned gofs[UB];
    int nv[UB];  // valid pixels of the quad (0: no item)
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int v = v0 + u * 64;
      const int il = v / NQ, f0 = 4 * (v % NQ);
      const bool act = v < NI && img0 + il < p.nimg;
      nv[u] = act ? min(4, PB - f0) : 0;
      gofs[u] = (unsigned)(((img0 + il) * G::CB + cb) * PB + f0);
      const f32x4acc* ibase = reinterpret_cast<const f32x4acc*>(planes) + (il * 16 + q) * PLANE;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = f0 + e, y = f / WB, x = f % WB;
        w[u][e] = f32x4acc{0.f, 0.f, 0.f, 0.f};
        if (e < nv[u]) w[u][e] = ibase[(((y & 1) << 1) | (x & 1)) * 4 * PLANE + (y >> 1) * NXM + (x >> 1)];
      }
      if (p.epi == REPO_EPI_MUL_CMASK) {
        // channel-quad mask: one byte per pixel, bit i = channel cb + i; the quad's four bytes are one aligned dword
        // (PB % 4 == 0 here); a ragged quad reads its bytes one by one
        const unsigned char* cm = reinterpret_cast<const unsigned char*>(p.aux) +
                                  ((size_t)(img0 + il) * (G::CB / 4) + (cb >> 2)) * PB + f0;
        unsigned word = 0;
        if (nv[u] == 4) word = *reinterpret_cast<const unsigned*>(cm);
        else
          for (int e = 0; e < 3; ++e)
            if (e < nv[u]) word |= (unsigned)cm[e] << (8 * e);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) msk[u][i][e] = (word >> (8 * e + i)) & 1u ? 1.f : 0.f;
      }
      if (p.epi == REPO_EPI_FILM_RELU && nv[u] > 0) {   // the item's image and four channels: (scale, shift) in msk[.][0..1]
        const float* tb = p.aux + (size_t)(2 * (img0 + il)) * G::CB + cb;
#pragma unroll
        for (int i = 0; i < 4; ++i) msk[u][i] = f32x4acc{tb[i], tb[G::CB + i], 0.f, 0.f};
      }
      if (p.epi == REPO_EPI_MUL_DRELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          msk[u][i] = f32x4acc{1.f, 1.f, 1.f, 1.f};
          if (nv[u] == 4) {
            msk[u][i] = __builtin_bit_cast(f32x4acc, __builtin_amdgcn_raw_buffer_load_b128(raux, 4u * (gofs[u] + (unsigned)i * PB), 0, 0));
          } else {
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (e < nv[u]) msk[u][i][e] = p.aux[(size_t)gofs[u] + (size_t)i * PB + e];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (nv[u] == 0) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4acc t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = w[u][e][i] + bv[i];
          if (p.epi == REPO_EPI_RELU) x = fmaxf(x, 0.f);
          else if (p.epi == REPO_EPI_MUL_DRELU || p.epi == REPO_EPI_MUL_CMASK) x = msk[u][i][e] > 0.f ? x : 0.f;
          else if (p.epi == REPO_EPI_FILM_RELU) x = fmaxf(fmaf(msk[u][i][0], x, msk[u][i][1]), 0.f);   // (scale, shift)
          t[e] = x;
        }
        const unsigned o = gofs[u] + (unsigned)i * PB;
        if (nv[u] == 4) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, t), rout, 4u * o, 0, 0);
        } else {
          // a plane's ragged last quad (copies pinned in registers of their own before the branches, cf. rowtile.h)
          float e0 = t[0], e1 = t[1], e2 = t[2];
          asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2));
          if (nv[u] > 0) p.out[o] = e0;
          if (nv[u] > 1) p.out[o + 1] = e1;
          if (nv[u] > 2) p.out[o + 2] = e2;
        }
      }
    }
  }
}

// The same drain one pixel per item (four dword stores): for planes with PB % 4 != 0, where three of the four channel
// planes of a quad start off a 16-byte boundary and the misaligned 16-byte accesses cost more than they save (enc2's
// 961-pixel planes: 421 -> 429 us; aligned planes: enc3 273 -> 241, dec3 560 -> 530).  Wave q owns channel quad q (channels 4q..4q+3 of the tile's 16): a work item is one output pixel --
// ONE 16-byte LDS read (+ the zero written back) and four dword stores, one per channel plane; the 64 lanes hold
// 64 consecutive pixels, so every store instruction writes 256 contiguous bytes and the wave's four biases are
// uniform.  Items are issued in batches (all LDS reads and mask loads of a batch before its first store): one at a
// time this phase is a chain of dependent latencies.
template <class G, class C>
__device__ __forceinline__ void uconv_drain1(const UScatArgs& p, float* planes, int grp, int img0, int q, int lane) {
  constexpr int GI = C::GI, NXM = C::NXM, PLANE = C::PLANE;
  constexpr int PB = G::PB, WB = G::WB;
  constexpr int NI = GI * PB;  // items of this wave: (image, output pixel)
  constexpr int UB = 16;
  const int cb = grp * 16 + 4 * q;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = p.bias[cb + i];
  }
  for (int v0 = lane; v0 < NI; v0 += 64 * UB) {
    f32x4acc val[UB], msk[UB];
    size_t gofs[UB];
    bool act[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int v = v0 + u * 64;
      const int il = v / PB, f = v % PB;  // image, output pixel
      const bool inb = v < NI;
      act[u] = inb && img0 + il < p.nimg;
      gofs[u] = ((size_t)(img0 + il) * G::CB + cb) * PB + f;
      const int y = f / WB, x = f % WB;
      const int cl2 = ((y & 1) << 1) | (x & 1);
      f32x4acc* w = reinterpret_cast<f32x4acc*>(planes) + ((il * 16 + cl2 * 4 + q) * PLANE + (y >> 1) * NXM + (x >> 1));
      val[u] = f32x4acc{0.f, 0.f, 0.f, 0.f};
      if (inb) val[u] = *w;
      if (p.epi == REPO_EPI_MUL_DRELU && act[u]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) msk[u][i] = p.aux[gofs[u] + (size_t)i * PB];
      }
      if (p.epi == REPO_EPI_FILM_RELU && act[u]) {   // scales in msk, shifts folded below
        const float* tb = p.aux + (size_t)(2 * (img0 + il)) * G::CB + cb;
#pragma unroll
        for (int i = 0; i < 4; ++i) msk[u][i] = tb[i];
      }
      if (p.epi == REPO_EPI_MUL_CMASK && act[u]) {  // one byte: the item's four channels (64 lanes: 64 contiguous bytes)
        const unsigned b = reinterpret_cast<const unsigned char*>(p.aux)[((size_t)(img0 + il) * (G::CB / 4) + (cb >> 2)) * PB + f];
#pragma unroll
        for (int i = 0; i < 4; ++i) msk[u][i] = (b >> i) & 1u ? 1.f : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (!act[u]) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = val[u][i] + bv[i];
        if (p.epi == REPO_EPI_RELU) t = fmaxf(t, 0.f);
        else if (p.epi == REPO_EPI_MUL_DRELU || p.epi == REPO_EPI_MUL_CMASK) t = msk[u][i] > 0.f ? t : 0.f;
        else if (p.epi == REPO_EPI_FILM_RELU)   // the shift: same image as the scale loaded above
          t = fmaxf(fmaf(msk[u][i], t, p.aux[(size_t)(2 * (gofs[u] / ((size_t)G::CB * PB)) + 1) * G::CB + cb + i]), 0.f);
        p.out[gofs[u] + (size_t)i * PB] = t;
      }
    }
  }
}

// One workgroup (4 waves = the 4 output parity classes) per tile = (image group, 16-channel group).  The workgroup
// zeroes its class planes, every wave runs its class over the tile's pixels, then wave q drains channel quad q.
// Nothing is overlapped INSIDE a workgroup on purpose: the update runs this kernel beside the weight-gradient
// kernels of the same layer (side stream) and the other update lane, and with <= 67 KB of LDS and <= 256 VGPRs two
// of these workgroups -- or one and the other kernels' -- share a CU, so one workgroup's load / drain phases are
// filled by its neighbours' MFMAs.  (A persistent variant -- compute waves + drain waves, double-buffered planes,
// one workgroup per CU -- was 8-20 % faster alone and made the whole update 6 % SLOWER: it owns every CU's LDS for
// its whole run and serialises against everything else)
template <class G, class C>
__global__ __launch_bounds__(256, 2) void uconv_scatter_kernel(UScatArgs p) {
  constexpr int GI = C::GI, NC = C::NC, NT = C::NT;
  constexpr int NFULL = NT / NC, NTAIL = NT % NC;
  constexpr int FIRST_NTL = NFULL > 0 ? NC : NTAIL;
  extern __shared__ __attribute__((aligned(16))) float planes[];  // [GI * IMG_LDS] + scratch strip

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int cls = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = output parity class (py, px), later channel quad

  // tiles: channel group fastest; dispatch slots are dealt so that an XCD works on a contiguous range of tiles
  // (the channel groups of an image group read the same input from one L2)
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int grp = tile % C::NGRP, img0 = (tile / C::NGRP) * GI;
  auto chunk_of = [&](int tile0, bool valid) { return UChunk{img0, grp, tile0, valid}; };

  // first chunk's operands are requested before the planes are zeroed
  float bfr[NC][C::KST], afr[C::NB][C::KST];
  {
    const UChunk d0 = chunk_of(0, true);
#pragma unroll
    for (int j = 0; j < FIRST_NTL; ++j) uconv_load_b<G, C>(p, d0, j, lane, bfr[j]);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wp, p.wp_bytes);
#pragma unroll
    for (int u = 0; u < C::KST / 4; ++u) {
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rw, 16u * (unsigned)lane + 1024u * (u % 4),
          4u * (unsigned)((((grp * 4 + cls) * C::J * C::J) * C::KST) * 64) + 4096u * (u / 4), 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) afr[0][4 * u + e] = v[e];
    }
  }
  for (int i = tid; i < C::LDS_TOTAL_FLOATS / 4; i += 256) reinterpret_cast<f32x4*>(planes)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  {
    char* buf = reinterpret_cast<char*>(planes);
    constexpr int dmy = 4 * C::LDS_FLOATS;  // the scratch strip (bytes)
    const UChunk none = chunk_of(0, false);
#pragma unroll
    for (int ch = 0; ch < NFULL; ++ch) {
      const UChunk d = chunk_of(ch * NC, true);
      if (ch + 1 < NFULL) uconv_chunk<G, C, NC, NC>(p, buf, cls, lane, d, chunk_of((ch + 1) * NC, true), dmy, bfr, afr);
      else if (NTAIL > 0) uconv_chunk<G, C, NC, (NTAIL > 0 ? NTAIL : 1)>(p, buf, cls, lane, d, chunk_of(NFULL * NC, true), dmy, bfr, afr);
      else uconv_chunk<G, C, NC, 1>(p, buf, cls, lane, d, none, dmy, bfr, afr);
    }
    if (NTAIL > 0) uconv_chunk<G, C, (NTAIL > 0 ? NTAIL : 1), 1>(p, buf, cls, lane, chunk_of(NFULL * NC, true), none, dmy, bfr, afr);
  }
  __syncthreads();
  if (G::PB % 4 == 0) uconv_drain<G, C>(p, planes, grp, img0, cls, lane);
  else uconv_drain1<G, C>(p, planes, grp, img0, cls, lane);
}

template <class G, class C>
inline int launch_uconv_pack(const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < C::PACK_FLOATS * sizeof(float)) return REPO_E_WS_TOO_SMALL;
  UPackArgs pa{w, (float*)ws};
  hipLaunchKernelGGL((uconv_pack_kernel<G>), dim3((unsigned)((C::PACK_FLOATS + 1023) / 1024)), dim3(256), 0, s, pa);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// `packed` != 0: ws already holds this layer's pack (repo_conv_up_pack); else it is written first.
template <class G, class C>
inline int launch_uconv_scatter(const float* small, const float* w, const float* bias, const float* aux, float* out,
                                int64_t nimg, int epi, int packed, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < C::PACK_FLOATS * sizeof(float)) return REPO_E_WS_TOO_SMALL;
  float* wp = (float*)ws;
  if (!packed) {
    const int rc = launch_uconv_pack<G, C>(w, ws, ws_bytes, s);
    if (rc) return rc;
  }
  const int ngi = (int)((nimg + C::GI - 1) / C::GI);
  UScatArgs a{small, wp, bias, aux, out, (int)nimg, epi, (unsigned)(nimg * G::CS * G::PS * sizeof(float)),
              (unsigned)(C::PACK_FLOATS * sizeof(float))};
  constexpr int lds_b = C::LDS_TOTAL_FLOATS * (int)sizeof(float);
  static_assert(lds_b <= 80 * 1024, "two workgroups per CU");
  hipError_t e = hipFuncSetAttribute((const void*)uconv_scatter_kernel<G, C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     lds_b);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((uconv_scatter_kernel<G, C>), dim3((unsigned)(ngi * C::NGRP)), dim3(256), lds_b, s, a);
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
