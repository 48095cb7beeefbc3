// Scatter-form transposed convolution (uconv.h) at fp32 accuracy on the bf16 matrix pipe ("bf16x6", bgemm.h).
//
//   big[img][cb][2iy+ky][2ix+kx] += sum_cs small[img][cs][iy][ix] * w[cs][cb][ky][kx]
//
// uconv.h's kernel, instruction for instruction -- input-stationary per-tap GEMMs over the REAL input pixels, one
// output parity class per wave, class planes in LDS updated with plain read-add-write pipelined into the MFMA chains,
// the same drains -- with the per-tap product formed by v_mfma_f32_16x16x32_bf16 on operands split into three bf16
// planes (a = a1 + a2 + a3 exactly; the six cross products with i + j <= 4):
//   * the input pixels of a tile stay in REGISTERS for all taps as before, now as [k-block of 32 channels][plane]
//     fragments of 8 bf16 (lane (pixel lp, octet lq) holds channels 32 kb + 8 lq .. + 7): loaded as fp32 straight from
//     global memory (16 dwords per 16 pixels, as before) and split once per chunk -- 88 vector instructions per tile
//     against 108 MFMAs per tile and chunk on the 3 x 3-tap layers;
//   * the weights of (16 output channels, class, tap) arrive pre-split from a fragment-ready pack: 3 planes x
//     (CS / 32) coalesced 16-byte loads per lane, one tap ahead, two rotating buffers;
//   * a tap of a tile is a chain of 6 * CS / 32 MFMAs of 16 cycles (fp32: CS / 4 of 32 cycles): 12 against 16 slots
//     at CS = 64 -- the write-back of the previous pair and the request of the own old values sit in slots 1-2 / 4-5 as
//     before.
// The fragments of a tile are kept for 64 input channels at a time (more do not fit the register file beside the
// weights): a layer with CS = 128 / 256 walks its channels in K-SLICES of 64 -- the accumulation is in the LDS class
// planes anyway, so a slice is just another pass of chunks over the same tiles with the next slice's weights.
// Odd kernels (k5): the tap set depends on the output parity class, so the tap loop is instantiated per class (the
// wave's class is uniform; even kernels share one instantiation).
#pragma once
#include "bgemm.h"
#include "uconv.h"

namespace repo {

template <class G, int GI_, int NC_>
struct BSConf {
  static constexpr int GI = GI_;                 // images per workgroup
  static constexpr int NC = NC_;                 // N tiles whose B fragments are resident at a time (even)
  static constexpr int KSL = G::CS / 64;         // K-slices of 64 input channels
  static constexpr int KB = 2;                   // k-blocks of v_mfma_f32_16x16x32_bf16 per slice
  static constexpr int KST = 16;                 // dword loads of a B tile per lane and slice (= 8 KB)
  static constexpr int NGRP = G::CB / 16;
  static constexpr int NPX = GI * G::PS;
  static constexpr int NT = (NPX + 15) / 16;
  static constexpr int J = (G::KS + 1) / 2;
  static constexpr int NYM = (G::HB + 1) / 2, NXM = (G::WB + 1) / 2;
  static constexpr int PLANE = NYM * NXM;
  static constexpr int IMG_LDS = 4 * 4 * PLANE * 4;
  static constexpr int LDS_FLOATS = GI * IMG_LDS;
  static constexpr int DUMMY_FLOATS = 4 * (64 + (J - 1) * (NXM + 1) + 4);
  static constexpr int LDS_TOTAL_FLOATS = LDS_FLOATS + DUMMY_FLOATS;
  // pack: [grp][cls][slice][tap][kb][plane][lane][8 bf16]
  static constexpr int TAP_BYTES = KB * 3 * 64 * 16;
  static constexpr size_t PACK_BYTES = (size_t)NGRP * 4 * KSL * J * J * TAP_BYTES;
  static constexpr size_t PACK_FLOATS = PACK_BYTES / 4;   // (the host code sizes workspaces in floats)
  static_assert(G::CS % 64 == 0 && G::CB % 16 == 0 && NC % 2 == 0, "64-channel slices, 16-channel groups, tile pairs");
  static_assert(6 * KB >= 8, "the pipelined write-back needs 8 slots per chain");
};

struct BUPackArgs {
  const float* w;
  char* wp;
};
// one thread per (grp, cls, slice, tap, kb, lane): the 8 k of one A fragment, three planes
template <class G, class C>
__global__ __launch_bounds__(256) void buconv_pack_kernel(BUPackArgs p) {
  constexpr int J = C::J, KB = C::KB;
  const int total = C::NGRP * 4 * C::KSL * J * J * KB * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63;
    int r = i >> 6;
    const int kb = r % KB;
    r /= KB;
    const int tap = r % (J * J);
    r /= J * J;
    const int ks = r % C::KSL;
    r /= C::KSL;
    const int cls = r & 3, grp = r >> 2;
    const int ky = (cls >> 1) + 2 * (tap / J), kx = (cls & 1) + 2 * (tap % J);
    const int cb = 16 * grp + (lane & 15);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cs = 64 * ks + 32 * kb + 8 * (lane >> 4) + j;
      v[j] = (ky < G::KS && kx < G::KS) ? p.w[((size_t)cs * G::CB + cb) * G::KK + ky * G::KS + kx] : 0.f;
    }
    unsigned pl[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bg_split3(v[2 * e], v[2 * e + 1], pl[0][e], pl[1][e], pl[2][e]);
    char* dst = p.wp + ((size_t)(((grp * 4 + cls) * C::KSL + ks) * J * J + tap) * KB + kb) * (3 * 64 * 16) + lane * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<u32x4s*>(dst + q * (64 * 16)) = u32x4s{pl[q][0], pl[q][1], pl[q][2], pl[q][3]};
  }
}

// A chunk = up to NC consecutive N tiles of one workgroup tile, over ONE K-slice; wave-uniform.
struct BUChunk {
  int img0, grp, tile0, ks;
  bool valid;
};

// raw fp32 values of N tile `j` of chunk d: lane (pixel lp, octet lq) loads small[img][64 ks + 32 kb + 8 lq + e][pixel]
template <class G, class C>
__device__ __forceinline__ void buconv_load_raw(const UScatArgs& p, const BUChunk& d, int j, int lane, float (&raw)[C::KST]) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.small, p.small_bytes);
  const int lp = lane & 15, lq = lane >> 4;
  const int q = (d.tile0 + j) * 16 + lp;
  const int il = q / G::PS, pix = q % G::PS;
  const bool ok = q < C::NPX && d.img0 + il < p.nimg;
  const unsigned base = ok ? 4u * (unsigned)(((d.img0 + il) * G::CS + 64 * d.ks + 8 * lq) * G::PS + pix) : kOobOffset;
#pragma unroll
  for (int kb = 0; kb < C::KB; ++kb)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      constexpr unsigned CH = 4u * G::PS;                 // bytes between channels
      const unsigned c = CH * (unsigned)(32 * kb + e);    // compile time
      raw[8 * kb + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, base + c % 4096u, c / 4096u * 4096u, 0));
    }
}

typedef bg_bf16x8 BFrag[3];  // the three planes of one 8-k fragment

template <class C>
__device__ __forceinline__ void buconv_split(const float (&raw)[C::KST], BFrag (&f)[C::KB]) {
#ifdef BU_NO_SPLIT   // ablation build (profiles/r05_split_ablation.txt): this kernel's per-chunk split removed
#pragma unroll
  for (int kb = 0; kb < C::KB; ++kb)
#pragma unroll
    for (int q = 0; q < 3; ++q)
      f[kb][q] = __builtin_bit_cast(bg_bf16x8, u32x4s{__builtin_bit_cast(unsigned, raw[8 * kb + q]), __builtin_bit_cast(unsigned, raw[8 * kb + q + 1]),
                                                      __builtin_bit_cast(unsigned, raw[8 * kb + q + 2]), __builtin_bit_cast(unsigned, raw[8 * kb + q + 3])});
  return;
#endif
#pragma unroll
  for (int kb = 0; kb < C::KB; ++kb) {
    unsigned pl[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bg_split3(raw[8 * kb + 2 * e], raw[8 * kb + 2 * e + 1], pl[0][e], pl[1][e], pl[2][e]);
#pragma unroll
    for (int q = 0; q < 3; ++q) f[kb][q] = __builtin_bit_cast(bg_bf16x8, u32x4s{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
  }
}

// All taps of one parity class over one chunk of NTL (compile-time) N tiles, for one compute wave: uconv_chunk with
// the chain of a (tile, tap) being 6 KB MFMAs.  PY, PX: the class whose TAP SET this instantiation walks (ty < JY, tx <
// JX; for even kernels every class has all J x J taps and (0, 0) serves them all -- the class itself, `cls`, only enters
// addresses).  A0 = the weight buffer the chunk's FIRST tap sits in: the taps alternate between the two buffers, the
// next chunk (same tiles' next K-slice, or the next tiles) starts where this one ends.
template <class G, class C, int PY, int PX, int NTL, int NEXT_NTL, int A0>
__device__ __forceinline__ void buconv_chunk(const UScatArgs& p, char* pl, int cls, int lane, const BUChunk& d,
                                             const BUChunk& nx, int dummy_ofs, BFrag (&bfr)[C::NC][C::KB],
                                             float (&raw)[C::NC][C::KST], BFrag (&afr)[2][C::KB]) {
  constexpr int KB = C::KB, J = C::J, NXM = C::NXM, PLANE = C::PLANE;
  constexpr int JY = (G::KS - PY + 1) / 2, JX = (G::KS - PX + 1) / 2, NTAP = JY * JX;
  constexpr int NP = (NTL + 1) / 2;
  constexpr int SLOTS = 6 * KB;
  const int lp = lane & 15, lq = lane >> 4;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wp, p.wp_bytes);
  const unsigned a_lane = 16u * (unsigned)lane;
  auto a_sbase = [&](const BUChunk& c) { return (unsigned)(((c.grp * 4 + cls) * C::KSL + c.ks) * J * J) * (unsigned)C::TAP_BYTES; };
  // tn = tap number in the pack (ty * J + tx)
  auto load_a = [&](BFrag (&a)[KB], unsigned sbase, int tn) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        a[kb][q] = __builtin_bit_cast(bg_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
            rw, a_lane + 1024u * (unsigned)q, sbase + (unsigned)tn * C::TAP_BYTES + 3072u * (unsigned)kb, 0));
  };

  int lbase[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int q = (d.tile0 + j) * 16 + lp;
    const int il = q / G::PS, pix = q % G::PS;
    const bool ok = q < C::NPX && d.img0 + il < p.nimg;
    const int iy = pix / G::WS, ix = pix % G::WS;
    lbase[j] = ok ? 16 * ((il * 16 + cls * 4 + lq) * PLANE + iy * NXM + ix) : dummy_ofs + 16 * lane;
  }
  // the chunk's tiles arrived as fp32 (requested during the previous chunk's last tap): split them
#pragma unroll
  for (int j = 0; j < NTL; ++j) buconv_split<C>(raw[j], bfr[j]);

  f32x4acc pa0 = {0.f, 0.f, 0.f, 0.f}, pa1 = pa0, po0 = pa0, po1 = pa0;
  int pb0 = 0, pb1 = 0;

#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int ty = t / JX, tx = t % JX;
    const bool last = t == NTAP - 1;
    const int ab = (A0 + t) & 1;  // compile time after unrolling
    if (!last) load_a(afr[ab ^ 1], a_sbase(d), ((t + 1) / JX) * J + (t + 1) % JX);
    else if (nx.valid) load_a(afr[ab ^ 1], a_sbase(nx), 0);
    const int shift = 16 * (ty * NXM + tx);
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int j0 = 2 * pr, j1 = (2 * pr + 1 < NTL) ? 2 * pr + 1 : 2 * pr;
      const bool two = 2 * pr + 1 < NTL;
      const bool have_prev = !(t == 0 && pr == 0);
      const bool ptwo = pr > 0 ? true : (NTL % 2 == 0);
      const int cb0 = lbase[j0] + shift, cb1 = lbase[j1] + shift;
      f32x4acc ca0 = {0.f, 0.f, 0.f, 0.f}, ca1 = ca0;
      f32x4acc co0 = ca0, co1 = ca0;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        // slot s = (kb, term): smallest terms first inside a k-block
        constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
        const int kb = s / 6, ta = TA[s % 6], tb = TB[s % 6];
        ca0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ab][kb][ta], bfr[j0][kb][tb], ca0, 0, 0, 0);
        if (two) ca1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ab][kb][ta], bfr[j1][kb][tb], ca1, 0, 0, 0);
        if (have_prev) {
          if (s == 1) *reinterpret_cast<f32x4acc*>(pl + pb0) = po0 + pa0;
          if (s == 2 && ptwo) *reinterpret_cast<f32x4acc*>(pl + pb1) = po1 + pa1;
        }
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
      if (last && nx.valid) {  // this pair's fragments are dead: request the next chunk's tiles (as fp32)
        if (2 * pr < NEXT_NTL) buconv_load_raw<G, C>(p, nx, 2 * pr, lane, raw[2 * pr]);
        if (2 * pr + 1 < NEXT_NTL) buconv_load_raw<G, C>(p, nx, 2 * pr + 1, lane, raw[(2 * pr + 1) % C::NC]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (last && nx.valid) {
#pragma unroll
      for (int j = 2 * NP; j < NEXT_NTL; ++j) buconv_load_raw<G, C>(p, nx, j, lane, raw[j % C::NC]);
    }
  }
  *reinterpret_cast<f32x4acc*>(pl + pb0) = po0 + pa0;
  if (NTL % 2 == 0) *reinterpret_cast<f32x4acc*>(pl + pb1) = po1 + pa1;
  __builtin_amdgcn_wave_barrier();
}

// The chunks of one workgroup tile in order: for each K-slice, the tile chunks (NFULL of NC tiles + a tail).  Chunk c
// is instantiated with its own tile count, its successor's, and the weight-buffer parity it starts in.
template <class G, class C, int PY, int PX, int CI>
__device__ __forceinline__ void buconv_run(const UScatArgs& p, char* buf, int cls, int lane, int img0, int grp,
                                           BFrag (&bfr)[C::NC][C::KB], float (&raw)[C::NC][C::KST], BFrag (&afr)[2][C::KB]) {
  constexpr int NC = C::NC, NT = C::NT;
  constexpr int NFULL = NT / NC, NTAIL = NT % NC, NCHT = NFULL + (NTAIL > 0 ? 1 : 0), NCH = C::KSL * NCHT;
  constexpr int NTAP = ((G::KS - PY + 1) / 2) * ((G::KS - PX + 1) / 2);
  if constexpr (CI < NCH) {
    constexpr int ct = CI % NCHT, ks = CI / NCHT;
    constexpr int NTL = ct < NFULL ? NC : NTAIL;
    constexpr bool more = CI + 1 < NCH;
    constexpr int nct = (CI + 1) % NCHT, nks = (CI + 1) / NCHT;
    constexpr int NEXT = more ? (nct < NFULL ? NC : NTAIL) : 1;
    const BUChunk d{img0, grp, ct * NC, ks, true};
    const BUChunk nx{img0, grp, nct * NC, nks, more};
    buconv_chunk<G, C, PY, PX, NTL, NEXT, (CI * NTAP) & 1>(p, buf, cls, lane, d, nx, 4 * C::LDS_FLOATS, bfr, raw, afr);
    buconv_run<G, C, PY, PX, CI + 1>(p, buf, cls, lane, img0, grp, bfr, raw, afr);
  }
}

template <class G, class C>
__global__ __launch_bounds__(256, 2) void buconv_scatter_kernel(UScatArgs p) {
  constexpr int GI = C::GI, NC = C::NC, NT = C::NT, JJ = C::J * C::J;
  constexpr int FIRST_NTL = NT / NC > 0 ? NC : NT % NC;
  extern __shared__ __attribute__((aligned(16))) float planes[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int cls = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int grp = tile % C::NGRP, img0 = (tile / C::NGRP) * GI;

  BFrag bfr[NC][C::KB], afr[2][C::KB];
  float raw[NC][C::KST];
  {
    const BUChunk d0{img0, grp, 0, 0, true};
#pragma unroll
    for (int j = 0; j < FIRST_NTL; ++j) buconv_load_raw<G, C>(p, d0, j, lane, raw[j]);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wp, p.wp_bytes);
#pragma unroll
    for (int kb = 0; kb < C::KB; ++kb)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        afr[0][kb][q] = __builtin_bit_cast(bg_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
            rw, 16u * (unsigned)lane + 1024u * (unsigned)q,
            (unsigned)(((grp * 4 + cls) * C::KSL) * JJ) * (unsigned)C::TAP_BYTES + 3072u * (unsigned)kb, 0));
  }
  for (int i = tid; i < C::LDS_TOTAL_FLOATS / 4; i += 256) reinterpret_cast<f32x4*>(planes)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  char* buf = reinterpret_cast<char*>(planes);
  if constexpr (G::KS % 2 == 0) {
    buconv_run<G, C, 0, 0, 0>(p, buf, cls, lane, img0, grp, bfr, raw, afr);
  } else {  // odd kernels: the class's tap set is a compile-time property of its own instantiation
    if (cls == 0) buconv_run<G, C, 0, 0, 0>(p, buf, cls, lane, img0, grp, bfr, raw, afr);
    else if (cls == 1) buconv_run<G, C, 0, 1, 0>(p, buf, cls, lane, img0, grp, bfr, raw, afr);
    else if (cls == 2) buconv_run<G, C, 1, 0, 0>(p, buf, cls, lane, img0, grp, bfr, raw, afr);
    else buconv_run<G, C, 1, 1, 0>(p, buf, cls, lane, img0, grp, bfr, raw, afr);
  }
  __syncthreads();
  if (G::PB % 4 == 0) uconv_drain<G, C>(p, planes, grp, img0, cls, lane);
  else uconv_drain1<G, C>(p, planes, grp, img0, cls, lane);
}

template <class G, class C>
inline int launch_buconv_pack(const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < C::PACK_BYTES) return REPO_E_WS_TOO_SMALL;
  const int total = C::NGRP * 4 * C::KSL * C::J * C::J * C::KB * 64;
  hipLaunchKernelGGL((buconv_pack_kernel<G, C>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, BUPackArgs{w, (char*)ws});
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

template <class G, class C>
inline int launch_buconv_scatter(const float* small, const float* w, const float* bias, const float* aux, float* out,
                                 int64_t nimg, int epi, int packed, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < C::PACK_BYTES) return REPO_E_WS_TOO_SMALL;
  if (!packed) {
    const int rc = launch_buconv_pack<G, C>(w, ws, ws_bytes, s);
    if (rc) return rc;
  }
  const int ngi = (int)((nimg + C::GI - 1) / C::GI);
  UScatArgs a{small, (const float*)ws, bias, aux, out, (int)nimg, epi, (unsigned)(nimg * G::CS * G::PS * sizeof(float)),
              (unsigned)C::PACK_BYTES};
  constexpr int lds_b = C::LDS_TOTAL_FLOATS * (int)sizeof(float);
  static_assert(lds_b <= 80 * 1024, "two workgroups per CU");
  hipError_t e = hipFuncSetAttribute((const void*)buconv_scatter_kernel<G, C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((buconv_scatter_kernel<G, C>), dim3((unsigned)(ngi * C::NGRP)), dim3(256), lds_b, s, a);
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
