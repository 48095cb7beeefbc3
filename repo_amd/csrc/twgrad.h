// Conv weight gradient on the bf16 matrix pipe ("bf16x6", bgemm.h) with BOTH operands split ONCE, at staging, and the
// stride-2 gather of `big` done by the LDS's transposing read (ds_read_b64_tr_b16, gfx950).
//
//   dw[cs][cb][ky][kx] = sum_{img,sy,sx} small[img][cs][sy][sx] * big[img][cb][2sy+ky][2sx+kx]
//   per tap (ky,kx): M = cs (A = small), N = cb (B = big), K = (img, sy, sx)
//
// bwgrad.h keeps `big` as an fp32 LDS image and splits every B fragment in registers at every use (44 dependent vector
// instructions per 8 elements, again for every tap that meets the element): the matrix pipe is busy 0.38 of the time.
// Here `big` is staged CHANNEL-INNERMOST -- three bf16 planes [row][column parity][16-channel block][x/2][16 channels],
// 32 B per pixel and block -- and a B fragment (column = channel, 8 k = 8 output pixels) is two ds_read_b64_tr_b16: each
// 16-lane group names four PIXEL addresses (any four: the stride-2 walk, a row wrap, costs nothing) and receives, per
// lane = channel, the four pixels' values.  Consecutive output pixels of a tap are consecutive entries of one parity
// plane; the k of a 32-block are dealt to the lane groups so that the two groups of a half-wave read EIGHT consecutive
// pixels (256 B: every bank once; `small`'s planes store their k in the same order).  A tap is an IMMEDIATE offset: the
// taps of a parity class (ky & 1, kx & 1) sit at a * (row bytes) + c * 32 B from the class's first, a, c < KS / 2.
//   * a PAIR of workgroups per split, one per ROW parity ky & 1 of the taps: a workgroup stages only the rows of `big` of that parity (no row is
//     staged twice by the pair; `small` is) and owns those taps of dw for its images, all cs x cb;
//   * 8 (or 12) waves, SPECIALISED: waves 0-3 multiply -- wave = (32 cs rows) x (32 channels) x (the (KS/2)^2 taps of one column
//     parity): 9 x 16 accumulator registers for k6, `small` (A, k-contiguous) in planes [cs][k] as in bwgrad.h (one
//     aligned ds_read_b128 per plane and fragment); one (block, tap) step's six MFMAs run while the next step's six
//     transposing reads are in flight -- waves 4-7 (4-11 where the multiplying wave fits 168 registers) stage: global -> registers -> split -> the OTHER of two LDS buffers,
//     their loads one chunk further ahead.  Every SIMD holds one wave of each kind: the split's vector instructions issue
//     in the shadow of the MFMAs instead of in a phase of their own (this kernel's first version, all 8 waves doing
//     both in turn: MFMA pipe busy 0.59, a quarter of the time in the staging phase).  One LDS-only barrier per chunk;
//   * K is the FLAT pixel index of an image, 16 at a time, in chunks of NBK blocks; a chunk stages the rows of `big` its
//     pixels touch; phantom k (>= PS, last block) have A = 0 and a clamped B address;
//   * slab[z][tap][cs][cb] (the lane = cb runs are 128 B), reduced in fixed order by conv_slab_reduce_wave_kernel's
//     tap-major mode.
// Reference: autograd's weight gradient of nn.Conv2d / nn.ConvTranspose2d (models/encoder.py:35-38, decoder.py:43-47).
#pragma once
#include "bgemm.h"
#include "dconv.h"
#include "rowtile.h"

namespace repo {

typedef short tw_s16x4 __attribute__((ext_vector_type(4)));
#define TW_LDS(p) ((__attribute__((address_space(3))) tw_s16x4*)(p))

// bg_split3 with the residuals as single v_sub_f32: left to itself the compiler packs the two lanes' subtractions into
// v_pk_add_f32, which beside another wave's MFMAs costs an order of magnitude more than its issue slot
// (MI355X_MICROARCH.md, per-instruction constants)
__device__ __forceinline__ float tw_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void tw_split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(bg_f32x2{x0, x1}, bg_bf16x2));
  const float r0 = tw_sub(x0, __builtin_bit_cast(float, p1 << 16)), r1 = tw_sub(x1, __builtin_bit_cast(float, p1 & 0xffff0000u));
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(bg_f32x2{r0, r1}, bg_bf16x2));
  const float s0 = tw_sub(r0, __builtin_bit_cast(float, p2 << 16)), s1 = tw_sub(r1, __builtin_bit_cast(float, p2 & 0xffff0000u));
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(bg_f32x2{s0, s1}, bg_bf16x2));
}

template <class G, int NBK_, int NPW_ = 4>
struct TWGeo {
  static_assert(G::CB == 32 && G::CS == 64 && G::KS % 2 == 0, "twgrad: 32 -> 64 channel layers with an even kernel");
  static constexpr int NBK = NBK_, NP = 64 * NPW_, NT = 256 + NP;   // staging threads (NPW_ waves); threads
  static constexpr int KC = 32 * NBK;                  // k per chunk
  static constexpr int NBLK = (G::PS + 31) / 32;       // k-blocks (one v_mfma_f32_16x16x32_bf16 deep) per image
  static constexpr int NCH = (NBLK + NBK - 1) / NBK;   // chunks per image
  static constexpr int H2 = G::KS / 2, TPW = H2 * H2;
  static constexpr int XH = ((G::WB + 1) / 2) | 1;     // pixels per (row, column parity), odd: see the B store
  static constexpr int PIXB = 32;                      // bytes per pixel, 16-channel block and plane
  static constexpr int ROWB = 4 * XH * PIXB;           // bytes per staged row of `big` and plane: [parity][channel block][x/2][16]
  static constexpr int nblk(int j) { return cmin(NBK, NBLK - j * NBK); }
  static constexpr int sy0(int j) { return (KC * j) / G::WS; }
  static constexpr int klast(int j) { return cmin(KC * (j + 1), G::PS) - 1; }
  static constexpr int nrows(int j) { return klast(j) / G::WS - sy0(j) + H2; }   // rows of ONE parity a chunk touches
  static constexpr int nimax() {
    int m = 0;
    for (int j = 0; j < NCH; ++j) m = cmax(m, nrows(j));
    return m;
  }
  static constexpr int NIMAX = nimax();
  static constexpr int BPLANE = NIMAX * ROWB;
  static constexpr int AP = KC + 16;                   // row stride 10 x 16 B (mod 16): the 16-row x 4-octet ds_read_b128 is conflict-free
  static_assert((AP * 2 / 16) % 16 == 10, "A pitch");
  static constexpr int APLANE = G::CS * AP * 2;
  static constexpr int BUF = 3 * (BPLANE + APLANE);
  static constexpr int LDS_BYTES = 2 * BUF;
  static constexpr int AQ = KC / 4;
  static_assert((AQ & (AQ - 1)) == 0 && AQ <= 64, "the bias-gradient lanes of a row are one aligned lane group");
  static constexpr int A_NV = G::CS * AQ, A_PER = A_NV / NP;
  static_assert(A_NV % NP == 0, "every staging thread has its A items");
  static constexpr int QPR = (G::WB + 3) / 4;          // pixel quads per row (the last one ends WITH the row)
  static constexpr int B_NV = 8 * QPR * NIMAX, B_PER = (B_NV + NP - 1) / NP;
  // floats per split: [tap][cs][cb], db[cs], then the two row-parity halves of the channel sums of `big`
  static constexpr int SLAB = G::CS * (G::CB * G::KK + 1) + 2 * G::CB;
  // rows of chunk j that chunk j - 1 has not staged: every element of `big` is counted once
  static constexpr int own_from(int j) { return j == 0 ? 0 : cmax(0, sy0(j - 1) + nrows(j - 1) - sy0(j)); }
};

template <class G, int NBK, int NPW>
__global__ __launch_bounds__(256 + 64 * NPW) void tconv_wgrad_kernel(WgradArgs p) {
  using TG = TWGeo<G, NBK, NPW>;
  constexpr int NP = TG::NP;
  constexpr int KC = TG::KC, NCH = TG::NCH, XH = TG::XH, AP = TG::AP, AQ = TG::AQ;
  constexpr int BPLANE = TG::BPLANE, APLANE = TG::APLANE, H2 = TG::H2, TPW = TG::TPW;
  constexpr int A_PER = TG::A_PER, B_PER = TG::B_PER, QPR = TG::QPR;
  extern __shared__ __attribute__((aligned(16))) char tw_lds[];

  const int tid = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The two row parities of a split read the same cache lines of `big` (a row is 120-124 B) and the same `small`:
  // consecutive workgroups go to consecutive XCDs (8 L2s), so the pair is dealt 8 apart -- same XCD, same L2
  const int L = blockIdx.x;
  const int pky = (L >> 3) & 1, z = (L >> 4) * 8 + (L & 7);
  if (z >= p.nsplits_tw) return;
  const int img_beg = z * p.imgs_per_split, img_end = min(p.nimg, img_beg + p.imgs_per_split);
  float* sl = p.slab + (size_t)z * TG::SLAB;
  // lane constants are recomputed from an opaque copy of the thread index where they are used: as kernel-long values
  // they are a dozen registers the allocator would rather spill
  auto otid = [&]() __attribute__((always_inline)) {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t;
  };

  if (wid < 4) {
    // =================================================================== the multiplying waves
    __builtin_amdgcn_s_setprio(2);   // (measured neutral against priority 0: 260.9 / 354.5 vs 262.0 / 356.5 us)
    const int cblk = wid & 1, pkx = wid >> 1;   // 16 channels; column parity of the taps
    f32x4 acc[TPW][4];                          // [tap][16 cs rows]
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](auto jc, const char* Bl, const char* Al) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int NB = TG::nblk(j), s0 = TG::sy0(j), NS = NB * TPW;
      // byte offset of this lane's pixel quad r of block b; lane 4q+p of the 16-lane group g names pixel q of the quad,
      // channels 16 cblk + 4p .. + 3
      auto baddr = [&](int b, int r) __attribute__((always_inline)) {
        const int lane = otid() & 63;
        const int g = lane >> 4;
        // element e = 4 r + q of lane group g is pixel 16 (g >> 1) + 8 r + 4 (g & 1) + q of the block
        const int k = min(KC * j + 32 * b + 16 * (g >> 1) + 8 * r + 4 * (g & 1) + ((lane & 15) >> 2), G::PS - 1);
        const int sy = k / G::WS, sx = k - sy * G::WS;
        return (sy - s0) * TG::ROWB + ((pkx * 2 + cblk) * XH + sx) * TG::PIXB + 4 * (lane & 3) * 2;
      };
      auto load_b = [&](bg_bf16x8(&fb)[3], int o0, int o1, int t) __attribute__((always_inline)) {
        const int toff = (t / H2) * TG::ROWB + (t % H2) * TG::PIXB;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const tw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(TW_LDS(Bl + q * BPLANE + o0 + toff));
          const tw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(TW_LDS(Bl + q * BPLANE + o1 + toff));
          fb[q] = __builtin_bit_cast(bg_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      };
      // A fragment of row tile m: row 16 m + (lane & 15), k = 32 b + 8 (lane >> 4) .. + 7
      auto load_a = [&](bg_bf16x8(&fa)[4][3], int b) __attribute__((always_inline)) {
        const int lane = otid() & 63;
        const int abase = ((lane & 15) * AP + 8 * (lane >> 4)) * 2;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            fa[m][q] = *reinterpret_cast<const bg_bf16x8*>(Al + q * APLANE + abase + (m * 16 * AP + 32 * b) * 2);
      };
      // one step = one (block, tap): its 24 MFMAs (four row tiles x six products: a B fragment read once meets all 64 cs
      // -- with one row tile per wave the transposing reads alone kept the LDS busy 0.75 of the MFMA time) run while the
      // next step's six reads are in flight
      bg_bf16x8 fa[4][3], fb[2][3];
      int o0 = baddr(0, 0), o1 = baddr(0, 1);
      load_a(fa, 0);
      load_b(fb[0], o0, o1, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int b = s / TPW, t = s % TPW;
        if (s + 1 < NS) {
          const int b2 = (s + 1) / TPW, t2 = (s + 1) % TPW;
          if (t2 == 0) o0 = baddr(b2, 0), o1 = baddr(b2, 1);
          load_b(fb[(s + 1) & 1], o0, o1, t2);
        }
        __builtin_amdgcn_sched_barrier(0);
        // per accumulator the smallest terms first; the four row tiles interleaved: no MFMA waits for its predecessor
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#ifdef TW_NO_MFMA   // ablation builds (tools/build_variant.sh): results wrong, time meaningful
            acc[t][m][0] += __builtin_bit_cast(float, (int)fa[m][PA[pr]][0] + (int)fb[s & 1][PB[pr]][0]);
#else
            acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][PA[pr]], fb[s & 1][PB[pr]], acc[t][m], 0, 0, 0);
#endif
        if (t == TPW - 1 && b + 1 < NB) load_a(fa, b + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    lds_barrier();   // chunk 0 is staged
    int buf = 0;
    for (int img = img_beg; img < img_end; ++img) {
      auto body = [&](auto jc) __attribute__((always_inline)) {
        const char* Bl = tw_lds + buf * TG::BUF;
#ifndef TW_NO_COMPUTE
        compute(jc, Bl, Bl + 3 * BPLANE);
#endif
        lds_barrier();
        buf ^= 1;
      };
      body(std::integral_constant<int, 0>{});
      if constexpr (NCH > 1) body(std::integral_constant<int, 1>{});
      if constexpr (NCH > 2) body(std::integral_constant<int, 2>{});
      if constexpr (NCH > 3) body(std::integral_constant<int, 3>{});
      if constexpr (NCH > 4) body(std::integral_constant<int, 4>{});
      if constexpr (NCH > 5) body(std::integral_constant<int, 5>{});
      if constexpr (NCH > 6) body(std::integral_constant<int, 6>{});
      static_assert(NCH <= 7, "chunks per image");
    }
    if (p.want_dbig) lds_barrier();   // the staging waves' channel sums meet in LDS
    // ---- slab[z][tap][cs][cb]: lane & 15 = channel of the wave's 16, accumulator register r = row 4 (lane >> 4) + r of
    // the 16 of its row tile
    const int lane = tid & 63;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int ky = 2 * (t / H2) + pky, kx = 2 * (t % H2) + pkx;
      float* dst = sl + ((size_t)(ky * G::KS + kx) * G::CS + 4 * (lane >> 4)) * G::CB + 16 * cblk + (lane & 15);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * m + r) * G::CB] = acc[t][m][r];
    }
  } else {
    // =================================================================== the staging waves
    const __amdgpu_buffer_rsrc_t rsm = make_rsrc(p.small, p.small_bytes), rbg = make_rsrc(p.big, p.big_bytes);
    float dbs[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) dbs[i] = 0.f;
    float dsum[4] = {0.f, 0.f, 0.f, 0.f};   // want_dbig: this thread's channel quad (tid & 7), over the pixels it stages FIRST
    // A item i: v = ptid + 256 i -> (cs = v / AQ, quad of k = v % AQ); B item i: v -> (channel quad v % 8, pixel quad
    // (v / 8) % QPR of staged row v / (8 QPR))
    // two register sets: a chunk's loads are issued two barriers before its split
    f32x4 rav[2][A_PER], rbv[2][B_PER][4];
    auto gload = [&](auto jc, auto sc, int img) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value, S = decltype(sc)::value;
      constexpr int k0 = KC * j, s0 = TG::sy0(j), NI = TG::nrows(j);
      const unsigned dead_img = img < img_end ? 0u : ~0u;
      const int t_ = (TPW <= 4 ? tid : otid()) - 256;   // k4: registers to spare, the lane constants may be hoisted (enc2 215 -> 199 us; dec3 309 -> 350)
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int a_e4 = (t_ + NP * i) % AQ, a_cs = (t_ + NP * i) / AQ;
        const int k = k0 + 4 * a_e4;
        // the quad that holds a row's last PS % 4 pixels is loaded from PS - 4 (never past the row: the last row of
        // `small` ends the buffer) and rotated into place by the store.  Inactive items read out of range (zeros): the
        // sign of (PS - 1 - k) becomes the offset's top bit -- arithmetic, not a select the compiler turns into branches
        // around the loads (with an s_waitcnt vmcnt(0) between the two writers of the same registers)
        const int kl = (G::PS % 4 != 0) ? min(k, G::PS - 4) : k;
        const unsigned dead = ((unsigned)((G::PS - 1 - k) >> 31) | dead_img) & kOobOffset;
        rav[S][i] = VecLoad<4>::load(rsm, (4u * (unsigned)((img * G::CS + a_cs) * G::PS + kl)) | dead);
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int v = t_ + NP * i;
        const int cq = v & 7, q = (v >> 3) % QPR, ri = (v >> 3) / QPR;
        const unsigned dead = ((unsigned)((NI - 1 - ri) >> 31) | dead_img) & kOobOffset;
        const int x0 = min(4 * q, G::WB - 4);   // the row's last quad ends WITH the row (it re-stages up to 3 pixels)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          rbv[S][i][c] = VecLoad<4>::load(
              rbg, (4u * (unsigned)((img * G::CB + 4 * cq + c) * G::PB + (2 * (s0 + ri) + pky) * G::WB + x0)) | dead);
      }
    };
    auto lstore = [&](auto jc, auto sc, char* Bl, char* Al) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value, S = decltype(sc)::value;
      constexpr int k0 = KC * j, NI = TG::nrows(j);
      const int t_ = (TPW <= 4 ? tid : otid()) - 256;   // k4: registers to spare, the lane constants may be hoisted (enc2 215 -> 199 us; dec3 309 -> 350)
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int a_e4 = (t_ + NP * i) % AQ, a_cs = (t_ + NP * i) / AQ;
        const int k = k0 + 4 * a_e4;
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (k + e < G::PS) ? rav[S][i][e] : 0.f;
        if (G::PS % 4 != 0 && k + 3 >= G::PS && k < G::PS) {   // loaded from PS - 4: element e sits at 4 - PS % 4 + e
          constexpr int R = G::PS % 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = e < R ? rav[S][i][(4 - R + e) & 3] : 0.f;
        }
        if (p.want_db) dbs[i] += (x[0] + x[1]) + (x[2] + x[3]);
        unsigned a1, a2, a3, b1, b2, b3;
        tw_split3(x[0], x[1], a1, a2, a3);
        tw_split3(x[2], x[3], b1, b2, b3);
        // quad m of a 32-block sits at quad position (m with its bits 0 and 1 swapped): the order the B groups read in
        const int a_pos = (a_e4 & ~3) | ((a_e4 & 1) << 1) | ((a_e4 >> 1) & 1);
        char* dst = Al + (a_cs * AP + 4 * a_pos) * 2;
        *reinterpret_cast<bg_u32x2*>(dst) = bg_u32x2{a1, b1};
        *reinterpret_cast<bg_u32x2*>(dst + APLANE) = bg_u32x2{a2, b2};
        *reinterpret_cast<bg_u32x2*>(dst + 2 * APLANE) = bg_u32x2{a3, b3};
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int v = t_ + NP * i;
        const int cq = v & 7, q = (v >> 3) % QPR, ri = (v >> 3) / QPR;
        if (ri < NI) {
          const int x0 = min(4 * q, G::WB - 4);
          if (p.want_dbig && ri >= TG::own_from(j)) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int e = 0; e < 4; ++e) dsum[c] += (x0 + e >= 4 * q) ? rbv[S][i][c][e] : 0.f;   // the row's last quad overlaps its neighbour
          }
          // pixel x0 + e: column parity and x/2 -- immediates when every quad starts on an even column.  XH is odd: the
          // two channel blocks of a pixel are XH * 32 = 32 or 96 B (mod 128) apart, so the 16 lanes of a store group
          // (8 channel quads x 2 pixel quads, the pixels 64 B apart) hit 32 different banks
          char* base = Bl + ri * TG::ROWB + ((cq >> 2) * XH) * TG::PIXB + (cq & 3) * 8 + (G::WB % 2 == 0 ? (x0 >> 1) * TG::PIXB : 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cc = x0 + e;
            unsigned a1, a2, a3, b1, b2, b3;
            tw_split3(rbv[S][i][0][e], rbv[S][i][1][e], a1, a2, a3);
            tw_split3(rbv[S][i][2][e], rbv[S][i][3][e], b1, b2, b3);
            char* dst = G::WB % 2 == 0 ? base + ((e & 1) * 2 * XH + (e >> 1)) * TG::PIXB
                                       : base + ((cc & 1) * 2 * XH + (cc >> 1)) * TG::PIXB;
            *reinterpret_cast<bg_u32x2*>(dst) = bg_u32x2{a1, b1};
            *reinterpret_cast<bg_u32x2*>(dst + BPLANE) = bg_u32x2{a2, b2};
            *reinterpret_cast<bg_u32x2*>(dst + 2 * BPLANE) = bg_u32x2{a3, b3};
          }
        }
      }
    };

    // chunk c is multiplied from buffer c & 1 while chunk c + 1 (register set (c + 1) & 1, loaded two barriers ago) is
    // split into the other buffer and chunk c + 3's loads are issued into the set that just emptied
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    gload(I0{}, I0{}, img_beg);
    gload(std::integral_constant<int, 1 % NCH>{}, I1{}, img_beg + 1 / NCH);
    lstore(I0{}, I0{}, tw_lds, tw_lds + 3 * BPLANE);
    gload(std::integral_constant<int, 2 % NCH>{}, I0{}, img_beg + 2 / NCH);
    lds_barrier();
    int buf = 0, par = 0;   // par: parity of the image's first chunk index
    for (int img = img_beg; img < img_end; ++img) {
      auto body = [&](auto jc, auto sc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int j1 = (j + 1) % NCH, j3 = (j + 3) % NCH;
        char* Bl = tw_lds + (buf ^ 1) * TG::BUF;
#ifndef TW_NO_STAGE
        if (img + (j + 1) / NCH < img_end) lstore(std::integral_constant<int, j1>{}, sc, Bl, Bl + 3 * BPLANE);
        gload(std::integral_constant<int, j3>{}, sc, img + (j + 3) / NCH);
#endif
        lds_barrier();
        buf ^= 1;
      };
      auto image = [&](auto pc) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value;   // set of chunk c + 1 = (P + j + 1) & 1
        body(std::integral_constant<int, 0>{}, std::integral_constant<int, (P + 1) & 1>{});
        if constexpr (NCH > 1) body(std::integral_constant<int, 1>{}, std::integral_constant<int, (P + 2) & 1>{});
        if constexpr (NCH > 2) body(std::integral_constant<int, 2>{}, std::integral_constant<int, (P + 3) & 1>{});
        if constexpr (NCH > 3) body(std::integral_constant<int, 3>{}, std::integral_constant<int, (P + 4) & 1>{});
        if constexpr (NCH > 4) body(std::integral_constant<int, 4>{}, std::integral_constant<int, (P + 5) & 1>{});
        if constexpr (NCH > 5) body(std::integral_constant<int, 5>{}, std::integral_constant<int, (P + 6) & 1>{});
        if constexpr (NCH > 6) body(std::integral_constant<int, 6>{}, std::integral_constant<int, (P + 7) & 1>{});
      };
      if (NCH % 2 == 0 || par == 0) image(I0{});
      else image(I1{});
      par ^= NCH & 1;
    }
    if (p.want_dbig) {   // channel c: the NP / 8 threads of quad c >> 2, in thread order (bit-reproducible)
      float* Dl = reinterpret_cast<float*>(tw_lds);
#pragma unroll
      for (int c = 0; c < 4; ++c) Dl[(tid - 256) * 4 + c] = dsum[c];
      lds_barrier();
      if (tid - 256 < G::CB) {
        const int c = tid - 256;
        float sum = 0.f;
        for (int t = c >> 2; t < NP; t += 8) sum += Dl[t * 4 + (c & 3)];
        sl[G::CS * (G::CB * G::KK + 1) + pky * G::CB + c] = sum;
      }
    }
    if (p.want_db && pky == 0) {   // both row parities staged `small`: one of the pair writes its sums
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int a_e4 = (tid - 256 + NP * i) % AQ, a_cs = (tid - 256 + NP * i) / AQ;
        float s = dbs[i];
#pragma unroll
        for (int d = AQ / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if (a_e4 == 0) sl[G::KK * G::CS * G::CB + a_cs] = s;
      }
    }
  }
}

template <class G, int NBK, int NPW>
inline int launch_tconv_wgrad(const WgradArgs& a, int splits, hipStream_t s) {
  using TG = TWGeo<G, NBK, NPW>;
  static_assert(TG::LDS_BYTES <= 160 * 1024, "twgrad: LDS");
  hipError_t e = hipFuncSetAttribute((const void*)tconv_wgrad_kernel<G, NBK, NPW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     TG::LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  WgradArgs b = a;
  b.nsplits_tw = splits;
  hipLaunchKernelGGL((tconv_wgrad_kernel<G, NBK, NPW>), dim3(16u * (unsigned)((splits + 7) / 8)), dim3(TG::NT), TG::LDS_BYTES, s, b);
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
