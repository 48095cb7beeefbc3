// Conv weight gradient on the bf16 matrix pipe ("bf16x6", bgemm.h) with BOTH operands split ONCE, at staging, and the
// stride-2 gather of `big` done by the LDS's transposing read (ds_read_b64_tr_b16, gfx950).
//
//   dw[cs][cb][ky][kx] = sum_{img,sy,sx} small[img][cs][sy][sx] * big[img][cb][2sy+ky][2sx+kx]
//   per tap (ky,kx): M = cs (A = small), N = cb (B = big), K = (img, sy, sx)
//
// bwgrad.h keeps `big` as an fp32 LDS image and splits every B fragment in registers at every use (44 dependent vector
// instructions per 8 elements, again for every tap that meets the element): the matrix pipe is busy 0.38 of the time.
// Here `big` is staged CHANNEL-INNERMOST -- three bf16 planes [row][column parity][x/2][32 channels], 64 B per pixel
// -- and a B fragment (column = channel cb, 8 consecutive k = 8 consecutive output pixels of a row) is two
// ds_read_b64_tr_b16: each 16-lane group names four PIXEL addresses (any four: the stride-2 walk, a row wrap, costs
// nothing) and receives, per lane = channel, the four pixels' values.  Consecutive output pixels of a tap are
// consecutive entries of one parity plane: 4 x 64 B contiguous, conflict-free.  A tap is an IMMEDIATE offset:
// the taps of a parity class (ky & 1, kx & 1) sit at (4 a XH + c) * 64 B from the class's first, a, c < KS / 2.
//   * a workgroup (8 waves) owns the WHOLE dw for its images: wave = (32 cs rows) x (32 channels) x (the (KS/2)^2 taps
//     of one parity class): 9 x 16 accumulator registers for k6; `small` (A, k-contiguous) is split at staging into
//     planes [cs][k] exactly as in bwgrad.h (one aligned ds_read_b128 per plane and fragment);
//   * K is the FLAT pixel index of an image, 16 at a time, in chunks of NBK blocks; a chunk stages the rows of `big`
//     its pixels touch (<= 2 * rows spanned + KS - 2); phantom k (>= PS, last block) have A = 0 and a clamped B address;
//   * the next chunk's global loads are in flight during the MFMA loop (registers), one barrier pair per chunk;
//   * slab[z][tap][cs][cb] (the lane = cb runs are 128 B), reduced in fixed order by conv_slab_reduce_wave_kernel's
//     tap-major mode.
// Reference: autograd's weight gradient of nn.Conv2d / nn.ConvTranspose2d (models/encoder.py:35-38, decoder.py:43-47).
#pragma once
#include "bgemm.h"
#include "dconv.h"

namespace repo {

typedef short tw_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned tw_u32x4 __attribute__((ext_vector_type(4)));
#define TW_LDS(p) ((__attribute__((address_space(3))) tw_s16x4*)(p))

// global -> LDS, 16 B per lane, lane l at LDS byte m0v + 16 l: no register holds the data
__device__ __forceinline__ void tw_dma_b128(unsigned m0v, unsigned off, tw_u32x4 rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(m0v), "v"(off), "s"(rsrc) : "memory", "m0");
}

template <class G, int NBK_>
struct TWGeo {
  static_assert(G::CB == 32 && G::CS == 64 && G::KS % 2 == 0, "twgrad: 32 -> 64 channel layers with an even kernel");
  static constexpr int NBK = NBK_, NT = 512;
  static constexpr int KC = 16 * NBK;                  // k per chunk
  static constexpr int NBLK = (G::PS + 15) / 16;       // k-blocks per image
  static constexpr int NCH = (NBLK + NBK - 1) / NBK;   // chunks per image
  static constexpr int XH = (G::WB + 1) / 2;
  static constexpr int PIXB = G::CB * 2;               // bytes per pixel and plane
  static constexpr int ROWB = 2 * XH * PIXB;           // bytes per row of `big` and plane
  static constexpr int nblk(int j) { return cmin(NBK, NBLK - j * NBK); }
  static constexpr int row0(int j) { return 2 * ((KC * j) / G::WS); }
  static constexpr int klast(int j) { return cmin(KC * (j + 1), G::PS) - 1; }
  static constexpr int nrows(int j) { return 2 * (klast(j) / G::WS) + G::KS - row0(j); }
  static constexpr int brmax() {
    int m = 0;
    for (int j = 0; j < NCH; ++j) m = cmax(m, nrows(j));
    return m;
  }
  static constexpr int BRMAX = brmax();
  static constexpr int BPLANE = BRMAX * ROWB;
  static constexpr int AP = KC + 8;                    // == 8 (mod 16) bf16: conflict-free ds_read_b128
  static constexpr int APLANE = G::CS * AP * 2;
  static constexpr int RAW_BYTES = 8 * 4 * 1024;        // 8 waves x 4 channel loads x (64 lanes x 16 B): the second B item, by LDS-DMA
  static constexpr int LDS_BYTES = 3 * (BPLANE + APLANE) + RAW_BYTES;
  static constexpr int H2 = G::KS / 2, TPW = H2 * H2;
  static constexpr int AQ = KC / 4;
  static_assert((AQ & (AQ - 1)) == 0 && AQ <= 64, "the bias-gradient lanes of a row are one aligned lane group");
  static constexpr int A_NV = G::CS * AQ, A_PER = (A_NV + NT - 1) / NT;
  static constexpr int NQMAX = (BRMAX * G::WB + 3) / 4;
  static constexpr int B_NV = 8 * NQMAX, B_PER = (B_NV + NT - 1) / NT;
  static_assert(B_PER <= 2, "one B item in registers, one through the raw LDS area");
  static constexpr int SLAB = G::CS * (G::CB * G::KK + 1);   // floats per split: [tap][cs][cb], then db[cs]
};

template <class G, int NBK>
__global__ __launch_bounds__(512) void tconv_wgrad_kernel(WgradArgs p) {
  using TG = TWGeo<G, NBK>;
  constexpr int KC = TG::KC, NCH = TG::NCH, XH = TG::XH, AP = TG::AP, AQ = TG::AQ;
  constexpr int BPLANE = TG::BPLANE, APLANE = TG::APLANE, H2 = TG::H2, TPW = TG::TPW;
  constexpr int A_PER = TG::A_PER, B_PER = TG::B_PER;
  extern __shared__ __attribute__((aligned(16))) char tw_lds[];
  char* Bl = tw_lds;
  char* Al = tw_lds + 3 * BPLANE;
  char* Raw = tw_lds + 3 * (BPLANE + APLANE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wid & 1, cls = wid >> 1;          // 32 cs rows; parity class (ky & 1, kx & 1)
  const int pky = cls >> 1, pkx = cls & 1;
  const int z = blockIdx.x;
  const int img_beg = z * p.imgs_per_split, img_end = min(p.nimg, img_beg + p.imgs_per_split);

  const __amdgpu_buffer_rsrc_t rsm = make_rsrc(p.small, p.small_bytes), rbg = make_rsrc(p.big, p.big_bytes);
  // the same descriptor as four scalars, for the LDS-DMA loads (inline asm), and this wave's 4 KB of the raw area
  tw_u32x4 dma_rsrc = {(unsigned)(uintptr_t)p.big, (unsigned)((uintptr_t)p.big >> 32) & 0xffffu, p.big_bytes, 0x00020000u};
  unsigned raw_m0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(tw_lds + 3 * (TG::BPLANE + TG::APLANE)) + wid * 4096);

  // Lane constants (staging maps, fragment bases) are RECOMPUTED from an opaque copy of the thread index where they are
  // used: kept across the kernel they are a dozen registers the allocator spills, and a reload of a spilled value is a
  // scratch load whose s_waitcnt vmcnt(0) also waits for the prefetch in flight.
  //   A item j: v = tid + 512 j -> (cs = v / AQ, quad of k = v % AQ); B item j: (channel quad tid & 7, pixel quad (tid >> 3) + 64 j)
  //   A fragment: byte offset of (row mt*32 + li, k = 8 lh) in a plane.  B fragment: lane 4q+p of a 16-lane group names
  //   row (pixel) q, channels 16 * (group & 1) + 4p .. + 3; the groups of the upper half-wave hold k + 8
  auto otid = [&]() __attribute__((always_inline)) {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t;
  };
  const int cls_off = (pky * 2 * XH + pkx * XH) * TG::PIXB;

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbs[A_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) dbs[j] = 0.f;

  // the next chunk's operands: A and the first B item wait in registers, the second B item goes global -> LDS by DMA
  // (one 1 KB piece per wave and channel load) and is picked up, split and re-stored with the rest: 16 registers less
  // across the MFMA loop, where 9 x 16 accumulators and two B fragment sets live
  f32x4 rav[A_PER], rbv[4];
  auto gload = [&](auto jc, int img) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int k0 = KC * j, r0 = TG::row0(j), NPIX = TG::nrows(j) * G::WB, NQ = (NPIX + 3) / 4;
    const unsigned dead_img = img < img_end ? 0u : ~0u;
    const int t_ = otid();
    const int b_cq = t_ & 7, b_fq0 = t_ >> 3;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int a_e4 = (t_ + 512 * i) % AQ, a_cs = (t_ + 512 * i) / AQ;
      const int k = k0 + 4 * a_e4;
      static_assert(TG::A_NV % 512 == 0, "every thread has its A items");
      // the quad that holds a row's last PS % 4 pixels is loaded from PS - 4 (never past the row: the last row of
      // `small` ends the buffer) and rotated into place by the store.  Inactive items read out of range (zeros): the
      // sign of (PS - 1 - k) becomes the offset's top bit -- arithmetic, not a select the compiler turns into branches
      // around the loads (with an s_waitcnt vmcnt(0) between the two writers of the same registers)
      const int kl = (G::PS % 4 != 0) ? min(k, G::PS - 4) : k;
      const unsigned dead = ((unsigned)((G::PS - 1 - k) >> 31) | dead_img) & kOobOffset;
      rav[i] = VecLoad<4>::load(rsm, (4u * (unsigned)((img * G::CS + a_cs) * G::PS + kl)) | dead);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int fq = b_fq0 + 64 * i;
      const unsigned dead = ((unsigned)((NQ - 1 - fq) >> 31) | dead_img) & kOobOffset;
      const int f0 = min(4 * fq, NPIX - 4);   // the band's last quad ends WITH the band (it re-stages up to 3 pixels)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned off = (4u * (unsigned)((img * G::CB + 4 * b_cq + c) * G::PB + r0 * G::WB + f0)) | dead;
        if (i == 0) rbv[c] = VecLoad<4>::load(rbg, off);
        else   // as asm: behind the builtin the compiler orders EVERY later LDS read after the DMA with s_waitcnt vmcnt(0)
          tw_dma_b128(raw_m0 + c * 1024, off, dma_rsrc);
      }
    }
  };
  auto lstore = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int k0 = KC * j, NPIX = TG::nrows(j) * G::WB, NQ = (NPIX + 3) / 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces too: the compiler does not see that dependence
    const int t_ = otid();
    const int b_cq = t_ & 7, b_fq0 = t_ >> 3, lane = t_ & 63;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int a_e4 = (t_ + 512 * i) % AQ, a_cs = (t_ + 512 * i) / AQ;
      if (TG::A_NV % 512 == 0 || a_cs < G::CS) {
        const int k = k0 + 4 * a_e4;
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (k + e < G::PS) ? rav[i][e] : 0.f;
        if (G::PS % 4 != 0 && k + 3 >= G::PS && k < G::PS) {   // loaded from PS - 4: element e sits at 4 - PS % 4 + e
          constexpr int R = G::PS % 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = e < R ? rav[i][(4 - R + e) & 3] : 0.f;
        }
        dbs[i] += (x[0] + x[1]) + (x[2] + x[3]);
        unsigned a1, a2, a3, b1, b2, b3;
        bg_split3(x[0], x[1], a1, a2, a3);
        bg_split3(x[2], x[3], b1, b2, b3);
        char* dst = Al + (a_cs * AP + 4 * a_e4) * 2;
        *reinterpret_cast<bg_u32x2*>(dst) = bg_u32x2{a1, b1};
        *reinterpret_cast<bg_u32x2*>(dst + APLANE) = bg_u32x2{a2, b2};
        *reinterpret_cast<bg_u32x2*>(dst + 2 * APLANE) = bg_u32x2{a3, b3};
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int fq = b_fq0 + 64 * i;
      if (fq < NQ) {
        const int f0 = min(4 * fq, NPIX - 4);
        int rr = f0 / G::WB, cc = f0 % G::WB;
        f32x4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          v[c] = i == 0 ? rbv[c] : *reinterpret_cast<const f32x4*>(Raw + (wid * 4 + c) * 1024 + lane * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          {
            unsigned a1, a2, a3, b1, b2, b3;
            bg_split3(v[0][e], v[1][e], a1, a2, a3);
            bg_split3(v[2][e], v[3][e], b1, b2, b3);
            char* dst = Bl + ((rr * 2 + (cc & 1)) * XH + (cc >> 1)) * TG::PIXB + b_cq * 8;
            *reinterpret_cast<bg_u32x2*>(dst) = bg_u32x2{a1, b1};
            *reinterpret_cast<bg_u32x2*>(dst + BPLANE) = bg_u32x2{a2, b2};
            *reinterpret_cast<bg_u32x2*>(dst + 2 * BPLANE) = bg_u32x2{a3, b3};
          }
          ++cc;
          if (cc == G::WB) cc = 0, ++rr;
        }
      }
    }
  };
  auto compute = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int NB = TG::nblk(j), r0 = TG::row0(j), NS = NB * TPW;
    // byte offset of this lane's pixel quad r of block b: k = KC j + 16 b + 8 lh + 4 r + bq.  Recomputed per block from an
    // opaque copy of bq: as loop invariants the 2 NBLK offsets would be hoisted out of the image loop and spilled
    auto baddr = [&](int b, int r) __attribute__((always_inline)) {
      const int lane = otid() & 63;
      const int k = min(KC * j + 16 * b + 8 * (lane >> 5) + 4 * r + ((lane & 15) >> 2), G::PS - 1);
      const int sy = k / G::WS, sx = k - sy * G::WS;
      return ((2 * sy - r0) * 2 * XH + sx) * TG::PIXB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2 + cls_off;
    };
    auto load_b = [&](bg_bf16x8(&fb)[3], int o0, int o1, int t) __attribute__((always_inline)) {
      const int toff = ((t / H2) * 4 * XH + (t % H2)) * TG::PIXB;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const tw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(TW_LDS(Bl + q * BPLANE + o0 + toff));
        const tw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(TW_LDS(Bl + q * BPLANE + o1 + toff));
        fb[q] = __builtin_bit_cast(bg_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    auto load_a = [&](bg_bf16x8(&fa)[3], int b) __attribute__((always_inline)) {
      const int lane = otid() & 63;
      const int abase = ((mt * 32 + (lane & 31)) * AP + 8 * (lane >> 5)) * 2;
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bg_bf16x8*>(Al + q * APLANE + abase + 32 * b);
    };
    // one step = one (block, tap): its six MFMAs run while the NEXT step's six transposing reads are in flight
    bg_bf16x8 fa[3], fb[2][3];
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
      f32x16 c = acc[t];  // smallest terms first
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[s & 1][1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[s & 1][2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[s & 1][0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[s & 1][1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[s & 1][0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[s & 1][0], c, 0, 0, 0);
      acc[t] = c;
      if (t == TPW - 1 && b + 1 < NB) load_a(fa, b + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (img_beg < img_end) {
    gload(std::integral_constant<int, 0>{}, img_beg);
    for (int img = img_beg; img < img_end; ++img) {
      auto body = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        lstore(jc);
        __syncthreads();
        gload(std::integral_constant<int, (j + 1) % NCH>{}, img + (j + 1 == NCH ? 1 : 0));
        compute(jc);
        __syncthreads();
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
  }

  // ---- slab[z][tap][cs][cb]: lane = cb, accumulator register r = row (r & 3) + 8 (r >> 2) + 4 lh of the 32
  float* sl = p.slab + (size_t)z * TG::SLAB;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int ky = 2 * (t / H2) + pky, kx = 2 * (t % H2) + pkx;
    float* dst = sl + ((size_t)(ky * G::KS + kx) * G::CS + mt * 32 + 4 * lh) * G::CB + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * G::CB] = acc[t][r];
  }
  if (p.want_db) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int a_e4 = (tid + 512 * i) % AQ, a_cs = (tid + 512 * i) / AQ;
      float s = dbs[i];
#pragma unroll
      for (int d = AQ / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
      if (a_e4 == 0 && (TG::A_NV % 512 == 0 || a_cs < G::CS)) sl[G::KK * G::CS * G::CB + a_cs] = s;
    }
  }
}

template <class G, int NBK>
inline int launch_tconv_wgrad(const WgradArgs& a, int splits, hipStream_t s) {
  using TG = TWGeo<G, NBK>;
  static_assert(TG::LDS_BYTES <= 160 * 1024, "twgrad: LDS");
  hipError_t e = hipFuncSetAttribute((const void*)tconv_wgrad_kernel<G, NBK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     TG::LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((tconv_wgrad_kernel<G, NBK>), dim3((unsigned)splits), dim3(512), TG::LDS_BYTES, s, a);
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
