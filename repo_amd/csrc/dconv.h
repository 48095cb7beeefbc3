// Direct (patch-resident) stride-2 convolutions on the fp32 matrix cores of gfx950.
//
// igemm.h stages an im2col slice per K step: every input element is fetched (KS/2)^2 times by
// dword-granular gathers, and in-kernel stamps show those gathers -- not the MFMAs -- set the pace
// (the CU's vector-memory front end needs 30-80 cycles per wave-level gather instruction).  Here the
// workgroup copies the RAW input rows its output pixels need into LDS once per channel chunk, with
// 16-byte buffer loads of contiguous memory, and the MFMA operand fragments are read straight from
// that patch:   LDS address = base(pixel of this lane) + offset(k = channel, ky, kx)
// is separable, the offset part is a compile-time constant of the unrolled K loop, so a fragment
// read is one ds_read_b32 with an immediate offset and no address arithmetic.
//
//   down : small[img][cs][sy][sx] = sum_{cb,ky,kx} big[img][cb][2sy+ky][2sx+kx] w[cs][cb][ky][kx]
//          M = CS (weights: A operand, [k][m] slice in LDS), N = pixels (img,sy,sx), K = (cb,ky,kx)
//
// Pipeline: one register staging set, two LDS buffers, one barrier per channel chunk (the loads of
// chunk t+1 are in flight during the MFMAs of chunk t); occupancy (3-5 workgroups per CU) hides the rest.
#pragma once
#include "vgemm.h"

namespace repo {

// LDWPAD: padding of the weight slice's LDS rows.  The k-major staging stores (4 dwords of one weight row to 4
// LDS rows) conflict by the row pitch: 2 suits k-per-chunk / 4 = 8 or 18 rows per lane group, 1 suits 25 (dec2:
// 322 -> 309 us, A/B on one box)
template <int BM_, int BN_, int CK_, int WM_, int WN_, int LDWPAD_ = 2>
struct DTile {
  static constexpr int BM = BM_, BN = BN_, CK = CK_, WM = WM_, WN = WN_, LDWPAD = LDWPAD_;
  static constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  static constexpr int NT = WM * WN * 64;
  static_assert(TM >= 1 && TN >= 1 && BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile / wave grid mismatch");
};

#ifdef REPO_IGEMM_STAMPS
#define REPO_STAMP_FLUSH(nslices)                                              \
  do {                                                                         \
    REPO_STAMP(4);                                                             \
    if ((threadIdx.x & 63) == 0 && blockIdx.x % 16 == 0) {                     \
      for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_igemm_stamps[i_], st_[i_]);  \
      atomicAdd(&g_igemm_stamps[6], 1ull);                                     \
      atomicAdd(&g_igemm_stamps[7], (unsigned long long)(nslices));            \
    }                                                                          \
  } while (0)
#else
#define REPO_STAMP_FLUSH(nslices)
#endif

// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MB L2).  Neighbouring pixel
// tiles share input rows / planes, so dispatch slot b is mapped to a tile id such that every XCD walks a
// CONTIGUOUS range of tiles: XCD x owns [x*q + min(x,r), ...) with q = n/8, r = n%8 (a bijection on [0,n)).
__device__ __forceinline__ int xcd_tile(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return x * q + min(x, r) + i;
}

constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// 4 consecutive input elements as floats (fp32 frames: one 16-byte load; uint8 frames: one dword).
template <class InT>
struct Patch4;
template <>
struct Patch4<float> {
  typedef f32x4 raw_t;
  static constexpr int BYTES = 4;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  }
  static __device__ __forceinline__ f32x4 cvt(raw_t v) { return v; }
};
template <>
struct Patch4<uint8_t> {
  typedef unsigned raw_t;
  static constexpr int BYTES = 1;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
  }
  static __device__ __forceinline__ f32x4 cvt(raw_t v) {
    f32x4 o;
    o[0] = pix_norm((uint8_t)(v & 0xff));
    o[1] = pix_norm((uint8_t)((v >> 8) & 0xff));
    o[2] = pix_norm((uint8_t)((v >> 16) & 0xff));
    o[3] = pix_norm((uint8_t)(v >> 24));
    return o;
  }
};

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

struct DownArgs {
  const void* big;
  const float* w;
  const float* bias;
  const float* aux;
  float* out;
  int nimg, epi;
  unsigned big_bytes, w_bytes;
  float* chan_part;  // nullable: [pixel tiles][CS] sums of the written values per output channel (bias gradients)
  unsigned char* cmask;  // nullable (REPO_EPI_RELU): the output's CHANNEL-QUAD mask, repo_hip.h REPO_EPI_MUL_CMASK
};

// The epilogue of the stride-2 "down" kernels (dconv_down_kernel here, bconv_down_kernel in bconv.h): `acc` holds the
// workgroup's BM x BN tile in the 32 x 32 MFMA layout (T::TM x T::TN tiles per wave, waves as T::WM x T::WN), `lds` is the
// kernel's LDS (free by now; >= (NT / 64) * 32 * 36 + (NT / 64) * TM * 32 floats).
template <class G, class T>
__device__ __forceinline__ void dconv_down_epilogue(const DownArgs& p, float* lds, f32x16 (&acc)[T::TM][T::TN], int n0,
                                                    int m0, int Ntot) {
  constexpr int BM = T::BM, NT = T::NT, TM = T::TM, TN = T::TN;
  constexpr int EP = 36;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / T::WN, wn = wid % T::WN;
  const int li = lane & 31, lh = lane >> 5;
  // ---- epilogue.  An accumulator tile has its pixel on the lane and 16 channel rows in registers: stored as it
  // stands, that is 16 dword stores (+ 16 dword loads of the ReLU operand) of 128 contiguous bytes per half-wave,
  // and the store ISSUE, not the bytes, set the pace (ablation on one box, round 3: the epilogue was 11 % of decoder
  // conv3's data gradient, 40 % of the 3-channel layers').  Each wave therefore turns its tile through a private
  // LDS strip (the K loop's buffers are free by now) so that a lane owns ONE channel and FOUR consecutive pixels:
  // one 16-byte load of the ReLU operand, one 16-byte store (raw-buffer accesses need dword alignment only); a quad
  // that runs over the end of an image or of the tensor falls back to dwords.
  // (strip pitch EP = 36: 16-byte aligned rows, the quads of 8 consecutive channels on distinct banks)
  __syncthreads();  // every wave is done with the last chunk's operands
  float* strip = lds + wid * 32 * EP;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out, 4u * (unsigned)Ntot * G::CS);
  // channel sums of what this workgroup writes (the bias gradient of the layer whose pre-activation gradient this
  // is): a lane's quad, then the 8 lanes of a channel by shuffles, summed over the wave's pixel tiles in registers;
  // the waves' rows meet in LDS and ONE thread per channel adds them in a fixed order -- no atomics, reproducible
  float csum[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) csum[i][q4] = 0.f;
  const __amdgpu_buffer_rsrc_t raux =
      make_rsrc((p.aux && p.epi != REPO_EPI_FILM_RELU) ? p.aux : p.out, (p.epi == REPO_EPI_MUL_MASK4 ? 1u : 4u) * (unsigned)Ntot * G::CS);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + li] = acc[i][j][r];
      __builtin_amdgcn_wave_barrier();
      const int nb = n0 + (wn * TN + j) * 32, mt = m0 + (wm * TM + i) * 32;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int ml = pass * 8 + (lane >> 3), q = lane & 7;
        const int m = mt + ml, n = nb + 4 * q;
        f32x4 v = *reinterpret_cast<const f32x4*>(strip + ml * EP + 4 * q);
        float qs = 0.f;  // this lane's contribution to its channel's sum
        if (m < G::CS && n < Ntot) {
          const float bv = p.bias ? p.bias[m] : 0.f;
          const int img = n / G::PS, pix = n % G::PS;
          const unsigned o = (unsigned)((img * G::CS + m) * G::PS + pix);
          if (pix + 3 < G::PS && n + 3 < Ntot) {
            f32x4 a4 = {1.f, 1.f, 1.f, 1.f};
            if (p.epi == REPO_EPI_MUL_DRELU) {
              a4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, 4u * o, 0, 0));
            } else if (p.epi == REPO_EPI_MUL_MASK4) {
              // bits o .. o+3 of the quad mask: one byte when the quad is aligned (always, for PS % 4 == 0), else two
              unsigned bits = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(raux, o >> 2, 0, 0) >> (o & 3);
              if (o & 3) bits |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(raux, (o >> 2) + 1, 0, 0) << (4 - (o & 3));
#pragma unroll
              for (int e = 0; e < 4; ++e) a4[e] = (bits >> e) & 1u ? 1.f : 0.f;
            }
            float fsc = 1.f, fsh = 0.f;   // REPO_EPI_FILM_RELU: this image's (scale, shift) of channel m
            if (p.epi == REPO_EPI_FILM_RELU) {
              fsc = p.aux[(size_t)(2 * img) * G::CS + m];
              fsh = p.aux[(size_t)(2 * img + 1) * G::CS + m];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = v[e] + bv;
              if (p.epi == REPO_EPI_RELU) x = fmaxf(x, 0.f);
              else if (p.epi == REPO_EPI_MUL_DRELU || p.epi == REPO_EPI_MUL_MASK4) x = a4[e] > 0.f ? x : 0.f;
              else if (p.epi == REPO_EPI_FILM_RELU) x = fmaxf(fmaf(fsc, x, fsh), 0.f);
              v[e] = x;
            }
            qs = (v[0] + v[1]) + (v[2] + v[3]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rout, 4u * o, 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ne = n + e;
              if (ne < Ntot) {
                const int oe = ((ne / G::PS) * G::CS + m) * G::PS + ne % G::PS;
                float x = v[e] + bv;
                if (p.epi == REPO_EPI_RELU) x = fmaxf(x, 0.f);
                else if (p.epi == REPO_EPI_FILM_RELU)
                  x = fmaxf(fmaf(p.aux[(size_t)(2 * (ne / G::PS)) * G::CS + m], x, p.aux[(size_t)(2 * (ne / G::PS) + 1) * G::CS + m]), 0.f);
                else if (p.epi == REPO_EPI_MUL_DRELU) x = p.aux[oe] > 0.f ? x : 0.f;
                else if (p.epi == REPO_EPI_MUL_MASK4)
                  x = (reinterpret_cast<const unsigned char*>(p.aux)[oe >> 2] >> (oe & 3)) & 1 ? x : 0.f;
                p.out[oe] = x;
                qs += x;
              }
            }
          }
        }
        if (p.chan_part) {  // wave-uniform
          qs += __shfl_xor(qs, 1, 64);
          qs += __shfl_xor(qs, 2, 64);
          qs += __shfl_xor(qs, 4, 64);
          csum[i][pass] += qs;
        }
      }
      if (p.cmask) {  // wave-uniform.  A lane: channel quad lane / 8 of the tile's 32 channels, pixels 4 (lane % 8) .. + 3:
        // one byte per pixel, bit c = relu(channel 4 quad + c) > 0 -- what the layer's data gradient needs of the
        // activation (its drain owns four channels of a pixel: one byte instead of four floats)
        const int cq = lane >> 3, q = lane & 7;
        const int m4 = mt + 4 * cq, n = nb + 4 * q;
        if (m4 < G::CS && n < Ntot) {
          f32x4 rws[4];
          float bq[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            rws[c] = *reinterpret_cast<const f32x4*>(strip + (4 * cq + c) * EP + 4 * q);
            bq[c] = p.bias ? p.bias[m4 + c] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ne = n + e;
            if (ne < Ntot) {
              unsigned bits = 0;
#pragma unroll
              for (int c = 0; c < 4; ++c) bits |= (rws[c][e] + bq[c] > 0.f ? 1u : 0u) << c;
              p.cmask[((size_t)(ne / G::PS) * (G::CS / 4) + (m4 >> 2)) * G::PS + ne % G::PS] = (unsigned char)bits;
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  if (p.chan_part) {
    float* red = lds + (NT / 64) * 32 * EP;  // [wave][TM * 32]
    if ((lane & 7) == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) red[wid * (TM * 32) + i * 32 + pass * 8 + (lane >> 3)] = csum[i][pass];
    }
    __syncthreads();
    if (tid < BM && m0 + tid < G::CS) {
      const int wmc = tid / (TM * 32), c = tid % (TM * 32);
      float t = 0.f;
#pragma unroll
      for (int x = 0; x < T::WN; ++x) t += red[(wmc * T::WN + x) * (TM * 32) + c];
      p.chan_part[(size_t)blockIdx.x * G::CS + m0 + tid] = t;
    }
  }
}

template <class G, class BigT, class T>
__global__ __launch_bounds__(T::NT) void dconv_down_kernel(DownArgs p) {
  constexpr int BM = T::BM, BN = T::BN, CK = T::CK, NT = T::NT, TM = T::TM, TN = T::TN;
  constexpr int KSL = CK * G::KK;  // k per chunk
  constexpr int NSL = G::CB / CK;
  static_assert(G::CB % CK == 0 && KSL % 4 == 0, "channel chunk must divide CB and give k % 4 == 0");
  constexpr int LDW = BM + T::LDWPAD;
  // ---- patch geometry: a tile is BN consecutive pixels of (img, sy, sx); image i of the tile needs
  // input rows [2*f_i, 2*l_i + KS) -- full width, hence ONE contiguous span per (image, channel)
  constexpr int ROWS_FULL = 2 * (G::HS - 1) + G::KS;
  constexpr int LENFULL = (ROWS_FULL * G::WB + 3) & ~3;
  constexpr int NIMG_MAX = (BN - 1) / G::PS + 2;
  constexpr int PLMAX = cmin(NIMG_MAX * LENFULL,
                             ((2 * (BN / G::WS + 2) + NIMG_MAX * (G::KS - 2)) * G::WB + 4 * NIMG_MAX + 3) & ~3);
  constexpr int PLV = PLMAX / 4;
  constexpr int W_NV = BM * KSL / 4, W_KV = KSL / 4;  // weight vectors per chunk / per row
  constexpr int W_PER = (W_NV + NT - 1) / NT, P_PER = (CK * PLV + NT - 1) / NT;
  constexpr int NBUF = NSL > 1 ? 2 : 1;  // a single channel chunk (the 3-channel layers) needs one buffer
  constexpr int EP = 36;  // epilogue strip pitch (below)
  // (+ one row of per-wave channel sums behind the strips: DownArgs::chan_part)
  __shared__ __attribute__((aligned(16))) float lds[cmax(NBUF * KSL * LDW + NBUF * CK * PLMAX, (NT / 64) * 32 * EP + (NT / 64) * TM * 32)];
  float* Wl = lds;
  float* Pl = lds + NBUF * KSL * LDW;

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
  // span of image i: rows [2*f_i, 2*l_i+KS); LDS start S_i (floats, multiple of 4)
  const int la_ = (ia == ib) ? lb : G::HS - 1;
  const int lenA = ((2 * (la_ - fa) + G::KS) * G::WB + 3) & ~3;
  auto span_start = [&](int i) __attribute__((always_inline)) { return i == ia ? 0 : lenA + (i - ia - 1) * LENFULL; };
  const int PL = span_start(ib) + (ib == ia ? lenA : ((2 * lb + G::KS) * G::WB + 3) & ~3);

  const __amdgpu_buffer_rsrc_t rbig = make_rsrc(p.big, p.big_bytes), rw = make_rsrc(p.w, p.w_bytes);

  // ---- staging roles
  unsigned woff[W_PER];
  int wlds[W_PER];
#pragma unroll
  for (int j = 0; j < W_PER; ++j) {
    const int v = tid + j * NT, kv = v % W_KV, m = v / W_KV;
    const bool act = (W_NV % NT == 0) || v < W_NV;
    woff[j] = act ? 4u * (unsigned)(min(m0 + m, G::CS - 1) * (G::CB * G::KK) + kv * 4) : kOobOffset;
    wlds[j] = act ? (kv * 4) * LDW + m : -1;
  }
  unsigned poff[P_PER];
  int plds[P_PER];
#pragma unroll
  for (int j = 0; j < P_PER; ++j) {
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
    poff[j] = act ? (unsigned)Patch4<BigT>::BYTES * (unsigned)((i * G::CB + c) * G::PB + 2 * f * G::WB + rel)
                  : kOobOffset;
    plds[j] = act ? c * PLMAX + q : -1;
  }
  constexpr unsigned W_STEP = 4u * KSL, P_STEP = (unsigned)Patch4<BigT>::BYTES * CK * G::PB;

  // ---- per-lane LDS bases of the B (pixel) fragments
  int bbase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = min(n0 + (wn * TN + j) * 32 + li, Ntot - 1);
    const int i = n / G::PS, pix = n % G::PS;
    const int f = (i == ia) ? fa : 0;
    bbase[j] = span_start(i) + 2 * (pix / G::WS - f) * G::WB + 2 * (pix % G::WS);
  }
  int bbase1[TN];  // + lh: the two k of an MFMA step are horizontally adjacent taps (always, for even KS)
#pragma unroll
  for (int j = 0; j < TN; ++j) bbase1[j] = bbase[j] + lh;
  const int abase = lh * LDW + wm * (TM * 32) + li;

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
    for (int j = 0; j < W_PER; ++j)
      rwv[j] = VecLoad<4>::load(rw, woff[j] == kOobOffset ? kOobOffset : woff[j] + (unsigned)t * W_STEP);
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
      rpv[j] = Patch4<BigT>::load(rbig, poff[j] == kOobOffset ? kOobOffset : poff[j] + (unsigned)t * P_STEP);
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    float* wl = Wl + buf * KSL * LDW;
    float* pl = Pl + buf * CK * PLMAX;
#pragma unroll
    for (int j = 0; j < W_PER; ++j)
      if ((W_NV % NT == 0) || wlds[j] >= 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) wl[wlds[j] + e * LDW] = rwv[j][e];
      }
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
      if (plds[j] >= 0) *reinterpret_cast<f32x4*>(pl + plds[j]) = Patch4<BigT>::cvt(rpv[j]);
  };
  auto koff = [](int k) constexpr { return (k / G::KK) * PLMAX + ((k % G::KK) / G::KS) * G::WB + (k % G::KK) % G::KS; };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float* wl = Wl + buf * KSL * LDW + abase;
    const float* pl = Pl + buf * CK * PLMAX;
#pragma unroll
    for (int ks = 0; ks < KSL / 2; ++ks) {
      const int o0 = koff(2 * ks), o1 = koff(2 * ks + 1);
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = wl[2 * ks * LDW + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = (o1 == o0 + 1) ? pl[bbase1[j] + o0] : pl[bbase[j] + (lh ? o1 : o0)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };

  REPO_STAMP_DECL
  gload(0);
  lstore(0);
  __syncthreads();
  REPO_STAMP(5);
  if (NSL == 1) {
    compute(0);
  } else {
    int buf = 0;
    for (int t = 0; t < NSL; ++t) {
      gload(min(t + 1, NSL - 1));
      __builtin_amdgcn_sched_barrier(0);
      REPO_STAMP(0);
      compute(buf);
      __builtin_amdgcn_sched_barrier(0);
      REPO_STAMP(1);
      lstore(buf ^ 1);
      REPO_STAMP(2);
      __syncthreads();
      REPO_STAMP(3);
      buf ^= 1;
    }
  }

  dconv_down_epilogue<G, T>(p, lds, acc, n0, m0, Ntot);
  REPO_STAMP_FLUSH(NSL);
}

template <class G, class BigT, class T>
inline int launch_dconv_down(const DownArgs& a, hipStream_t s) {
  const long gx = ((long)a.nimg * G::PS + T::BN - 1) / T::BN, gy = (G::CS + T::BM - 1) / T::BM;
  hipLaunchKernelGGL((dconv_down_kernel<G, BigT, T>), dim3((unsigned)gx, (unsigned)gy), dim3(T::NT), 0, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// ---------------------------------------------------------------------------------------- wgrad
//   dw[cs][cb][ky][kx] = sum_{img,sy,sx} small[img][cs][sy][sx] * big[img][cb][2sy+ky][2sx+kx]
//   M = cs (A = small), N = (cb,ky,kx) (B = big), K = (img, sy, sx), split over image groups into slabs
//   [z][CS][NW+1] (column NW = bias gradient of `small`), reduced in fixed order by the caller.
// A chunk is GI images x RB rows of `small` (and the 2*RB+KS-2 rows of `big` under them): per (image,
// channel) both are ONE contiguous run, copied to LDS as they lie in memory with 16-byte loads.  Both
// MFMA operands are then read straight from those copies:
//   A(m, k=(g,sy,sx))  = Al[(g*BM + m)*AP + sy*WS + sx]               lane: m, +1 for the odd k
//   B(k, n=(cb,ky,kx)) = Bl[(g*CBT + cb)*BP + (2sy+ky)*WB + 2sx + kx]  lane: (cb,ky,kx), +2 for the odd k
// k runs over sx pairs of a row; for odd WS the second k of a row's last pair does not exist and the
// odd lanes' A value is forced to zero there.
struct WgradArgs {
  const float* small;
  const void* big;
  float* slab;
  int nimg, imgs_per_split, want_db;
  unsigned small_bytes, big_bytes;
  int nsplits_tw = 0;   // twgrad.h: splits (its grid is padded to whole XCD rounds)
  int want_dbig = 0;    // twgrad.h: also the channel sums of `big`
};

// SPREAD: 0 = the next chunk's global loads are issued back to back in front of the MFMA loop; n > 0 = one at a
// time between the k-pairs of its first 1/n (A/B on one box, round 3: enc1 171 -> 164 us, enc2 370 -> 360 with
// n = 2; dec3 542 -> 572, enc3 226 -> 230: per layer)
template <int BM_, int BN_, int WM_, int WN_, int GI_, int RB_, int SPREAD_ = 0>
struct WTile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, GI = GI_, RB = RB_, SPREAD = SPREAD_;
  static constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN), NT = WM * WN * 64;
  static_assert(TM >= 1 && TN >= 1 && BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile / wave grid mismatch");
};

constexpr int pitch4(int len) {  // >= len, multiple of 4, == 4 (mod 8): at most 2-way conflicts for row-strided lanes
  int p = (len + 3) & ~3;
  return (p / 4) % 2 ? p : p + 4;
}

template <class G, class BigT, class T>
__global__ __launch_bounds__(T::NT) void dconv_wgrad_kernel(WgradArgs p) {
  constexpr int BM = T::BM, BN = T::BN, TM = T::TM, TN = T::TN, NT = T::NT, GI = T::GI, RB = T::RB;
  constexpr int NW = G::CB * G::KK;
  constexpr int NB = (G::HS + RB - 1) / RB;  // row bands per image
  constexpr int BR = 2 * RB + G::KS - 2;     // big rows under a band
  // rows of `big` keep their memory pitch in LDS unless that pitch is a multiple of the 64 banks (the
  // 64-pixel frames): lanes (ky, kx) of a B fragment would then hit one bank KS times over
  constexpr int BRP = (G::WB % 64 == 0) ? G::WB + 4 : G::WB;
  constexpr int ALEN = RB * G::WS, BLEN = BR * BRP;
  constexpr int AP = pitch4(ALEN), BP = pitch4(BLEN);
  constexpr int CBT = cmin(G::CB, (BN + G::KK - 2) / G::KK + 1);  // big channels an N tile can touch
  constexpr int A_VPC = (ALEN + 3) / 4, B_VPC = (BR * G::WB + 3) / 4;  // vectors copied per (image, channel)
  constexpr int A_NV = GI * BM * A_VPC, B_NV = GI * CBT * B_VPC;
  constexpr int A_PER = (A_NV + NT - 1) / NT, B_PER = (B_NV + NT - 1) / NT;
  constexpr int SXP = (G::WS + 1) / 2;
  __shared__ __attribute__((aligned(16))) float lds[GI * BM * AP + GI * CBT * BP + 64];
  float* Al = lds;
  float* Bl = lds + GI * BM * AP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / T::WN, wn = wid % T::WN;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its
  // own L2).  All tiles of one split z read the same images, so they are placed on ONE XCD: within a
  // group of 8 splits, workgroup L takes split (L % 8) and tile (L / 8) of that split.
  int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
  {
    const int tpz = gridDim.x * gridDim.y, gz = gridDim.z;
    const int L = bx + gridDim.x * (by + gridDim.y * z);
    const int full = (gz / 8) * 8 * tpz;  // workgroups covered by complete groups of 8 splits
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

  // ---- staging roles: element offsets at (image 0, row band 0); g = image within the group
  int aoff[A_PER], boff[B_PER], ag[A_PER], bg[B_PER];
  int alds[A_PER], blds[B_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int v = tid + j * NT, e4 = v % A_VPC, rm = v / A_VPC;
    const int m = rm % BM, g = rm / BM;
    const bool act = g < GI && m0 + m < G::CS;
    ag[j] = act ? g : 1 << 20;
    aoff[j] = (g * G::CS + m0 + m) * G::PS + e4 * 4;
    alds[j] = (g * BM + m) * AP + e4 * 4;
  }
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int v = tid + j * NT, e4 = v % B_VPC, rc = v / B_VPC;
    const int c = rc % CBT, g = rc / CBT;
    const bool act = g < GI && cbf + c < G::CB;
    bg[j] = act ? g : 1 << 20;
    boff[j] = (g * G::CB + cbf + c) * G::PB + e4 * 4;
    blds[j] = (g * CBT + c) * BP + (BRP == G::WB ? e4 * 4 : (e4 * 4 / G::WB) * BRP + e4 * 4 % G::WB);
  }

  // ---- per-lane fragment bases
  int abase[TM], bbase[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) abase[i] = ((wm * TM + i) * 32 + li) * AP + lh;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = min(n0 + (wn * TN + j) * 32 + li, NW - 1);
    const int cb = n / G::KK, r = n % G::KK;
    bbase[j] = (cb - cbf) * BP + (r / G::KS) * BRP + r % G::KS + 2 * lh;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float dbsum = 0.f;  // bias-gradient partial: row (tid % BM), element subset (tid / BM)

  f32x4 rav[A_PER];
  typename Patch4<BigT>::raw_t rbv[B_PER];
  // global loads of chunk c, one staged vector (piece q of NLD) at a time: the pieces of chunk c+1 are issued
  // BETWEEN the row blocks of chunk c's MFMA loop -- issued back to back in front of it they queue in the CU's
  // vector-memory front end (in-kernel stamps: 1800-3100 cycles per chunk in which the wave issues nothing else)
  constexpr int NLD = A_PER + B_PER;
  int ld_img0 = 0, ld_abias = 0, ld_bbias = 0;
  auto gload_begin = [&](int c) __attribute__((always_inline)) {
    const int grp = c / NB, band = c % NB;
    const int r0 = band * RB;
    ld_img0 = img_beg + grp * GI;
    ld_abias = ld_img0 * G::CS * G::PS + r0 * G::WS;
    ld_bbias = ld_img0 * G::CB * G::PB + 2 * r0 * G::WB;
  };
  auto gload_piece = [&](int q) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j)
      if (j == q)
        rav[j] = VecLoad<4>::load(rsm, ld_img0 + ag[j] < img_end ? 4u * (unsigned)(aoff[j] + ld_abias) : kOobOffset);
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      if (j + A_PER == q)
        rbv[j] = Patch4<BigT>::load(
            rbg, ld_img0 + bg[j] < img_end ? (unsigned)Patch4<BigT>::BYTES * (unsigned)(boff[j] + ld_bbias) : kOobOffset);
  };
  auto gload = [&](int c) __attribute__((always_inline)) {
    gload_begin(c);
#pragma unroll
    for (int q = 0; q < NLD; ++q) gload_piece(q);
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j)
      if ((A_NV % NT == 0) || tid + j * NT < A_NV) *reinterpret_cast<f32x4*>(Al + alds[j]) = rav[j];
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      if ((B_NV % NT == 0) || tid + j * NT < B_NV) *reinterpret_cast<f32x4*>(Bl + blds[j]) = Patch4<BigT>::cvt(rbv[j]);
  };
  // K of a chunk = the pixels (g, sy, sx) of `small`, walked as FLAT pairs kf = sy*WS + sx over a band (the A copy
  // is one contiguous run per (image, channel), so A needs no row logic at all): pairing inside rows instead
  // pads every odd-width row with a phantom k (WS = 5: 6 MFMA k-slots per 5 pixels, 13: 14 per 13).  B of pixel
  // kf sits at 2*sy*BRP + 2*sx: the odd k of a pair is 2 floats on, or -- when the pair straddles a row end --
  // 2*BRP - 2*WS + 2 floats on; which of the two is a compile-time property of the pair, so each B tile keeps two
  // per-lane bases (same row / straddling) and the pair's offset is an immediate.
  int bbase_w[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bbase_w[j] = bbase[j] - 2 * lh + lh * (2 * BRP - 2 * G::WS + 2);
  constexpr int RL = G::HS - (NB - 1) * RB;  // rows of the last band
  auto compute = [&](auto rows_tag, int c) __attribute__((always_inline)) {
    constexpr int ROWS = decltype(rows_tag)::value;
    constexpr int NK = ROWS * G::WS, NP = (NK + 1) / 2;
    const int band = c % NB;
    const int R = min(RB, G::HS - band * RB);
#pragma unroll
    for (int g = 0; g < GI; ++g) {
      const float* ar = Al + g * BM * AP;
      const float* br = Bl + g * CBT * BP;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const int k0 = 2 * pp;
        const int off0 = 2 * (k0 / G::WS) * BRP + 2 * (k0 % G::WS);
        const bool straddle = (k0 % G::WS) == G::WS - 1;  // the odd k of the pair is the next row's first pixel
        const bool half = k0 + 1 >= NK;                    // ... or does not exist (odd pixel count): zero both sides
        float av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          av[i] = ar[abase[i] + k0];
          if (half) av[i] = lh ? 0.f : av[i];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bv[j] = (straddle && !half) ? br[bbase_w[j] + off0] : br[bbase[j] + off0];
          // a phantom k reads past the band's copy (uninitialised LDS): 0 * NaN would poison the sum
          if (half) bv[j] = lh ? 0.f : bv[j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        // next chunk's loads, spread over the first 1/SPREAD of the pairs (pinned: the scheduler otherwise
        // sinks them all to the end of the loop, where they have no time left to land before the LDS stores)
        constexpr int NSLOT = GI * NP;
        constexpr int SSLOT = (NSLOT + cmax(T::SPREAD, 1) - 1) / cmax(T::SPREAD, 1);
        const int slot = g * NP + pp;
        if (T::SPREAD && slot < SSLOT && slot * NLD / SSLOT < (slot + 1) * NLD / SSLOT) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = slot * NLD / SSLOT; q < (slot + 1) * NLD / SSLOT; ++q) gload_piece(q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (p.want_db && bx == 0) {  // bias gradient of `small`: row sums of the A chunk
      constexpr int PARTS = NT / BM;
      const int m = tid % BM, part = tid / BM;
      if (part < PARTS) {
        for (int g = 0; g < GI; ++g)
          for (int e = part; e < R * G::WS; e += PARTS) dbsum += Al[(g * BM + m) * AP + e];
      }
    }
  };

  REPO_STAMP_DECL
  if (nch > 0) {
    gload(0);
    REPO_STAMP(5);
    for (int c = 0; c < nch; ++c) {
      lstore();
      REPO_STAMP(2);
      __syncthreads();
      REPO_STAMP(3);
      gload_begin(min(c + 1, nch - 1));
      if (!T::SPREAD) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) gload_piece(q);
        __builtin_amdgcn_sched_barrier(0);
      }
      REPO_STAMP(0);
      if (RL == RB || c % NB != NB - 1) compute(std::integral_constant<int, RB>{}, c);
      else compute(std::integral_constant<int, RL>{}, c);
      REPO_STAMP(1);
      __syncthreads();
      REPO_STAMP(3);
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
    float* red = lds;  // K loop is over (trailing barrier above)
    red[tid] = dbsum;
    __syncthreads();
    if (tid < BM && m0 + tid < G::CS) {
      float s = 0.f;
      for (int q = 0; q < PARTS; ++q) s += red[q * BM + tid];
      p.slab[((size_t)z * G::CS + m0 + tid) * LDS_ + NW] = s;
    }
  }
  REPO_STAMP_FLUSH(nch);
}

template <class G, class BigT, class T>
inline int launch_dconv_wgrad(const WgradArgs& a, int splits, hipStream_t s) {
  constexpr int NW = G::CB * G::KK;
  dim3 grid((NW + T::BN - 1) / T::BN, (G::CS + T::BM - 1) / T::BM, (unsigned)splits);
  hipLaunchKernelGGL((dconv_wgrad_kernel<G, BigT, T>), grid, dim3(T::NT), 0, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// --------------------------------------------------------------------- decoder output layer + pixel NLL
// The last transposed convolution has 3 output channels: on 32x32 MFMA tiles nine tenths of the M rows
// are padding.  Here M = (py, cb, px) = 12 rows sits on v_mfma_f32_16x16x4_f32 (16 rows, K = 4 input
// channels of one tap per instruction), N = 16 class pixels.  All 32x36x3 weights live in LDS for the
// whole kernel in FRAGMENT-READY order (one conflict-free ds_read_b32 per MFMA, rows 12..15 zero), the
// workgroup is persistent over (image, 8 class rows) tiles, and the epilogue is the unit-variance pixel
// likelihood: d = conv + bias - target, loss += d^2/2, dpre = d * grad_scale (8-byte stores: the two
// px classes of a pixel are registers 2q, 2q+1 of the same lane).
struct NllArgs {
  const float* h3;
  const float* w;
  const float* bias;
  const void* target;
  float* recon;     // nullable
  float* dpre;      // nullable
  unsigned char* mask4;  // nullable: quad mask of h3 (REPO_EPI_MUL_MASK4)
  float* partials;  // one per workgroup
  float grad_scale;
  int nimg;
  unsigned h3_bytes;
  float* chan_partials = nullptr;  // bdec4.h, nullable: [3][workgroups] sums of (recon - target) per channel
};

typedef float f32x4acc __attribute__((ext_vector_type(4)));

template <class TgtT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void dconv_dec4_nll_kernel(NllArgs p) {
  constexpr int CS = 32, HS = 30, PS = 900, HB = 64, PP = 34, PLANE = 400, ROWS = 10;
  constexpr int NVEC = CS * 75, P_PER = (NVEC + 255) / 256;  // 300 floats (10 rows x 30) per channel
  __shared__ __attribute__((aligned(16))) float lds[9 * 8 * 64 + CS * PLANE];
  __shared__ float red[16];
  float* Wf = lds;
  float* Pl = lds + 9 * 8 * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int ntiles = p.nimg * 4;

  // ---- fragment-ready weights: Wf[(tap*8 + cq)*64 + lane] = w[c = 4cq + lg][cb][py + 2ty][px + 2tx]
  //      for m = lj = py*6 + cb*2 + px < 12 (tap = ty_*3 + tx_, ty = 2 - ty_, tx = 2 - tx_), else 0
  for (int i = tid; i < 9 * 8 * 64; i += 256) {
    const int l = i & 63, cq = (i >> 6) & 7, tap = i >> 9;
    const int m = l & 15, c = 4 * cq + (l >> 4);
    float v = 0.f;
    if (m < 12) {
      const int py = m / 6, cb = (m % 6) >> 1, px = m & 1;
      const int ky = py + 2 * (2 - tap / 3), kx = px + 2 * (2 - tap % 3);
      v = p.w[((c * 3 + cb) * 6 + ky) * 6 + kx];
    }
    Wf[i] = v;
  }
  for (int i = tid; i < CS * PLANE / 4; i += 256) reinterpret_cast<f32x4*>(Pl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- patch staging roles (identical for every tile)
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(p.h3, p.h3_bytes);
  int pgo[P_PER];
  int ppk[P_PER];  // LDS address of element 0 | its column << 16 (elements 1..3 follow, +4 across a row end)
  unsigned long long mtop = 0ull, mbot = 0ull;  // element (j,e) lies in patch rows 0..1 / 8..9
#pragma unroll
  for (int j = 0; j < P_PER; ++j) {
    const int v = tid + j * 256, c = v / 75, e4 = v % 75;
    const bool act = v < NVEC;
    pgo[j] = act ? c * PS + e4 * 4 : -1;
    const int q0 = e4 * 4;
    ppk[j] = (c * PLANE + (q0 / 30) * PP + q0 % 30 + 2) | ((q0 % 30) << 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = (q0 + e) / 30;
      if (r < 2) mtop |= 1ull << (j * 4 + e);
      if (r >= 8) mbot |= 1ull << (j * 4 + e);
    }
  }
  // ---- per-lane B bases: wave wv owns class rows 2wv, 2wv+1 of the tile, two 16-pixel halves each
  int bbase[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) bbase[t] = lg * PLANE + (2 * wv + (t >> 1)) * PP + 16 * (t & 1) + lj;

  f32x4 rpv[P_PER];
  auto gload = [&](int tile) __attribute__((always_inline)) {
    const int img = tile >> 2, cy0 = (tile & 3) * 8;
    const int bias_ = img * CS * PS + (cy0 - 2) * HS;
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
      rpv[j] = VecLoad<4>::load(rh, pgo[j] >= 0 ? 4u * (unsigned)(pgo[j] + bias_) : kOobOffset);
  };
  auto lstore = [&](int tile) __attribute__((always_inline)) {
    const int rg = tile & 3;
    const unsigned long long bad = (rg == 0 ? mtop : 0ull) | (rg == 3 ? mbot : 0ull);
    // quad mask of h3: this tile OWNS strip rows 2..9 (its 8 class rows; 6 in an image's last quarter) = the
    // 16-byte vectors 15..74 (..59) of a channel's strip, each exactly one aligned quad of the flat tensor
    const int e4_end = rg == 3 ? 60 : 75;
    const long mbase = (long)(tile >> 2) * CS * PS + ((tile & 3) * 8 - 2) * HS;  // flat index of the strip's element 0
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
      if (pgo[j] >= 0) {
        if (p.mask4) {
          const int v = tid + j * 256, e4 = v % 75;
          if (e4 >= 15 && e4 < e4_end) {
            const unsigned nib = (rpv[j][0] > 0.f ? 1u : 0u) | (rpv[j][1] > 0.f ? 2u : 0u) | (rpv[j][2] > 0.f ? 4u : 0u) |
                                 (rpv[j][3] > 0.f ? 8u : 0u);
            p.mask4[(mbase + pgo[j]) >> 2] = (unsigned char)nib;
          }
        }
        const int base = ppk[j] & 0xffff, col0 = ppk[j] >> 16;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          Pl[base + e + (col0 + e >= 30 ? PP - 30 : 0)] = ((bad >> (j * 4 + e)) & 1ull) ? 0.f : rpv[j][e];
      }
  };

  float lsum = 0.f;
  const int G = gridDim.x;
  int tile = blockIdx.x;
  if (tile < ntiles) gload(tile);
  __syncthreads();  // weights + zero fill visible
  for (; tile < ntiles; tile += G) {
    lstore(tile);
    __syncthreads();
    if (tile + G < ntiles) gload(tile + G);
    __builtin_amdgcn_sched_barrier(0);

    f32x4acc acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int cq = 0; cq < 8; ++cq) {
        const float a = Wf[(tap * 8 + cq) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float b = Pl[bbase[t] + 4 * cq * PLANE + (tap / 3) * PP + tap % 3];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
      }

    // ---- epilogue.  A lane holds, for each of its 4 pixel tiles, two (row parity, channel) pairs q = 2 lg + pr of the
    // pixel pair (2cx, 2cx+1).  Neighbouring lanes (cx, cx + 1) swap one pair each, so that the even lane owns FOUR
    // consecutive pixels of pair q0 and the odd lane four of pair q1: one 16-byte store of d recon and one dword of
    // uint8 targets per lane and tile instead of two 8-byte stores and two 2-byte loads (the epilogue was 22 % of
    // the kernel: ablation, round 3).
    const int img = tile >> 2, cy0 = (tile & 3) * 8;
    if (lg < 3) {
      const int odd = lj & 1;
      const int q = 2 * lg + odd, py = q / 3, cb = q % 3;  // the pair this lane ends up with
      const float bv = p.bias ? p.bias[cb] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // keep pair `odd`, send pair `1 - odd`; the partner (lane ^ 1) does the opposite
        const float k0 = odd ? acc[t][2] : acc[t][0], k1 = odd ? acc[t][3] : acc[t][1];
        const float s0 = odd ? acc[t][0] : acc[t][2], s1 = odd ? acc[t][1] : acc[t][3];
        const float r0 = __shfl_xor(s0, 1, 64), r1 = __shfl_xor(s1, 1, 64);
        // four consecutive pixels starting at x0 = 4 * (cx >> 1): the even lane's own pair comes first
        const float v0 = (odd ? r0 : k0) + bv, v1 = (odd ? r1 : k1) + bv, v2 = (odd ? k0 : r0) + bv, v3 = (odd ? k1 : r1) + bv;
        const int cy = cy0 + 2 * wv + (t >> 1), cx = 16 * (t & 1) + lj;
        const int o = ((img * 3 + cb) * HB + 2 * cy + py) * HB + 4 * (cx >> 1);
        float t0, t1, t2, t3;
        if (sizeof(TgtT) == 1) {
          const unsigned tw = *reinterpret_cast<const unsigned*>((const uint8_t*)p.target + o);
          t0 = pix_norm((uint8_t)(tw & 0xff));
          t1 = pix_norm((uint8_t)((tw >> 8) & 0xff));
          t2 = pix_norm((uint8_t)((tw >> 16) & 0xff));
          t3 = pix_norm((uint8_t)(tw >> 24));
        } else {
          const float4 tw = *reinterpret_cast<const float4*>((const float*)p.target + o);
          t0 = tw.x, t1 = tw.y, t2 = tw.z, t3 = tw.w;
        }
        const float d0 = v0 - t0, d1 = v1 - t1, d2 = v2 - t2, d3 = v3 - t3;
        lsum += 0.5f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        if (p.dpre)
          *reinterpret_cast<float4*>(p.dpre + o) = make_float4(d0 * p.grad_scale, d1 * p.grad_scale, d2 * p.grad_scale, d3 * p.grad_scale);
        if (p.recon) *reinterpret_cast<float4*>(p.recon + o) = make_float4(v0, v1, v2, v3);
      }
    }
    __syncthreads();  // patch reads done before the next tile overwrites it
  }
  const float s = block_sum(lsum, red);
  if (tid == 0) p.partials[blockIdx.x] = s;
}

inline int dec4_nll_grid(long nimg) {
  const long ntiles = nimg * 4;
  return (int)(ntiles < 1024 ? ntiles : 1024);
}

}  // namespace repo
