// Stride-2 convolution family of the reference's encoder/decoder on the igemm tile engine.
//
// A layer is a (big, small) pair with big = 2*small + KS - 2 and one weight tensor
// w[cs][cb][ky][kx] (nn.Conv2d (out,in,kh,kw) for the encoder, nn.ConvTranspose2d
// (in,out,kh,kw) for the decoder -- the same indexing).  GEMM roles put channels on M and
// pixels on N so the epilogue's lanes run along the contiguous pixel index of NCHW.
//
//   down : small[img][cs][sy][sx] = sum_{cb,ky,kx} big[img][cb][2sy+ky][2sx+kx] w[cs][cb][ky][kx]
//          M = CS, N = nimg*HS*WS, K = CB*KS*KS                 (A = weights, k-contiguous)
//   up   : for each output parity class (py,px):
//          big[img][cb][2y+py][2x+px] = sum_{cs,jy,jx} small[img][cs][y-jy][x-jx] w[cs][cb][py+2jy][px+2jx]
//          M = CB, N = nimg*ny*nx, K = CS*JY*JX  -- only the taps that exist, no zero-stuffing
//   upm  : the four classes merged on M (M = 4*CB) for CB = 3, where a 32-row tile per class
//          would waste 29 rows; the taps (3x3 for k=6) are shared by all four classes.
//   wgrad: dw[cs][cb][ky][kx] = sum_{img,sy,sx} small[img][cs][sy][sx] big[img][cb][2sy+ky][2sx+kx]
//          M = CS, N = CB*KS*KS (+1 column of ones = bias gradient of `small`), K = nimg*HS*WS
//          split over images into slabs, reduced in fixed order.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "igemm.h"
#include "dconv.h"
#include "uconv.h"
#include "dconv_up.h"
#include "bconv.h"
#include "buconv.h"
#include "bwgrad.h"
#include "bdec4.h"
#include "twgrad.h"
#include "tconv_up.h"
#include "tconv_down.h"


namespace repo {

// Test aid (repo_debug_bconv): 0 keeps every conv layer on the fp32-MFMA kernels.  Thread-local (api.hip): the setting
// of the calling host thread, read at launch time.
static thread_local int t_bconv_enabled = 1;

template <int CB_, int CS_, int HB_, int KS_>
struct Geo {
  static constexpr int CB = CB_, CS = CS_, HB = HB_, WB = HB_, KS = KS_;
  static constexpr int HS = (HB - KS) / 2 + 1, WS = HS;
  static constexpr int PB = HB * WB, PS = HS * WS, KK = KS * KS;
};
using GEnc1 = Geo<3, 32, 64, 4>;
using GEnc2 = Geo<32, 64, 31, 4>;
using GEnc3 = Geo<64, 128, 14, 4>;
using GEnc4 = Geo<128, 256, 6, 4>;
using GDec2 = Geo<64, 128, 13, 5>;
using GDec3 = Geo<32, 64, 30, 6>;
using GDec4 = Geo<3, 32, 64, 6>;
static_assert(GEnc1::HS == 31 && GEnc2::HS == 14 && GEnc3::HS == 6 && GEnc4::HS == 2, "encoder geometry");
static_assert(GDec2::HS == 5 && GDec3::HS == 13 && GDec4::HS == 30, "decoder geometry");
// The 128 x 128 stack (BASELINE config 4's frame size; build-defined: the reference's encoder hard-codes the 64 x 64
// flatten, encoder.py:39).  Same kernel sizes and strides, one more decoder layer (layers 7..12 of the ABI):
//   encoder  3x128x128 -> 32x63x63 -> 64x30x30 -> 128x14x14 -> 256x6x6 (k4), then a build-defined fc 9216 -> 1024
//   decoder  ... -> 32x30x30 (layers 4, 5 as at 64 x 64) -> 16x64x64 (k6) -> 3x128x128 (k2)
using GX1 = Geo<3, 32, 128, 4>;
using GX2 = Geo<32, 64, 63, 4>;
using GX3 = Geo<64, 128, 30, 4>;
using GX4 = Geo<128, 256, 14, 4>;
using GY4 = Geo<16, 32, 64, 6>;
using GY5 = Geo<3, 16, 128, 2>;
using GT4 = Geo<6, 32, 64, 6>;  // TIAObservationModel.conv4 (models/decoder.py:165): 32 -> 6 = [recon | mask]
static_assert(GX1::HS == 63 && GX2::HS == 30 && GX3::HS == 14 && GX4::HS == 6 && GY4::HS == 30 && GY5::HS == 64,
              "128 x 128 geometry");

__device__ __forceinline__ float epi_apply(float v, int epi, const float* bias, int ch, const float* aux, int o) {
  if (bias) v += bias[ch];
  if (epi == REPO_EPI_RELU) v = fmaxf(v, 0.f);
  else if (epi == REPO_EPI_MUL_DRELU) v = aux[o] > 0.f ? v : 0.f;
  return v;
}

// Compile-time k -> address-offset tables (one per geometry), read with scalar loads: inside the K
// loop k is wave-uniform for the n-major operands, so the (channel, ky, kx) decode -- ~10 scalar ALU
// instructions per element when done with div/mod by constants -- becomes one s_load per element.
template <class G, int JY, int JX>
struct UpKTab {
  int off[G::CS * JY * JX];
  int sh[G::CS * JY * JX];
  constexpr UpKTab() : off(), sh() {
    for (int k = 0; k < G::CS * JY * JX; ++k) {
      const int cs = k / (JY * JX), r = k % (JY * JX);
      const int jy = r / JX, jx = r % JX;
      off[k] = cs * G::PS - jy * G::WS - jx;
      sh[k] = jy | ((4 + jx) << 8);
    }
  }
};
template <class G, int JY, int JX>
__device__ const UpKTab<G, JY, JX> g_up_ktab{};

// ------------------------------------------------------------------------------- down
template <class G, class TgtT, int MODE>
struct ConvUpMergedOp {
  static constexpr bool A_KMAJOR = true, B_KMAJOR = false;
  static constexpr int NY = (G::HB + 1) / 2, NX = (G::WB + 1) / 2;
  static constexpr int J = (G::KS + 1) / 2, JJ = J * J;
  const float* small;
  const float* w;
  const float* bias;
  const float* aux;
  float* out;  // MODE 0: output; MODE 1: recon (nullable)
  int nimg, epi;
  // MODE 1
  const TgtT* target;
  float* dpre;       // nullable
  float* partials;   // one per workgroup
  float grad_scale;
  float lsum;

  struct AM {
    int off;
    int py, px;
  };
  struct AK {
    int off;
    int jy2, jx2;  // 2*jy, 2*jx
  };
  struct BN {
    int off;
    unsigned mask;
  };
  struct BK {
    int off;
    int sh;
  };
  __device__ void init(int) { lsum = 0.f; }
  __device__ int M() const { return 4 * G::CB; }
  __device__ int N() const { return nimg * NY * NX; }
  __device__ int kbeg() const { return 0; }
  __device__ int kend() const { return G::CS * JJ; }
  // M index = (py, cb, px) with px fastest: two consecutive accumulator registers of a lane are the
  // two horizontally adjacent output pixels, so the epilogue stores them as one 8-byte word and a
  // wave writes whole 256-byte runs (the separate px classes wrote every line twice, half each:
  // WRITE_SIZE was 2.0x the tensor).
  __device__ AM a_m(int m) const {
    const int py = m / (2 * G::CB), rem = m % (2 * G::CB);
    const int cb = rem >> 1, px = rem & 1;
    return AM{cb * G::KK + py * G::KS + px, py, px};
  }
  __device__ AK a_k(int k) const {
    const int cs = k / JJ, r = k % JJ;
    const int jy2 = 2 * (r / J), jx2 = 2 * (r % J);
    return AK{cs * (G::CB * G::KK) + jy2 * G::KS + jx2, jy2, jx2};
  }
  __device__ float a(const AM& m, const AK& k) const {
    const bool ok = (m.py + k.jy2 < G::KS) && (m.px + k.jx2 < G::KS);
    const float v = w[(unsigned)(ok ? m.off + k.off : 0)];
    return ok ? v : 0.f;
  }
  __device__ BN b_n(int n) const {
    const int img = n / (NY * NX), q = n % (NY * NX);
    const int y = q / NX, x = q % NX;
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mask |= ((y - j >= 0 && y - j < G::HS) ? 1u : 0u) << j;
      mask |= ((x - j >= 0 && x - j < G::WS) ? 1u : 0u) << (4 + j);
    }
    return BN{(img * G::CS * G::HS + y) * G::WS + x, mask};
  }
  __device__ BK b_k(int k) const { return BK{g_up_ktab<G, J, J>.off[k], g_up_ktab<G, J, J>.sh[k]}; }
  __device__ float b(const BK& k, const BN& n) const {
    const bool ok = ((n.mask >> (k.sh & 0xff)) & (n.mask >> (k.sh >> 8)) & 1u) != 0;
    const float v = small[(unsigned)(ok ? n.off + k.off : 0)];
    return ok ? v : 0.f;
  }
  __device__ void store_col(int mb, int n, const f32x16& acc, int M) {
    const int img = n / (NY * NX), q = n % (NY * NX);
    const int y2 = 2 * (q / NX), x2 = 2 * (q % NX);
    const int obase = img * G::CB * G::PB;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int m = mb + (r & 3) + 8 * (r >> 2);  // even: px = 0; register r+1 is px = 1
      if (m < M) {
        const int py = m / (2 * G::CB), cb = (m % (2 * G::CB)) >> 1;
        const int by = y2 + py;
        if (by < G::HB && x2 < G::WB) {
          const int o = obase + (cb * G::HB + by) * G::WB + x2;
          const bool two = x2 + 1 < G::WB;
          float v0 = acc[r], v1 = acc[r + 1];
          if (MODE == 0) {
            v0 = epi_apply(v0, epi, bias, cb, aux, o);
            if (two) v1 = epi_apply(v1, epi, bias, cb, aux, o + 1);
          } else {
            const float bv = bias ? bias[cb] : 0.f;
            v0 += bv;
            v1 += bv;
            const float d0 = v0 - load_as_float(target, (unsigned)o);
            const float d1 = two ? v1 - load_as_float(target, (unsigned)(o + 1)) : 0.f;
            lsum += 0.5f * (d0 * d0 + d1 * d1);
            if (dpre) {
              if (G::WB % 2 == 0) {
                *reinterpret_cast<float2*>(dpre + o) = make_float2(d0 * grad_scale, d1 * grad_scale);
              } else {
                dpre[o] = d0 * grad_scale;
                if (two) dpre[o + 1] = d1 * grad_scale;
              }
            }
          }
          if (out) {
            if (G::WB % 2 == 0) {
              *reinterpret_cast<float2*>(out + o) = make_float2(v0, v1);
            } else {
              out[o] = v0;
              if (two) out[o + 1] = v1;
            }
          }
        }
      }
    }
  }
  __device__ void finish() {
    if (MODE == 1) {
      __shared__ float red[16];
      const float s = block_sum(lsum, red);
      if (threadIdx.x == 0)
        partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
    }
  }
};

// ------------------------------------------------------------------------------- wgrad
__global__ void conv_slab_reduce_kernel(const float* __restrict__ slab, int splits, int Mrows, int Ncols,
                                        float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const int total = Mrows * (Ncols + 1);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int m = i / (Ncols + 1), n = i % (Ncols + 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // fixed-order independent chains
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
      s0 += slab[(size_t)z * total + i];
      s1 += slab[(size_t)(z + 1) * total + i];
      s2 += slab[(size_t)(z + 2) * total + i];
      s3 += slab[(size_t)(z + 3) * total + i];
    }
    for (; z < splits; ++z) s0 += slab[(size_t)z * total + i];
    const float s = (s0 + s1) + (s2 + s3);
    if (n < Ncols) {
      float* p = dw + (size_t)m * Ncols + n;
      *p = accumulate ? *p + s : s;
    } else if (db) {
      db[m] = accumulate ? db[m] + s : s;
    }
  }
}

// Many splits, few outputs (the 3-channel layers: 613 slabs of 32 x 49): a workgroup of 1024 threads owns 64 consecutive
// outputs; thread (o, g) adds slabs g, g + 16, ... of output o -- the 64 lanes of a wave read 256 contiguous bytes of
// one slab, four loads in flight per thread -- and the 16 partial sums of an output meet in LDS in a fixed order
// (bit-reproducible).  (Round 4's one-wave-per-output form read 64 different slabs per load instruction, one cache line
// each: 70-80 us inside the update for 3.8 MB of slabs.)
// SLAB_REDUCE_WAVES waves per workgroup walk the 16 slab groups (16: the round-4 form, one group per wave; 4: each wave takes
// four groups in turn -- the same 16 partial sums in the same order, bit for bit, from a 256-thread workgroup that finds a
// place on a busy chip; a 1024-thread workgroup needs a CU with 16 free wave slots, and inside the update these launches
// waited for one: 78-135 us each in the trace against ~15 us alone).
#ifndef SLAB_REDUCE_WAVES
#define SLAB_REDUCE_WAVES 4
#endif
__global__ __launch_bounds__(64 * SLAB_REDUCE_WAVES) void conv_slab_reduce_wave_kernel(const float* __restrict__ slab, int splits, int Mrows,
                                                                     int Ncols, float* __restrict__ dw,
                                                                     float* __restrict__ db, int accumulate, int tkk,
                                                                     int tcb, float* __restrict__ dbig) {
  // tkk > 0: the slabs are twgrad.h's [tap][row][channel] (+ db[row] at the end) instead of [row][channel * tkk + tap | db]
  // and, behind them, the two row-parity halves of the channel sums of `big` (2 x tcb: one aligned block of 64 outputs)
  __shared__ float red[16][64];
  const int total = Mrows * (Ncols + 1) + (tkk > 0 ? 2 * tcb : 0);
  const int o = threadIdx.x & 63, gw = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  for (int g = gw; g < 16; g += SLAB_REDUCE_WAVES) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
      int z = g;
      for (; z + 48 < splits; z += 64) {
        s0 += slab[(size_t)z * total + i];
        s1 += slab[(size_t)(z + 16) * total + i];
        s2 += slab[(size_t)(z + 32) * total + i];
        s3 += slab[(size_t)(z + 48) * total + i];
      }
      for (; z < splits; z += 16) s0 += slab[(size_t)z * total + i];
    }
    red[g][o] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  const int g = gw;
  if (g == 0 && i < total) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][o];
    int m = i / (Ncols + 1), n = i % (Ncols + 1);
    if (tkk > 0) {
      const int body = Mrows * Ncols;
      if (i >= body + Mrows) {   // channel sums of `big`: output c = half 0 + half 1 (both in this block's `red`)
        const int c = i - body - Mrows;
        if (dbig && c < tcb) {
          float s2 = 0.f;
#pragma unroll
          for (int q = 0; q < 16; ++q) s2 += red[q][o + tcb];
          dbig[c] = accumulate ? dbig[c] + (s + s2) : s + s2;
        }
        return;
      }
      if (i < body) m = (i / tcb) % Mrows, n = (i % tcb) * tkk + i / (tcb * Mrows);
      else m = i - body, n = Ncols;
    }
    if (n < Ncols) {
      float* p = dw + (size_t)m * Ncols + n;
      *p = accumulate ? *p + s : s;
    } else if (db) {
      db[m] = accumulate ? db[m] + s : s;
    }
  }
}

// out[c] (+)= scale * sum of parts[c][0 .. n): one workgroup per c
__global__ void partial_sum_rows_kernel(const float* __restrict__ parts, int n, float* __restrict__ out, float scale,
                                        int accumulate) {
  __shared__ float red[16];
  const float* row = parts + (size_t)blockIdx.x * n;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += row[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = accumulate ? out[blockIdx.x] + scale * s : scale * s;
}

__global__ void partial_sum_kernel(const float* __restrict__ parts, int n, float* __restrict__ out, int accumulate) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += parts[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out = accumulate ? *out + s : s;
}

// x[n][c][p] -> partial[s][c] over image chunk s.  Threads walk the flattened (image, pixel) index of
// the chunk with an incremental (img, p) carry, so small planes (P = 25, 169) keep all lanes busy.
__global__ void channel_sum_kernel(const float* __restrict__ x, int nimg, int C, int P, int imgs_per_split,
                                   float* __restrict__ parts) {
  __shared__ float red[16];
  const int c = blockIdx.x, s = blockIdx.y;
  const int i0 = s * imgs_per_split, i1 = min(nimg, i0 + imgs_per_split);
  const int nt = blockDim.x;
  const int dq = nt / P, dr = nt % P;
  int img = i0 + (int)threadIdx.x / P, p = (int)threadIdx.x % P;
  float a0 = 0.f, a1 = 0.f;
  const size_t cstride = (size_t)C * P;
  const float* base = x + (size_t)c * P;
  while (img < i1) {
    a0 += base[img * cstride + p];
    p += dr;
    img += dq;
    if (p >= P) {
      p -= P;
      ++img;
    }
    if (img >= i1) break;
    a1 += base[img * cstride + p];
    p += dr;
    img += dq;
    if (p >= P) {
      p -= P;
      ++img;
    }
  }
  const float acc = block_sum(a0 + a1, red);
  if (threadIdx.x == 0) parts[s * C + c] = acc;
}
// one wave per channel: lanes stride over the splits, fixed-order shuffle reduction (bit-reproducible)
__global__ void channel_sum_final_kernel(const float* __restrict__ parts, int splits, int C, float* __restrict__ out,
                                         int accumulate) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  float s = 0.f;
  for (int z = lane; z < splits; z += 64) s += parts[z * C + c];
  s = wave_sum(s);
  if (lane == 0) out[c] = accumulate ? out[c] + s : s;
}

__global__ void relu_mask_kernel(int64_t n, const float* __restrict__ dy, const float* __restrict__ h,
                                 float* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = h[i] > 0.f ? dy[i] : 0.f;
}

// ------------------------------------------------------------------------------- host-side dispatch
// direct-conv tiles per layer: <BM, BN, CK, WM, WN>
template <class G>
struct DTileFor;
// wgrad tiles: <BM, BN, WM, WN, images per chunk, small rows per chunk>; WGT = workgroups the split-K over image
// groups aims at (sweep of 768 / 1024 / 1536 / 2048 / 3072 on one box, round 3: enc2 371 -> 335 us at 768, enc3
// 239 -> 226 at 1024, dec3 542 -> 531 at 2048; fewer splits also mean smaller slabs to reduce)
// the 3-channel layers are one channel chunk per workgroup (no pipelining inside it): 8 waves per tile and short
// weight-gradient bands measured best (tile sweep, round 2: enc1 fwd 148 -> 140 us, enc1 wgrad 224 -> 171,
// dec4 dgrad 311-338 -> 247, dec4 wgrad 210 -> 201)
template <> struct DTileFor<GEnc1> { using Down = DTile<32, 512, 3, 1, 8>; using Wgrad = WTile<32, 64, 1, 2, 1, 4, 2>; static constexpr int WGT = 3072; };
template <> struct DTileFor<GEnc2> { using Down = DTile<64, 128, 2, 2, 2>; using Wgrad = WTile<64, 128, 2, 2, 1, 7, 2>; static constexpr int WGT = 768; };
template <> struct DTileFor<GEnc3> { using Down = DTile<128, 128, 2, 2, 2>; using Wgrad = WTile<64, 128, 2, 2, 2, 6>; static constexpr int WGT = 1024; };
// enc4 forward has only 9800 output pixels: 128 x 128 tiles are 154 workgroups on 256 CUs (171 us); 32 x 64: 125 us
template <> struct DTileFor<GEnc4> { using Down = DTile<32, 64, 2, 1, 2>; using Wgrad = WTile<64, 128, 2, 2, 8, 2>; static constexpr int WGT = 768; };
// (dec2 down: 8 waves, 306 -> 274 us, A/B on one box)
template <> struct DTileFor<GDec2> { using Down = DTile<128, 128, 4, 2, 4, 1>; using Wgrad = WTile<64, 128, 2, 2, 4, 5>; static constexpr int WGT = 1536; };
template <> struct DTileFor<GDec3> { using Down = DTile<64, 128, 2, 2, 2>; using Wgrad = WTile<64, 128, 2, 2, 1, 7>; static constexpr int WGT = 2048; };
template <> struct DTileFor<GDec4> { using Down = DTile<32, 256, 3, 1, 8>; using Wgrad = WTile<32, 128, 1, 4, 1, 2>; static constexpr int WGT = 1536; };
// 128 x 128 stack: tiles by analogy with the 64 x 64 layer of the same role (not swept)
// (GX1's weight gradient walks 2-row bands: with 4 rows of 63 pixels the 126 k-pairs of a band exceed what the
// compiler unrolls, the chunk-ahead loads then index their registers at run time: 1935 us instead of ~400)
template <> struct DTileFor<GX1> { using Down = DTile<32, 512, 3, 1, 8>;  using Wgrad = WTile<32, 64, 1, 2, 1, 2, 2>; static constexpr int WGT = 3072; };
template <> struct DTileFor<GX2> { using Down = DTile<64, 128, 2, 2, 2>;  using Wgrad = WTile<64, 128, 2, 2, 1, 3, 2>; static constexpr int WGT = 1536; };
template <> struct DTileFor<GX3> { using Down = DTile<128, 128, 2, 2, 2>; using Wgrad = WTile<64, 128, 2, 2, 1, 7>; static constexpr int WGT = 1024; };
template <> struct DTileFor<GX4> { using Down = DTile<64, 128, 2, 2, 2>;  using Wgrad = WTile<64, 128, 2, 2, 2, 6>; static constexpr int WGT = 1024; };
template <> struct DTileFor<GY4> { using Down = DTile<32, 256, 2, 1, 4>;  using Wgrad = WTile<32, 128, 1, 4, 1, 2>; static constexpr int WGT = 1536; };
template <> struct DTileFor<GT4> { using Down = DTile<32, 256, 3, 1, 8>;  using Wgrad = WTile<32, 128, 1, 4, 1, 2>; static constexpr int WGT = 1536; };
template <> struct DTileFor<GY5> { using Down = DTile<32, 512, 3, 1, 8>;  using Wgrad = WTile<32, 64, 1, 2, 1, 2>; static constexpr int WGT = 1536; };

// A handful of frames (the acting path encodes ONE per environment step): the throughput tiles leave 1-2
// workgroups walking 16-64 dependent channel chunks (enc4: 147 us for one frame).  Latency tiles use 32
// output channels per workgroup (4-8x more workgroups) and 4x larger channel chunks (4x fewer barriers).
template <class G> struct DLatTile { using type = typename DTileFor<G>::Down; };
template <> struct DLatTile<GEnc2> { using type = DTile<32, 128, 4, 1, 4>; };
template <> struct DLatTile<GEnc3> { using type = DTile<32, 128, 8, 1, 4>; };
template <> struct DLatTile<GEnc4> { using type = DTile<32, 128, 8, 1, 4>; };

// The bf16x6 down kernel (bconv.h: fp32-accurate, six bf16 MFMAs per 16 k) for the MFMA-bound geometries with an even
// big-row pitch; NoBTile = the layer stays on the fp32-MFMA kernel (3-channel layers: bandwidth / epilogue bound).  Test aid repo_debug_bconv(0) keeps every layer on the fp32 kernel.
struct NoBTile {};
template <class G> struct BDownFor { using type = NoBTile; };
template <> struct BDownFor<GDec3> { using type = BTile<64, 256, 2, 1, 4>; };
// enc3 / enc4: ONE M tile per workgroup (the patch is staged and split once): 169 -> 146 us, 112 -> 101 (round 5)
template <> struct BDownFor<GEnc3> { using type = BTile<128, 128, 4, 2, 2>; };
template <> struct BDownFor<GEnc4> { using type = BTile<128, 64, 4, 2, 2>; };
// enc2 forward (31 x 31 planes, k4): with the element-wise staging of its padded pitch it measured equal on both kernels
// (312 vs 309 us, round 4); staged by LDS quads (bconv.h, QROW) it is on the bf16 pipe
template <> struct BDownFor<GEnc2> { using type = BTile<64, 256, 4, 1, 4>; };
template <> struct BDownFor<GDec2> { using type = BTile<64, 128, 2, 1, 4>; };   // 13 x 13 planes, k5 (32 slots for 25 taps)
template <class G> constexpr bool kBDown = !std::is_same<typename BDownFor<G>::type, NoBTile>::value;
// ... and for uint8 frames (the encoder's first layer as train_agent() feeds it): the bytes are exact in ONE bf16, so the
// product needs three MFMAs per block instead of six and no split of the activation (bconv.h, U8); float frames of the
// same layer stay on the fp32 kernel.
template <class G> struct BDownU8For { using type = NoBTile; };
template <> struct BDownU8For<GEnc1> { using type = BTile<32, 512, 4, 1, 8>; };
template <class G> constexpr bool kBDownU8 = !std::is_same<typename BDownU8For<G>::type, NoBTile>::value;
template <class G, class BigT>
constexpr bool kBDownT = std::is_same<BigT, float>::value ? kBDown<G> : kBDownU8<G>;
template <class G, class BigT>
using BDownTile = typename std::conditional<std::is_same<BigT, float>::value, typename BDownFor<G>::type, typename BDownU8For<G>::type>::type;

template <class G, class BigT = float>
static bool bconv_down_on(int64_t nimg) {
  if constexpr (kBDownT<G, BigT>) return t_bconv_enabled && nimg * (int64_t)G::PS > 512;
  return false;
}
template <class G, class BigT = float>
static size_t bconv_pack_bytes() {
  if constexpr (kBDownT<G, BigT>) {
    typedef BPack<G, BDownTile<G, BigT>> P;
    return ((std::is_same<BigT, float>::value ? P::BYTES : P::BYTES_U8) + 255) & ~(size_t)255;
  }
  return 0;
}

// pixel tiles of the direct conv's grid = rows of the channel-sum partials (repo_conv_down's dbias)
template <class G, class BigT = float>
static long conv_down_tiles(int64_t nimg) {
  const long px = nimg * (long)G::PS;
  long bn = px <= 512 ? DLatTile<G>::type::BN : DTileFor<G>::Down::BN;
  if constexpr (kBDownT<G, BigT>)
    if (bconv_down_on<G, BigT>(nimg)) bn = BDownTile<G, BigT>::BN;
  return (px + bn - 1) / bn;
}

template <class G, class BigT>
static int conv_down_t(int64_t nimg, const BigT* big, const float* w, const float* bias, float* small, int epi,
                       const float* aux, float* dbias, int accumulate_dbias, unsigned char* cmask, void* ws,
                       size_t ws_bytes, hipStream_t s) {
  if (cmask && G::CS % 4 != 0) return REPO_E_BADARG;
  if (nimg * (int64_t)G::CB * G::PB >= kMaxBufElems || nimg * (int64_t)G::CS * G::PS >= kMaxBufElems) return REPO_E_SHAPE;
  // workspace: [weight pack of the bf16x6 kernel | channel-sum partials]; without room for the pack the layer runs
  // on the fp32-MFMA kernel (ws stays optional for callers that want no dbias)
  bool bf = false;
  size_t pack_bytes = 0;
  if constexpr (kBDownT<G, BigT>) {
    pack_bytes = bconv_pack_bytes<G, BigT>();
    bf = bconv_down_on<G, BigT>(nimg) && ws && ws_bytes >= pack_bytes + (dbias ? (size_t)conv_down_tiles<G, BigT>(nimg) * G::CS * sizeof(float) : 0);
    if (!bf) pack_bytes = 0;
  }
  long tiles = conv_down_tiles<G, BigT>(nimg);
  if constexpr (kBDownT<G, BigT>)
    if (!bf && bconv_down_on<G, BigT>(nimg)) {   // the pack did not fit: the fp32 kernel's grid
      const long px = nimg * (long)G::PS;
      tiles = (px + DTileFor<G>::Down::BN - 1) / DTileFor<G>::Down::BN;
    }
  if (dbias && (!ws || ws_bytes < pack_bytes + (size_t)tiles * G::CS * sizeof(float))) return REPO_E_WS_TOO_SMALL;
  float* parts = dbias ? (float*)((char*)ws + pack_bytes) : nullptr;
  DownArgs a{big, w, bias, aux, small, (int)nimg, epi, (unsigned)(nimg * G::CB * G::PB * sizeof(BigT)),
             (unsigned)(G::CS * G::CB * G::KK * sizeof(float)), parts, cmask};
  int rc;
#ifndef TCD_DISABLE
  // staging waves beside multiplying waves (tconv_down.h): decoder conv3's data gradient; the kernel's pack takes the bf16x6
  // kernel's place in the workspace.  Encoder conv2's forward runs on the same kernel since round 6 (242 -> 203 us alone,
  // results within 3e-6 of fp64, masks identical to the fp32 engine's on random data; -DTCD_NO_ENC2 = bconv_down, for A/B).
  // Round 5 built it and left it un-routed: on the TIA oracle test's frames its 5e-7 differences flip ONE ReLU decision of
  // conv3 at a pre-activation 5e-7 from zero, and that pixel alone takes the encoder's gradient 5e-3 from the oracle's --
  // the test now hands the oracle the kernels' decision inside a 2e-6 band around zero (tests/test_tia_gpu.py).
  if constexpr (std::is_same<G, GDec3>::value && std::is_same<BigT, float>::value) {
    if (bf && (epi == REPO_EPI_NONE || epi == REPO_EPI_MUL_DRELU) && !bias && !dbias && !cmask && nimg >= 32 &&
        pack_bytes >= TcdGeo::PACK_BYTES)
      return launch_tconv_down<TcdGeo>(a, w, (char*)ws, s);
  }
#ifndef TCD_NO_ENC2
  if constexpr (std::is_same<G, GEnc2>::value && std::is_same<BigT, float>::value) {
    if (bf && epi == REPO_EPI_RELU && !dbias && nimg >= 32 && pack_bytes >= TcdGeoE2::PACK_BYTES)
      return launch_tconv_down<TcdGeoE2>(a, w, (char*)ws, s);
  }
#endif
#endif
  if constexpr (kBDownT<G, BigT>) {
    if (bf) rc = launch_bconv_down<G, BDownTile<G, BigT>, BigT>(a, w, (char*)ws, s);
    else rc = (nimg * (int64_t)G::PS <= 512) ? launch_dconv_down<G, BigT, typename DLatTile<G>::type>(a, s)
                                            : launch_dconv_down<G, BigT, typename DTileFor<G>::Down>(a, s);
  } else {
    rc = (nimg * (int64_t)G::PS <= 512) ? launch_dconv_down<G, BigT, typename DLatTile<G>::type>(a, s)
                                        : launch_dconv_down<G, BigT, typename DTileFor<G>::Down>(a, s);
  }
  if (rc || !dbias) return rc;
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cdiv(G::CS, 4)), dim3(256), 0, s, (const float*)parts, (int)tiles,
                     (int)G::CS, dbias, accumulate_dbias);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

// Scatter-form configuration per geometry: images per workgroup, resident N tiles, weight prefetch.
template <class G> struct UConf { using type = void; };
template <> struct UConf<GDec3> { using type = SConf<GDec3, 1, 6>; };
template <> struct UConf<GEnc2> { using type = SConf<GEnc2, 1, 8>; };
template <> struct UConf<GDec2> { using type = SConf<GDec2, 5, 2>; };
template <> struct UConf<GEnc3> { using type = SConf<GEnc3, 4, 2>; };
template <> struct UConf<GEnc4> { using type = SConf<GEnc4, 8, 1>; };
// 128 x 128 stack: the parity-class planes of a 16-channel group must fit LDS (<= 80 KB: two workgroups per CU), which
// 30 x 30 and 14 x 14 outputs do; 63 x 63 / 64 x 64 / 128 x 128 outputs (262 KB) take the gather engine below
template <> struct UConf<GX3> { using type = SConf<GX3, 1, 2>; };
template <> struct UConf<GX4> { using type = SConf<GX4, 4, 1>; };

// Direct (register-accumulating) transposed conv, dconv_up.h: tile per geometry, or NoTile = not used.  It takes
// precedence over the scatter kernel / the gather engine where a tile is given.  A/B on one box (round 3, us):
//   enc2@128 data gradient  gather engine 1634 -> direct 1243 (8 waves; 4 waves 1424)
//   dec4@128 forward        gather engine  717 -> direct  535
//   TIA conv4 forward       gather engine  775 -> direct  434
//   enc2 data gradient      scatter        425 vs direct  606 (its K loop alone runs 424, staging +76, epilogue
//                           +106; staggered starts, 16-byte-aligned stores and prefetching the ReLU operand under
//                           the last chunk's MFMAs each changed nothing): the scatter kernel stays
struct NoTile {};
template <class G> struct UpDirect { using type = NoTile; };
template <> struct UpDirect<GEnc2> { using type = NoTile; };
template <> struct UpDirect<GEnc3> { using type = NoTile; };
template <> struct UpDirect<GEnc4> { using type = NoTile; };
template <> struct UpDirect<GDec2> { using type = NoTile; };
template <> struct UpDirect<GDec3> { using type = NoTile; };
template <> struct UpDirect<GX2> { using type = DTile<128, 128, 8, 2, 4>; };
template <> struct UpDirect<GX3> { using type = NoTile; };
template <> struct UpDirect<GX4> { using type = NoTile; };
template <> struct UpDirect<GY4> { using type = DTile<64, 256, 4, 2, 4>; };
template <> struct UpDirect<GT4> { using type = DTile<32, 256, 4, 1, 8>; };
template <class G> constexpr bool kUpDirect = !std::is_same<typename UpDirect<G>::type, NoTile>::value;

// The bf16x6 scatter kernel (buconv.h): the decoder's conv3 forward (the update's largest launch), conv2 forward, and the
// encoder's conv2 / conv3 / conv4 data gradients.  The weight pack's format follows the kernel: repo_debug_bconv toggles both, so
// a pack written under one setting must not be used under the other (tests re-pack).
template <class G> struct BUConf { using type = void; };
template <> struct BUConf<GDec3> { using type = BSConf<GDec3, 1, 4>; };
template <> struct BUConf<GEnc2> { using type = BSConf<GEnc2, 1, 4>; };
template <> struct BUConf<GEnc3> { using type = BSConf<GEnc3, 4, 4>; };   // CS = 128: two K-slices
template <> struct BUConf<GEnc4> { using type = BSConf<GEnc4, 8, 2>; };   // CS = 256: four
template <> struct BUConf<GDec2> { using type = BSConf<GDec2, 5, 4>; };   // CS = 128, k5: per-class tap sets
template <class G>
static bool buconv_on() {
  if constexpr (!std::is_void<typename BUConf<G>::type>::value) return t_bconv_enabled != 0;
  return false;
}

template <class G>
static size_t conv_up_ws_scatter() {
  using C = typename UConf<G>::type;
  using BC = typename BUConf<G>::type;
  if constexpr (kUpDirect<G>) return UpGeo<G>::PACK_FLOATS * sizeof(float);
  else if constexpr (std::is_void<C>::value) return 0;
  else if constexpr (!std::is_void<BC>::value) return BC::PACK_BYTES > C::PACK_FLOATS * sizeof(float) ? BC::PACK_BYTES : C::PACK_FLOATS * sizeof(float);
  else return C::PACK_FLOATS * sizeof(float);
}
// encoder conv2's data gradient also has a gather-form kernel (tconv_up.h): its pack sits behind the scatter kernels'
template <class G> constexpr bool kTconvUp = false;
#ifndef TCU_DISABLE   // A/B builds (tools/build_variant.sh)
template <> constexpr bool kTconvUp<GEnc2> = true;
#endif
template <class G>
static size_t conv_up_tcu_off() { return (conv_up_ws_scatter<G>() + 255) & ~(size_t)255; }
template <class G>
static size_t conv_up_ws_bytes() {
  if constexpr (kTconvUp<G>) return conv_up_tcu_off<G>() + kTcuPackBytes;
  return conv_up_ws_scatter<G>();
}

template <class G>
static int conv_up_pack_t(const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
  using UC = typename UConf<G>::type;
  using BC = typename BUConf<G>::type;
  if constexpr (kTconvUp<G>) {   // both packs: which kernel a later call takes depends on its epilogue and batch
    if (buconv_on<G>()) {
      if (!ws || ws_bytes < conv_up_ws_bytes<G>()) return REPO_E_WS_TOO_SMALL;
      const int rc = launch_tconv_up_pack(w, (char*)ws + conv_up_tcu_off<G>(), s);
      if (rc) return rc;
    }
  }
  if constexpr (!std::is_void<BC>::value)
    if (buconv_on<G>()) return launch_buconv_pack<G, BC>(w, ws, ws_bytes, s);
  if constexpr (kUpDirect<G>) return launch_dconv_up_pack<G>(w, ws, ws_bytes, s);
  else if constexpr (!std::is_void<UC>::value) return launch_uconv_pack<G, UC>(w, ws, ws_bytes, s);
  else return REPO_OK;  // the 3-channel layers read the native weights
}

template <class G>
static int conv_up_t(int64_t nimg, const float* small, const float* w, const float* bias, float* big, int epi,
                     const float* aux, int packed, void* ws, size_t ws_bytes, hipStream_t s) {
  if (nimg * (int64_t)G::CB * G::PB >= kMaxBufElems || nimg * (int64_t)G::CS * G::PS >= kMaxBufElems) return REPO_E_SHAPE;
  using UC = typename UConf<G>::type;
  using BC = typename BUConf<G>::type;
  // the channel-quad mask is what the scatter kernels' drain reads (a pixel's four channels per item)
  if (epi == REPO_EPI_MUL_CMASK && (kUpDirect<G> || std::is_void<UC>::value || G::CB % 4 != 0)) return REPO_E_BADARG;
  if (epi == REPO_EPI_FILM_RELU && (kUpDirect<G> || std::is_void<UC>::value)) return REPO_E_BADARG;   // the scatter kernels' drains
  if constexpr (kTconvUp<G>) {
    const bool epi_ok = epi == REPO_EPI_NONE || epi == REPO_EPI_MUL_DRELU || epi == REPO_EPI_MUL_CMASK;   // the encoder backward's forms; ReLU / FiLM: the scatter kernel
    if (buconv_on<G>() && epi_ok && nimg >= 4 && ws && ws_bytes >= conv_up_ws_bytes<G>()) {
      char* pack = (char*)ws + conv_up_tcu_off<G>();
      if (!packed) {
        const int rc = launch_tconv_up_pack(w, pack, s);
        if (rc) return rc;
      }
      return launch_tconv_up(small, pack, bias, aux, big, nimg, epi, s);
    }
  }
  if constexpr (!std::is_void<BC>::value)
    if (buconv_on<G>()) return launch_buconv_scatter<G, BC>(small, w, bias, aux, big, nimg, epi, packed, ws, ws_bytes, s);
  if constexpr (kUpDirect<G>) {
    return launch_dconv_up<G, typename UpDirect<G>::type>(small, w, bias, aux, big, nimg, epi, packed, ws, ws_bytes, s);
  } else if constexpr (!std::is_void<UC>::value) {
    return launch_uconv_scatter<G, UC>(small, w, bias, aux, big, nimg, epi, packed, ws, ws_bytes, s);
  } else {
    // 3-channel outputs (encoder conv1 data-gradient, plain decoder conv4) and, in the 128 x 128 stack, the outputs
    // whose class planes do not fit LDS: the four output parity classes read the same (J x J) input taps, so they
    // are merged on M (4 * CB rows) in the gather engine of igemm.h
    static_assert(G::KS % 2 == 0, "the gather engine pairs horizontally adjacent output pixels");
    ConvUpMergedOp<G, float, 0> op{small, w, bias, aux, big, (int)nimg, epi, nullptr, nullptr, nullptr, 0.f, 0.f};
    if constexpr (G::CB < 8) return launch_igemm<T32x256>(op, 4 * G::CB, nimg * (int64_t)op.NY * op.NX, 1, s);
    else return launch_igemm<T64x128>(op, 4 * G::CB, nimg * (int64_t)op.NY * op.NX, 1, s);
  }
}

template <class G>
static int dwgrad_ips(int64_t nimg) {
  using T = typename DTileFor<G>::Wgrad;
  const long tiles = ((G::CS + T::BM - 1) / T::BM) * ((G::CB * G::KK + T::BN - 1) / T::BN);
  long want = (DTileFor<G>::WGT + tiles - 1) / tiles;
  long ips = (nimg + want - 1) / want;
  const long min_ips = T::GI * ((G::PS >= 512) ? 1 : (G::PS >= 64 ? 2 : 4));
  if (ips < min_ips) ips = min_ips;
  ips = (ips + T::GI - 1) / T::GI * T::GI;
  return (int)ips;
}
template <class G>
static int dwgrad_splits(int64_t nimg) {
  const long ips = dwgrad_ips<G>(nimg);
  return (int)((nimg + ips - 1) / ips);
}
template <class G>
static size_t wgrad_ws_bytes(int64_t nimg) {
  return (size_t)dwgrad_splits<G>(nimg) * G::CS * (G::CB * G::KK + 1) * sizeof(float);
}

static void launch_conv_slab_reduce(const float* ws, int splits, int cs, int nw, float* dw, float* db, int accumulate,
                                    hipStream_t s) {
  const int total = cs * (nw + 1);
  if (splits >= 64 && total <= 65536) {
    hipLaunchKernelGGL(conv_slab_reduce_wave_kernel, dim3(cdiv(total, 64)), dim3(64 * SLAB_REDUCE_WAVES), 0, s, ws, splits, cs, nw, dw, db,
                       accumulate, 0, 0, (float*)nullptr);
  } else {
    const int blocks = cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048;
    hipLaunchKernelGGL(conv_slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, splits, cs, nw, dw, db, accumulate);
  }
}

// The bf16x6 weight-gradient kernel (bwgrad.h) for the layers whose bands fill whole 16-k blocks reasonably (enc4's 2 x 2
// planes would run 4 real k in a block of 16; the 3-channel layers are not MFMA-bound): tile BM x BN, waves, images per
// chunk, small rows per band.  The split-K over image groups (slabs, reduce) is the fp32 kernel's.
template <class G> struct BWgradFor { using type = NoBTile; };
template <> struct BWgradFor<GDec3> { using type = WTile<64, 128, 1, 4, 1, 7>; };   // 531 -> 465 us (64 x 64 wave tiles: 551)
template <> struct BWgradFor<GEnc3> { using type = WTile<64, 256, 1, 4, 1, 6>; };   // 215 -> 172-182 us
// measured and left on the fp32 kernel: enc2 (31 x 31 planes: 343-445 us against 320) and dec2 (232-244 against 247) --
// with one or two waves per SIMD the in-register split of the B fragments (44 dependent vector instructions per 8
// elements) is not hidden behind 12 MFMAs
template <class G> constexpr bool kBWgrad = !std::is_same<typename BWgradFor<G>::type, NoBTile>::value;

// twgrad.h (both operands split at staging, `big` read through the LDS's transposing load): k-blocks per chunk, 0 = not
// on this engine.  A PAIR of workgroups (the two row parities of the taps) owns the whole dw of its images: images per
// split = ceil(nimg / 128), one slab per pair.
template <class G> constexpr int kTWgradNBK = 0;
#ifndef TW_DISABLE   // A/B builds (tools/build_variant.sh): the previous engines
template <> constexpr int kTWgradNBK<GDec3> = 2;
template <> constexpr int kTWgradNBK<GEnc2> = 2;
#endif
// staging waves: 8 where the multiplying waves (4 taps: 64 accumulator registers) fit three waves per SIMD
template <class G> constexpr int kTWgradNPW = 4;
template <> constexpr int kTWgradNPW<GEnc2> = 8;
static int twgrad_ips(int64_t nimg) { return (int)((nimg + 127) / 128); }
static int twgrad_splits(int64_t nimg) { return (int)((nimg + twgrad_ips(nimg) - 1) / twgrad_ips(nimg)); }

static inline int chansum_splits(int64_t nimg, int64_t C, int64_t P) {
  long want = (4096 + C - 1) / C;
  long min_imgs = (8192 + P - 1) / P;
  long maxs = (nimg + min_imgs - 1) / min_imgs;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  long ips = (nimg + want - 1) / want;
  return (int)((nimg + ips - 1) / ips);
}
static int channel_sum_launch(int64_t nimg, int64_t C, int64_t P, const float* x, float* out, int accumulate, float* ws,
                              hipStream_t stream) {
  const int splits = chansum_splits(nimg, C, P);
  const int ips = (int)((nimg + splits - 1) / splits);
  hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)C, (unsigned)splits), dim3(256), 0, stream, x, (int)nimg, (int)C,
                     (int)P, ips, ws);
  REPO_CHECK_LAUNCH();
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3(cdiv(C, 4)), dim3(256), 0, stream, (const float*)ws, splits, (int)C,
                     out, accumulate);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

// slabs of the weight gradient (the larger of the engines' sets: the engine is a thread-local switch), then the channel
// sums' partials for dbias_big where the engine does not produce them
template <class G>
static size_t wgrad_ws_slabs(int64_t nimg) {
  size_t b = wgrad_ws_bytes<G>(nimg);
  if constexpr (kTWgradNBK<G> > 0) {
    const size_t t = (size_t)twgrad_splits(nimg) * TWGeo<G, kTWgradNBK<G>, kTWgradNPW<G>>::SLAB * sizeof(float);
    if (t > b) b = t;
  }
  return (b + 255) & ~(size_t)255;
}
template <class G>
static size_t wgrad_ws_total(int64_t nimg) {
  return wgrad_ws_slabs<G>(nimg) + (size_t)chansum_splits(nimg, G::CB, G::PB) * G::CB * sizeof(float);
}

template <class G, class BigT>
static int conv_wgrad_t(int64_t nimg, const float* small, const BigT* big, float* dw, float* db, float* dbig,
                        int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  if (nimg * (int64_t)G::CB * G::PB >= kMaxBufElems || nimg * (int64_t)G::CS * G::PS >= kMaxBufElems) return REPO_E_SHAPE;
  if (!ws || ws_bytes < wgrad_ws_total<G>(nimg)) return REPO_E_WS_TOO_SMALL;
  const int dips = dwgrad_ips<G>(nimg), dsplits = dwgrad_splits<G>(nimg);
  WgradArgs a{small, big, (float*)ws, (int)nimg, dips, db != nullptr,
              (unsigned)(nimg * G::CS * G::PS * sizeof(float)), (unsigned)(nimg * G::CB * G::PB * sizeof(BigT))};
  int rc;
  if constexpr (kTWgradNBK<G> > 0 && std::is_same<BigT, float>::value) {
    if (t_bconv_enabled) {
      using TG = TWGeo<G, kTWgradNBK<G>, kTWgradNPW<G>>;
      const int tsplits = twgrad_splits(nimg);
      a.imgs_per_split = twgrad_ips(nimg);
      // where the taps reach every row and column of `big` (decoder conv3; not the 31 x 31 planes of encoder conv2, whose
      // last row and column no window touches) every element is staged exactly once and its channel sums ride along
#ifdef TW_NO_DBIG   // A/B builds: the separate channel-sum pass
      constexpr bool covers = false;
#else
      constexpr bool covers = G::HB == 2 * (G::HS - 1) + G::KS;
#endif
      a.want_dbig = covers && dbig != nullptr;
      rc = launch_tconv_wgrad<G, kTWgradNBK<G>, kTWgradNPW<G>>(a, tsplits, s);
      if (rc) return rc;
      hipLaunchKernelGGL(conv_slab_reduce_wave_kernel, dim3(cdiv(TG::SLAB, 64)), dim3(64 * SLAB_REDUCE_WAVES), 0, s, (const float*)ws, tsplits,
                         G::CS, G::CB * G::KK, dw, db, accumulate, G::KK, G::CB, covers ? dbig : nullptr);
      REPO_CHECK_LAUNCH();
      if (!covers && dbig)
        return channel_sum_launch(nimg, G::CB, G::PB, big, dbig, accumulate, (float*)((char*)ws + wgrad_ws_slabs<G>(nimg)), s);
      return REPO_OK;
    }
  }
  if constexpr (kBWgrad<G> && std::is_same<BigT, float>::value) {
    static_assert(DTileFor<G>::Wgrad::GI % BWgradFor<G>::type::GI == 0, "images per split: a multiple of both kernels' chunks");
    if (t_bconv_enabled) rc = launch_bconv_wgrad<G, typename BWgradFor<G>::type>(a, dsplits, s);
    else rc = launch_dconv_wgrad<G, BigT, typename DTileFor<G>::Wgrad>(a, dsplits, s);
  } else {
    rc = launch_dconv_wgrad<G, BigT, typename DTileFor<G>::Wgrad>(a, dsplits, s);
  }
  if (rc) return rc;
  launch_conv_slab_reduce((const float*)ws, dsplits, G::CS, G::CB * G::KK, dw, db, accumulate, s);
  REPO_CHECK_LAUNCH();
  if (dbig) {
    if constexpr (std::is_same<BigT, float>::value)
      return channel_sum_launch(nimg, G::CB, G::PB, big, dbig, accumulate, (float*)((char*)ws + wgrad_ws_slabs<G>(nimg)), s);
    else
      return REPO_E_BADARG;
  }
  return REPO_OK;
}

}  // namespace repo

using namespace repo;

#define REPO_LAYER_SWITCH(layer, CALL)                 \
  switch (layer) {                                     \
    case 0: { using G = GEnc1; CALL; }                 \
    case 1: { using G = GEnc2; CALL; }                 \
    case 2: { using G = GEnc3; CALL; }                 \
    case 3: { using G = GEnc4; CALL; }                 \
    case 4: { using G = GDec2; CALL; }                 \
    case 5: { using G = GDec3; CALL; }                 \
    case 6: { using G = GDec4; CALL; }                 \
    case 7: { using G = GX1; CALL; }                   \
    case 8: { using G = GX2; CALL; }                   \
    case 9: { using G = GX3; CALL; }                   \
    case 10: { using G = GX4; CALL; }                  \
    case 11: { using G = GY4; CALL; }                  \
    case 12: { using G = GY5; CALL; }                  \
    case 13: { using G = GT4; CALL; }                  \
    default: return REPO_E_BADARG;                     \
  }

extern "C" int repo_conv_down(int layer, int64_t nimg, const void* big, int big_is_u8, const float* w,
                              const float* bias, float* small, int epi, const void* aux_, float* dbias_small,
                              int accumulate_dbias, unsigned char* relu_cmask, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(nimg >= 0, REPO_E_SHAPE);
  if (nimg == 0) return REPO_OK;
  REPO_REQUIRE(big && w && small, REPO_E_BADARG);
  REPO_REQUIRE(epi == REPO_EPI_NONE || epi == REPO_EPI_RELU ||
                   ((epi == REPO_EPI_MUL_DRELU || epi == REPO_EPI_MUL_MASK4 || epi == REPO_EPI_FILM_RELU) && aux_), REPO_E_BADARG);
  REPO_REQUIRE(!relu_cmask || epi == REPO_EPI_RELU, REPO_E_BADARG);
  REPO_REQUIRE(epi != REPO_EPI_FILM_RELU || !dbias_small, REPO_E_BADARG);
  const float* aux = (const float*)aux_;  // fp32 activations, or the quad mask's bytes (REPO_EPI_MUL_MASK4)
  if (big_is_u8) {
    REPO_REQUIRE(layer == 0 || layer == 7, REPO_E_BADARG);
    if (layer == 7)
      return conv_down_t<GX1, uint8_t>(nimg, (const uint8_t*)big, w, bias, small, epi, aux, dbias_small,
                                       accumulate_dbias, relu_cmask, ws, ws_bytes, stream);
    return conv_down_t<GEnc1, uint8_t>(nimg, (const uint8_t*)big, w, bias, small, epi, aux, dbias_small,
                                       accumulate_dbias, relu_cmask, ws, ws_bytes, stream);
  }
  REPO_LAYER_SWITCH(layer, return (conv_down_t<G, float>(nimg, (const float*)big, w, bias, small, epi, aux, dbias_small,
                                                         accumulate_dbias, relu_cmask, ws, ws_bytes, stream)))
}

extern "C" size_t repo_conv_down_workspace_bytes(int layer, int64_t nimg) {
  if (nimg <= 0) return 0;
  // (the frame type is not known here: room for whichever of the float / uint8 variants needs more)
  REPO_LAYER_SWITCH(layer, return (std::max(bconv_down_on<G, float>(nimg) ? bconv_pack_bytes<G, float>() : 0,
                                            bconv_down_on<G, uint8_t>(nimg) ? bconv_pack_bytes<G, uint8_t>() : 0) +
                                   (size_t)std::max(conv_down_tiles<G, float>(nimg), conv_down_tiles<G, uint8_t>(nimg)) * G::CS * sizeof(float)))
}

extern "C" int repo_debug_bconv(int enable) {
  const int prev = t_bconv_enabled;
  t_bconv_enabled = enable ? 1 : 0;
  return prev;
}

extern "C" size_t repo_conv_up_workspace_bytes(int layer) {
  REPO_LAYER_SWITCH(layer, return (conv_up_ws_bytes<G>()))
}

extern "C" int repo_conv_up_pack(int layer, const float* w, void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(w, REPO_E_BADARG);
  REPO_LAYER_SWITCH(layer, return (conv_up_pack_t<G>(w, ws, ws_bytes, stream)))
}

extern "C" int repo_conv_up(int layer, int64_t nimg, const float* small, const float* w, const float* bias,
                            float* big, int epi, const float* aux, int ws_is_packed, void* ws, size_t ws_bytes,
                            hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(nimg >= 0, REPO_E_SHAPE);
  if (nimg == 0) return REPO_OK;
  REPO_REQUIRE(small && w && big, REPO_E_BADARG);
  REPO_REQUIRE(epi == REPO_EPI_NONE || epi == REPO_EPI_RELU ||
                   ((epi == REPO_EPI_MUL_DRELU || epi == REPO_EPI_MUL_CMASK || epi == REPO_EPI_FILM_RELU) && aux), REPO_E_BADARG);
  REPO_LAYER_SWITCH(layer, return (conv_up_t<G>(nimg, small, w, bias, big, epi, aux, ws_is_packed, ws, ws_bytes, stream)))
}

extern "C" size_t repo_conv_wgrad_workspace_bytes(int layer, int64_t nimg) {
  if (nimg <= 0) return 0;
  REPO_LAYER_SWITCH(layer, return (wgrad_ws_total<G>(nimg)))
}

extern "C" int repo_conv_wgrad(int layer, int64_t nimg, const float* small, const void* big, int big_is_u8,
                               float* dw, float* dbias_small, float* dbias_big, int accumulate, void* ws,
                               size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(nimg > 0, REPO_E_SHAPE);
  REPO_REQUIRE(small && big && dw, REPO_E_BADARG);
  if (big_is_u8) {
    REPO_REQUIRE((layer == 0 || layer == 7) && !dbias_big, REPO_E_BADARG);
    if (layer == 7)
      return conv_wgrad_t<GX1, uint8_t>(nimg, small, (const uint8_t*)big, dw, dbias_small, nullptr, accumulate, ws,
                                        ws_bytes, stream);
    return conv_wgrad_t<GEnc1, uint8_t>(nimg, small, (const uint8_t*)big, dw, dbias_small, nullptr, accumulate, ws,
                                        ws_bytes, stream);
  }
  REPO_LAYER_SWITCH(layer, return (conv_wgrad_t<G, float>(nimg, small, (const float*)big, dw, dbias_small, dbias_big,
                                                          accumulate, ws, ws_bytes, stream)))
}

// loss partials, then either the kernel's per-channel partials or the channel-sum pass's (the fp32 twin)
static size_t dec4_nll_ws_floats(int64_t nimg) {
  const size_t n = (size_t)dec4_nll_grid(nimg);
  const size_t cs = (size_t)chansum_splits(nimg, GDec4::CB, GDec4::PB) * GDec4::CB;
  return n + (3 * n > cs ? 3 * n : cs);
}
extern "C" size_t repo_decoder_out_nll_workspace_bytes(int64_t nimg) {
  return nimg <= 0 ? 0 : dec4_nll_ws_floats(nimg) * sizeof(float);
}

template <class TgtT>
static int decoder_out_nll_t(int64_t nimg, const float* h3, const float* w, const float* bias, const TgtT* target,
                             float grad_scale, float* recon, float* dpre, unsigned char* mask4, float* loss_sum,
                             float* dbias, int accumulate_dbias, void* ws, hipStream_t stream) {
  using G = GDec4;
  const int nparts = dec4_nll_grid(nimg);
  float* chan = (float*)ws + nparts;
  NllArgs a{h3, w, bias, target, recon, dpre, mask4, (float*)ws, grad_scale, (int)nimg,
            (unsigned)(nimg * G::CS * G::PS * sizeof(float))};
  // the bf16x6 kernel (bdec4.h) unless repo_debug_bconv(0) asks for the fp32-MFMA twin
  if (t_bconv_enabled) {
    a.chan_partials = dbias ? chan : nullptr;
    hipLaunchKernelGGL((bdec4_nll_kernel<TgtT>), dim3(nparts), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((dconv_dec4_nll_kernel<TgtT>), dim3(nparts), dim3(256), 0, stream, a);
  }
  REPO_CHECK_LAUNCH();
  if (loss_sum) {
    hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(1024), 0, stream, (const float*)ws, nparts,
                       loss_sum, 0);
    REPO_CHECK_LAUNCH();
  }
  if (dbias) {   // the output layer's bias gradient = the channel sums of dpre
    if (t_bconv_enabled) {
      hipLaunchKernelGGL(partial_sum_rows_kernel, dim3(3), dim3(256), 0, stream, (const float*)chan, nparts, dbias, grad_scale,
                         accumulate_dbias);
      REPO_CHECK_LAUNCH();
    } else {
      if (!dpre) return REPO_E_BADARG;
      return channel_sum_launch(nimg, G::CB, G::PB, dpre, dbias, accumulate_dbias, chan, stream);
    }
  }
  return REPO_OK;
}

extern "C" int repo_decoder_out_nll(int64_t nimg, const float* h3, const float* w, const float* bias,
                                    const void* target, int target_is_u8, float grad_scale, float* recon, float* dpre,
                                    unsigned char* relu_mask4, float* loss_sum, float* dbias, int accumulate_dbias,
                                    void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(nimg > 0, REPO_E_SHAPE);
  REPO_REQUIRE(h3 && w && target, REPO_E_BADARG);
  REPO_REQUIRE(nimg * (int64_t)GDec4::CS * GDec4::PS < kMaxBufElems, REPO_E_SHAPE);
  REPO_REQUIRE(ws && ws_bytes >= repo_decoder_out_nll_workspace_bytes(nimg), REPO_E_WS_TOO_SMALL);
  if (target_is_u8)
    return decoder_out_nll_t<uint8_t>(nimg, h3, w, bias, (const uint8_t*)target, grad_scale, recon, dpre, relu_mask4,
                                      loss_sum, dbias, accumulate_dbias, ws, stream);
  return decoder_out_nll_t<float>(nimg, h3, w, bias, (const float*)target, grad_scale, recon, dpre, relu_mask4, loss_sum,
                                  dbias, accumulate_dbias, ws, stream);
}

// ---- a 3-channel transposed conv fused with the pixel likelihood on the gather engine (any output size): what the
//      128 x 128 stack's output layer (layer 12) runs on; layer 6 goes through it too in the tests, as a second
//      implementation of repo_decoder_out_nll.
template <class G>
static long up_nll_parts(int64_t nimg) {
  constexpr int NY = (G::HB + 1) / 2, NX = (G::WB + 1) / 2;
  return (nimg * (long)NY * NX + T32x256::BN - 1) / T32x256::BN;  // one workgroup row (M = 12 <= 32)
}
template <class G, class TgtT>
static int conv_up_nll_t(int64_t nimg, const float* small, const float* w, const float* bias, const TgtT* target,
                         float grad_scale, float* recon, float* dpre, float* loss_sum, void* ws, size_t ws_bytes,
                         hipStream_t stream) {
  static_assert(G::CB < 8 && G::KS % 2 == 0, "output layers only");
  if (nimg * (int64_t)G::CB * G::PB >= kMaxBufElems || nimg * (int64_t)G::CS * G::PS >= kMaxBufElems) return REPO_E_SHAPE;
  const long nparts = up_nll_parts<G>(nimg);
  if (!ws || ws_bytes < (size_t)nparts * sizeof(float)) return REPO_E_WS_TOO_SMALL;
  ConvUpMergedOp<G, TgtT, 1> op{small, w, bias, nullptr, recon, (int)nimg, REPO_EPI_NONE, target, dpre, (float*)ws,
                                grad_scale, 0.f};
  const int rc = launch_igemm<T32x256>(op, 4 * G::CB, nimg * (int64_t)op.NY * op.NX, 1, stream);
  if (rc) return rc;
  if (loss_sum) {
    hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(1024), 0, stream, (const float*)ws, (int)nparts, loss_sum, 0);
    REPO_CHECK_LAUNCH();
  }
  return REPO_OK;
}

extern "C" size_t repo_conv_up_nll_workspace_bytes(int layer, int64_t nimg) {
  if (nimg <= 0) return 0;
  if (layer == 6) return (size_t)up_nll_parts<GDec4>(nimg) * sizeof(float);
  if (layer == 12) return (size_t)up_nll_parts<GY5>(nimg) * sizeof(float);
  return 0;
}

extern "C" int repo_conv_up_nll(int layer, int64_t nimg, const float* small, const float* w, const float* bias,
                                const void* target, int target_is_u8, float grad_scale, float* recon, float* dpre,
                                float* loss_sum, void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(nimg > 0, REPO_E_SHAPE);
  REPO_REQUIRE(small && w && target, REPO_E_BADARG);
  REPO_REQUIRE(layer == 6 || layer == 12, REPO_E_BADARG);
#define REPO_UPNLL(G)                                                                                                  \
  return target_is_u8 ? conv_up_nll_t<G, uint8_t>(nimg, small, w, bias, (const uint8_t*)target, grad_scale, recon, dpre, \
                                                  loss_sum, ws, ws_bytes, stream)                                         \
                      : conv_up_nll_t<G, float>(nimg, small, w, bias, (const float*)target, grad_scale, recon, dpre,      \
                                                loss_sum, ws, ws_bytes, stream)
  if (layer == 6) { REPO_UPNLL(GDec4); }
  REPO_UPNLL(GY5);
#undef REPO_UPNLL
}

extern "C" size_t repo_channel_sum_workspace_bytes(int64_t nimg, int64_t C, int64_t P) {
  if (nimg <= 0 || C <= 0 || P <= 0) return 0;
  return (size_t)chansum_splits(nimg, C, P) * C * sizeof(float);
}

extern "C" int repo_channel_sum(int64_t nimg, int64_t C, int64_t P, const float* x, float* out, int accumulate,
                                void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(nimg > 0 && C > 0 && P > 0, REPO_E_SHAPE);
  REPO_REQUIRE(x && out, REPO_E_BADARG);
  REPO_REQUIRE(C <= 65535, REPO_E_SHAPE);
  const int splits = chansum_splits(nimg, C, P);
  REPO_REQUIRE(ws && ws_bytes >= (size_t)splits * C * sizeof(float), REPO_E_WS_TOO_SMALL);
  return channel_sum_launch(nimg, C, P, x, out, accumulate, (float*)ws, stream);
}

extern "C" int repo_relu_mask(int64_t n, const float* dy, const float* h, float* y, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(n >= 0, REPO_E_SHAPE);
  if (n == 0) return REPO_OK;
  REPO_REQUIRE(dy && h && y, REPO_E_BADARG);
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks), dim3(256), 0, stream, n, dy, h, y);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}
