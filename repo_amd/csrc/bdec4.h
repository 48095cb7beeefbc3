// The decoder's output layer + pixel likelihood on the bf16 matrix pipe ("bf16x6", bgemm.h): dconv_dec4_nll_kernel's
// frame, tiles and epilogue (dconv.h) with the K loop on v_mfma_f32_16x16x32_bf16.
//
// Why: the fp32 kernel spends 288 v_mfma_f32_16x16x4_f32 (32 cycles each) per wave and tile -- 147 us of matrix time per
// launch at 2450 frames against ~90 us of HBM traffic; its MFMA loop alone ran at 0.52 of the fp32 peak (12 of 16 rows
// useful).  With K ordered (tap, channel) a tap's 32 input channels are exactly ONE 32-k block: 9 taps x 6 partial
// products = 54 MFMAs of 16 cycles per pixel tile, 2.67 x less matrix time at the same accuracy (every fp32 product as six
// exact bf16 x bf16 products, fp32 accumulation).
//
//   M = (py, cb, px) = 12 of 16 rows: the WEIGHTS, split once per workgroup, stay fragment-ready for the whole kernel
//       (lane (m, g): channels 8g .. 8g+7 of tap t as bf16x8): planes 1 and 2 in registers (9 taps x 2 x 4 VGPRs), the
//       third -- one product of six -- in 9 KB of LDS;
//   N = 16 class pixels of a row; K = 32 channels of one tap.
//   The activation patch (10 rows x 34 columns x 32 channels of h3, zero borders) is split ONCE while it is staged and
//   kept as three bf16 planes laid out [row][channel octet g][column][8 channels]: the B fragment of lane (n, g) at tap
//   (ty, tx) is ONE ds_read_b128 per plane, 16 consecutive lanes read 256 consecutive bytes (conflict-free).  Staging: a
//   thread owns (channel quad, pixel quad) -- four 16-byte loads of contiguous pixels, as many wave-level load
//   instructions as the fp32 kernel issues (a first version with dword loads per (pixel, channel octet) was bound by the
//   CU's vector-memory issue rate: 237 us) -- and writes each pixel's four channels as one 8-byte store per plane.  The
//   quad mask of h3 (REPO_EPI_MUL_MASK4) falls out of the same registers.
#pragma once
#include "bgemm.h"
#include "dconv.h"

namespace repo {

template <class TgtT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void bdec4_nll_kernel(NllArgs p) {
  constexpr int CS = 32, HS = 30, PS = 900, HB = 64;
  constexpr int PC = 35;                        // columns x = -2 .. 31 of a patch row (+1: an odd pitch, see the staging)
  constexpr int PLB = 10 * 4 * PC * 16;         // bytes per bf16 plane: [row 10][octet 4][column 35][8 channels]
  constexpr int P_PER = 3;                      // staging items (channel quad, pixel quad): 8 x 75 over 256 threads
  __shared__ __attribute__((aligned(16))) char lds[3 * PLB + 9 * 64 * 16];
  __shared__ float red[16];
  char* const wlo = lds + 3 * PLB;   // the weights' third (smallest) plane, fragment-ready: [tap][lane] 16 bytes

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int ntiles = p.nimg * 4;

  // ---- weights: A fragments in registers.  Lane (m = lj, g = lg), tap = ty_ * 3 + tx_ (ty = 2 - ty_, tx = 2 - tx_):
  //      w[c = 8g + j][cb][py + 2ty][px + 2tx] for m = py*6 + cb*2 + px < 12, zero rows 12 .. 15
  // (planes 1 and 2 in registers -- 72 VGPRs; the third, used by one product of six, in LDS: with all three in
  // registers the kernel spilled at two waves per SIMD)
  bg_bf16x8 wa[9][2];
  {
    const int m = lj < 12 ? lj : 0;
    const int py = m / 6, cb = (m % 6) >> 1, px = m & 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = py + 2 * (2 - tap / 3), kx = px + 2 * (2 - tap % 3);
      u32x4s q1, q2, q3;
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const int c = 8 * lg + 2 * jp;
        float w0 = p.w[((c * 3 + cb) * 6 + ky) * 6 + kx], w1 = p.w[(((c + 1) * 3 + cb) * 6 + ky) * 6 + kx];
        if (lj >= 12) w0 = 0.f, w1 = 0.f;
        unsigned a1, a2, a3;
        bg_split3(w0, w1, a1, a2, a3);
        q1[jp] = a1, q2[jp] = a2, q3[jp] = a3;
      }
      wa[tap][0] = __builtin_bit_cast(bg_bf16x8, q1);
      wa[tap][1] = __builtin_bit_cast(bg_bf16x8, q2);
      if (wv == 0) *reinterpret_cast<u32x4s*>(wlo + (tap * 64 + lane) * 16) = q3;
    }
  }
  // zero borders (columns -2, -1, 30, 31) are never written again
  for (int i = tid; i < 3 * PLB / 16; i += 256) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- patch staging roles (identical for every tile).  An item = (channel quad cq, pixel quad qd of the 10 x 30 strip,
  //      which is CONTIGUOUS per channel in h3): four 16-byte loads (a wave's load instruction covers 8 channels x 128
  //      contiguous bytes), each pixel's four channels split into one 8-byte store per plane.  Lanes run over cq first:
  //      with the odd column count PC the 16 lanes of a store phase hit 32 distinct banks.
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(p.h3, p.h3_bytes);
  int pgo[P_PER], plds[P_PER], pq[P_PER];
#pragma unroll
  for (int j = 0; j < P_PER; ++j) {
    const int v = tid + j * 256, cq = v & 7, qd = v >> 3;
    const bool act = qd < 75;
    const int q0 = 4 * qd;
    pgo[j] = act ? 4 * cq * PS + q0 : -1;                     // element offset of channel 4cq at the quad's first pixel
    // byte offset inside a plane of pixel 0's (octet unit, half): pixels 1 .. 3 follow one unit each, +ROWSTEP units when
    // they run over the row end (only a quad starting at x = 28 does, behind its second pixel)
    plds[j] = (((q0 / 30) * 4 + (cq >> 1)) * PC + q0 % 30 + 2) * 16 + (cq & 1) * 8;
    pq[j] = q0;
  }
  constexpr int ROWSTEP = (4 * PC - 30) * 16;
  // ---- per-lane B unit offsets: wave wv owns class rows 2wv, 2wv+1 of the tile, two 16-pixel halves each
  int bbase[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) bbase[t] = (((2 * wv + (t >> 1)) * 4 + lg) * PC + 16 * (t & 1) + lj) * 16;

  f32x4 rpv[P_PER][4];
  auto gload = [&](int tile) __attribute__((always_inline)) {
    const int img = tile >> 2, cy0 = (tile & 3) * 8;
    const int bias_ = img * CS * PS + (cy0 - 2) * HS;
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        rpv[j][c] = VecLoad<4>::load(rh, pgo[j] >= 0 ? 4u * (unsigned)(pgo[j] + c * PS + bias_) : kOobOffset);
  };
  auto lstore = [&](int tile) __attribute__((always_inline)) {
    const int rg = tile & 3;
    // rows of the strip outside the image (above the first / below the last class rows) are zero
    const int q_lo = rg == 0 ? 60 : 0, q_hi = rg == 3 ? 240 : 300;
    // quad mask of h3: this tile OWNS strip rows 2 .. 9 (its 8 class rows; 6 in an image's last quarter) = pixel quads
    // 15 .. 74 (.. 59), each exactly one aligned quad of the flat tensor (900, 30 * 8k - 60 are multiples of 4)
    const long mbase = (long)(tile >> 2) * CS * PS + ((tile & 3) * 8 - 2) * HS;
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
      if (pgo[j] >= 0) {
#ifndef BD4_NO_MASK
        if (p.mask4 && pq[j] >= 60 && pq[j] < q_hi) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned nib = (rpv[j][c][0] > 0.f ? 1u : 0u) | (rpv[j][c][1] > 0.f ? 2u : 0u) |
                                 (rpv[j][c][2] > 0.f ? 4u : 0u) | (rpv[j][c][3] > 0.f ? 8u : 0u);
            p.mask4[(mbase + pgo[j] + c * PS) >> 2] = (unsigned char)nib;
          }
        }
#endif
        const bool zero = pq[j] < q_lo || pq[j] >= q_hi;
        const int x0 = pq[j] % 30;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned a1, a2, a3, b1, b2, b3;
          bg_split3(zero ? 0.f : rpv[j][0][e], zero ? 0.f : rpv[j][1][e], a1, a2, a3);
          bg_split3(zero ? 0.f : rpv[j][2][e], zero ? 0.f : rpv[j][3][e], b1, b2, b3);
          char* dst = lds + plds[j] + e * 16 + ((e >= 2 && x0 == 28) ? ROWSTEP : 0);
          *reinterpret_cast<bg_u32x2*>(dst) = bg_u32x2{a1, b1};
          *reinterpret_cast<bg_u32x2*>(dst + PLB) = bg_u32x2{a2, b2};
          *reinterpret_cast<bg_u32x2*>(dst + 2 * PLB) = bg_u32x2{a3, b3};
        }
      }
  };

  float lsum = 0.f, csum = 0.f;   // csum: sum of d = recon - target over this lane's channel (the output bias gradient / grad_scale)
  const int G = gridDim.x;
  int tile = blockIdx.x;
  if (tile < ntiles) gload(tile);
  __syncthreads();  // zero fill visible
  for (; tile < ntiles; tile += G) {
    lstore(tile);
    __syncthreads();
    if (tile + G < ntiles) gload(tile + G);
    __builtin_amdgcn_sched_barrier(0);

    // uint8 targets of this tile's epilogue, requested BEFORE the MFMA loop: loaded behind it they put one exposed
    // global-load latency into every tile (ablation: the epilogue was 65 of 222 us)
    const int img = tile >> 2, cy0 = (tile & 3) * 8;
    const int e_odd = lj & 1, e_q = 2 * (lg < 3 ? lg : 0) + e_odd, e_py = e_q / 3, e_cb = e_q % 3;
    unsigned tgt_pre[4];
    if (sizeof(TgtT) == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int cy = cy0 + 2 * wv + (t >> 1), cx = 16 * (t & 1) + lj;
        const int o = ((img * 3 + e_cb) * HB + 2 * cy + e_py) * HB + 4 * (cx >> 1);
        tgt_pre[t] = *reinterpret_cast<const unsigned*>((const uint8_t*)p.target + o);
      }
    }

    f32x4acc acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const bg_bf16x8 wa3 = *reinterpret_cast<const bg_bf16x8*>(wlo + (tap * 64 + lane) * 16);
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // two pixel tiles at a time: their MFMA chains interleave, 24 fragment registers
        bg_bf16x8 fb[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            fb[t][pl] = *reinterpret_cast<const bg_bf16x8*>(lds + pl * PLB + bbase[2 * h + t] + ((tap / 3) * 4 * PC + tap % 3) * 16);
        f32x4acc c0 = acc[2 * h], c1 = acc[2 * h + 1];   // smallest terms first
#ifndef BD4_NO_MFMA
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][1], fb[0][1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][1], fb[1][1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][0], fb[0][2], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][0], fb[1][2], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa3, fb[0][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa3, fb[1][0], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][0], fb[0][1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][0], fb[1][1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][1], fb[0][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][1], fb[1][0], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][0], fb[0][0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tap][0], fb[1][0], c1, 0, 0, 0);
#else
        c0[0] += (float)fb[0][0][0] + (float)fb[0][1][0] + (float)fb[0][2][0] + (float)wa3[0];
        c1[0] += (float)fb[1][0][0] + (float)fb[1][1][0] + (float)fb[1][2][0];
#endif
        acc[2 * h] = c0, acc[2 * h + 1] = c1;
      }
    }

    // ---- epilogue: dconv_dec4_nll_kernel's (the accumulator layout of the 16 x 16 tile is the same): neighbouring lanes
    // swap one (row parity, channel) pair each, a lane stores 16 bytes of d recon and loads one dword of uint8 targets
#ifdef BD4_NO_EPI
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.f) lsum += 1.f;
    if (false)
#endif
    if (lg < 3) {
      const int odd = lj & 1;
      const int q = 2 * lg + odd, py = q / 3, cb = q % 3;  // the pair this lane ends up with
      const float bv = p.bias ? p.bias[cb] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float k0 = odd ? acc[t][2] : acc[t][0], k1 = odd ? acc[t][3] : acc[t][1];
        const float s0 = odd ? acc[t][0] : acc[t][2], s1 = odd ? acc[t][1] : acc[t][3];
        const float r0 = __shfl_xor(s0, 1, 64), r1 = __shfl_xor(s1, 1, 64);
        const float v0 = (odd ? r0 : k0) + bv, v1 = (odd ? r1 : k1) + bv, v2 = (odd ? k0 : r0) + bv, v3 = (odd ? k1 : r1) + bv;
        const int cy = cy0 + 2 * wv + (t >> 1), cx = 16 * (t & 1) + lj;
        const int o = ((img * 3 + cb) * HB + 2 * cy + py) * HB + 4 * (cx >> 1);
        float t0, t1, t2, t3;
        if (sizeof(TgtT) == 1) {
          const unsigned tw = tgt_pre[t];
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
        csum += (d0 + d1) + (d2 + d3);
        if (p.dpre)
          *reinterpret_cast<float4*>(p.dpre + o) = make_float4(d0 * p.grad_scale, d1 * p.grad_scale, d2 * p.grad_scale, d3 * p.grad_scale);
        if (p.recon) *reinterpret_cast<float4*>(p.recon + o) = make_float4(v0, v1, v2, v3);
      }
    }
    __syncthreads();  // patch reads done before the next tile overwrites it
  }
  const float s = block_sum(lsum, red);
  if (tid == 0) p.partials[blockIdx.x] = s;
  if (p.chan_partials) {   // per channel, wave order then lane order inside block_sum: bit-reproducible
    const int my_cb = lg < 3 ? (2 * lg + (lj & 1)) % 3 : -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sc = block_sum(my_cb == c ? csum : 0.f, red);
      if (tid == 0) p.chan_partials[c * gridDim.x + blockIdx.x] = sc;
    }
  }
}

}  // namespace repo
