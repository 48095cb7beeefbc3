// Encoder conv2's data gradient (a stride-2 k4 transposed conv, 64 -> 32 channels, 14 x 14 -> 31 x 31) in GATHER form on
// the bf16 matrix pipe ("bf16x6", bgemm.h).
//
//   out[img][cb][2y'+py][2x'+px] = sum_{a,c < 2} sum_cs small[img][cs][y'-a][x'-c] * w[cs][cb][2a+py][2c+px]
//
// The scatter kernel (buconv.h) forms per-tap GEMMs over the INPUT pixels and adds them into output planes in LDS
// (read-add-write per tap, odd 31-wide planes drained two pixels at a time): 0.65 of the fp32 peak, the update's top
// kernel by time.  Per output parity class (py, px) the same sum is ONE product with M = 32 (cb), K = 4 taps x 64 cs,
// N = the class's pixels -- nothing is added in memory, every output element is written once:
//   * `small` is staged CHANNEL-INNERMOST, split once: bf16 planes [8-channel octet][17 x 17 bordered pixels][8 cs] (the
//     border is zero: taps that reach outside read it), one (image, 32-channel half) per chunk, two LDS buffers; a B
//     fragment (column = output pixel, 8 k = 8 cs of one tap) is ONE ds_read_b128 per plane at a per-lane pixel address,
//     consecutive pixels 16 B apart (conflict-free); a tap is an immediate offset;
//   * wave = (row parity, 16 output channels), BOTH column parities (they read the same input pixels with different
//     weights); its weights arrive fragment-ready from a pack (class, tap, half, 16-row tile, plane) straight into
//     registers, one tap ahead; per tap the wave walks 16 row tiles (tile = class row, lane = class column): 3 reads -> 12
//     MFMAs (v_mfma_f32_16x16x32_bf16, two column parities x six products) into 16 x 2 accumulators that live across the
//     image's two chunks; then a lane holds outputs (2 x', 2 x' + 1) of four channels: 8-byte stores, a row of 31 pixels is
//     written as 124 contiguous bytes (the first version, wave = parity class, stored single floats at a stride of 8 B and
//     loaded relu's operand the same way: 560 us of epilogue for 186 us of products);
//   * every thread stages (VALU per MFMA ~0.3: no specialised waves needed), the next chunk's loads in flight during
//     the MFMA loop; one barrier per chunk.
// Reference: autograd's input gradient of nn.Conv2d(32, 64, 4, stride=2) (models/encoder.py:36).
#pragma once
#include "twgrad.h"

namespace repo {

struct TcuArgs {
  const float* small;
  const char* pack;     // [class 4][tap 4][half 2][tile 2][plane 3][lane 64][16 B]
  const float* bias;
  const void* aux;
  float* out;
  int nimg, epi, ipw;   // images per workgroup
  unsigned small_bytes;
};

constexpr int kTcuPackBytes = 4 * 4 * 2 * 2 * 3 * 1024;
constexpr int kTcuGrid = 17, kTcuPix = kTcuGrid * kTcuGrid;           // bordered input grid
constexpr int kTcuOct = kTcuPix * 16, kTcuPlane = 4 * kTcuOct;         // bytes: one octet of one plane; one plane (32 cs)
constexpr int kTcuBuf = 3 * kTcuPlane, kTcuLds = 2 * kTcuBuf;          // 55488, 110976

// one thread per (class, tap, half, tile, lane): 8 k of one A fragment, three planes
__global__ __launch_bounds__(256) void tconv_up_pack_kernel(const float* __restrict__ w, char* __restrict__ pack) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 4 * 4 * 2 * 2 * 64) return;
  const int lane = i & 63, mt = (i >> 6) & 1, h = (i >> 7) & 1, t = (i >> 8) & 3, cls = i >> 10;
  const int cb = 16 * mt + (lane & 15), g = lane >> 4;
  const int ky = 2 * (t >> 1) + (cls >> 1), kx = 2 * (t & 1) + (cls & 1);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = w[((32 * h + 8 * g + e) * 32 + cb) * 16 + ky * 4 + kx];
  unsigned pl[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bg_split3(v[2 * e], v[2 * e + 1], pl[0][e], pl[1][e], pl[2][e]);
  char* dst = pack + ((size_t)(((cls * 4 + t) * 2 + h) * 2 + mt) * 3) * 1024 + lane * 16;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4s*>(dst + q * 1024) = u32x4s{pl[q][0], pl[q][1], pl[q][2], pl[q][3]};
}

template <int KEPI>   // REPO_EPI_NONE, REPO_EPI_MUL_DRELU (the encoder backward's) or REPO_EPI_MUL_CMASK: one epilogue per instance keeps the allocation clean
__global__ __launch_bounds__(256) void tconv_up_kernel(TcuArgs p) {
  extern __shared__ __attribute__((aligned(16))) char tcu_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = wid >> 1, mt = wid & 1;   // wave = (row parity, 16 output channels), BOTH column parities: see the epilogue
  const int img_beg = blockIdx.x * p.ipw, img_end = min(p.nimg, img_beg + p.ipw);
  const __amdgpu_buffer_rsrc_t rsm = make_rsrc(p.small, p.small_bytes);
  const __amdgpu_buffer_rsrc_t rpk = make_rsrc(p.pack, (unsigned)kTcuPackBytes);
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out, (unsigned)p.nimg * 32u * 961u * 4u);
  const __amdgpu_buffer_rsrc_t rax = make_rsrc(p.aux ? p.aux : (const void*)p.out,
                                              KEPI == REPO_EPI_MUL_CMASK ? (unsigned)p.nimg * 8u * 961u : (unsigned)p.nimg * 32u * 961u * 4u);

  for (int i = tid; i < kTcuLds / 16; i += 256) reinterpret_cast<f32x4*>(tcu_lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // the borders stay zero

  // tile t = class row y' (output row 2t + py; t = 15 is a phantom for py = 1), lane & 15 = class column x' (outputs 2x', 2x'+1;
  // x' = 15 has only the first); lane >> 4 = k-octet of a fragment / row quad of an accumulator
  const int j = lane & 15, g = lane >> 4;
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = p.bias ? p.bias[16 * mt + 4 * g + r] : 0.f;

  f32x4 acc[16][2];   // [row][column parity]
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: item = (channel quad cq of the half, pixel quad pq of the 196): 392 per chunk
  f32x4 rv[2][4];
  auto gload = [&](int img, int h) __attribute__((always_inline)) {
    const unsigned dead_img = img < img_end ? 0u : kOobOffset;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + 256 * i, cq = v & 7, pq = v >> 3;
      const unsigned dead = ((unsigned)((48 - pq) >> 31) & kOobOffset) | dead_img;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        rv[i][c] = VecLoad<4>::load(rsm, (4u * (unsigned)((img * 64 + 32 * h + 4 * cq + c) * 196 + 4 * pq)) | dead);
    }
  };
  auto lstore = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + 256 * i, cq = v & 7, pq = v >> 3;
      if (pq < 49) {
        int sy = (4 * pq) / 14, sx = 4 * pq - 14 * sy;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned a1, a2, a3, b1, b2, b3;
          bg_split3(rv[i][0][e], rv[i][1][e], a1, a2, a3);
          bg_split3(rv[i][2][e], rv[i][3][e], b1, b2, b3);
          char* dst = buf + (cq >> 1) * kTcuOct + ((sy + 1) * kTcuGrid + sx + 1) * 16 + (cq & 1) * 8;
          *reinterpret_cast<bg_u32x2*>(dst) = bg_u32x2{a1, b1};
          *reinterpret_cast<bg_u32x2*>(dst + kTcuPlane) = bg_u32x2{a2, b2};
          *reinterpret_cast<bg_u32x2*>(dst + 2 * kTcuPlane) = bg_u32x2{a3, b3};
          ++sx;
          if (sx == 14) sx = 0, ++sy;
        }
      }
    }
  };
  // ---- this wave's A fragments of (tap, half): [column parity][plane]
  auto load_a = [&](bg_bf16x8(&fa)[2][3], int t, int h) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const unsigned base = (unsigned)((((((2 * py + c) * 4 + t) * 2 + h) * 2 + mt) * 3)) * 1024u + 16u * (unsigned)lane;
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[c][q] = __builtin_bit_cast(bg_bf16x8, VecLoad<4>::load(rpk, base + (unsigned)q * 1024u));
    }
  };
  const int blane = g * kTcuOct + j * 16;
  auto load_b = [&](bg_bf16x8(&fb)[3], const char* buf, int t, int toff) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q) fb[q] = *reinterpret_cast<const bg_bf16x8*>(buf + q * kTcuPlane + blane + t * (kTcuGrid * 16) + toff);
  };
  // the pair (2 x', 2 x' + 1) of row t, channel 16 mt + 4 g (+ r via the scalar offset): lane 15 takes (29, 30) -- its own
  // second column does not exist and the row's last pixel must not spill into the next row
  auto pair_off = [&](int t, int img) __attribute__((always_inline)) {
    const int y = 2 * t + py;
    const unsigned dead = (unsigned)((30 - y) >> 31) & kOobOffset;
    return (4u * (unsigned)((img * 32 + 16 * mt + 4 * g) * 961 + y * 31 + (j == 15 ? 29 : 2 * j))) | dead;
  };
  bg_bf16x8 fa[4][2][3];   // the current chunk's weights: [tap][column parity][plane]
  // The epilogue's relu' operand (the activation, 64 pairs per lane and image) is requested DURING the image's second chunk, a
  // quarter behind each tap's weight prefetch, and compressed to sign bits while the next tap multiplies (bit (t & 3) * 8 + 2 r
  // + column of word t >> 2): loaded in the epilogue itself, 4 bytes at a stride, it cost three times the products
  auto compute = [&](auto ec, const char* buf, int h, int img, f32x2(&auxv)[4][4], unsigned(&auxb)[4]) __attribute__((always_inline)) {
    constexpr int EPI = decltype(ec)::value;   // -1: no prefetch (first chunk of an image)
    unsigned auxm[4][2];
    auto compress_mask = [&]() __attribute__((always_inline)) {
      unsigned bits = 0;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bits |= (((auxm[tt][0] >> r) & 1u) | (((auxm[tt][1] >> r) & 1u) << 1)) << (tt * 8 + 2 * r);
      return bits;
    };
    auto compress = [&]() __attribute__((always_inline)) {
      unsigned bits = 0;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bits |= ((auxv[tt][r][0] > 0.f ? 1u : 0u) | (auxv[tt][r][1] > 0.f ? 2u : 0u)) << (tt * 8 + 2 * r);
      return bits;
    };
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      if (EPI == REPO_EPI_MUL_DRELU) {
        if (t4 > 0) auxb[t4 - 1] = compress();   // the previous tap's batch has had a whole tap to arrive
#pragma unroll
        for (int t = 4 * t4; t < 4 * t4 + 4; ++t) {
          const unsigned vo = pair_off(t, img);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            auxv[t & 3][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rax, vo, r * 961 * 4, 0));
        }
      }
      if (EPI == REPO_EPI_MUL_CMASK) {   // the channel-quad mask: byte (image, quad 4 mt + g, pixel), bit r; two pixels per lane
        if (t4 > 0) auxb[t4 - 1] = compress_mask();
#pragma unroll
        for (int t = 4 * t4; t < 4 * t4 + 4; ++t) {
          const int y = 2 * t + py;
          const unsigned dead = (unsigned)((30 - y) >> 31) & kOobOffset;
          const unsigned mo = (unsigned)((img * 8 + 4 * mt + g) * 961 + y * 31 + (j == 15 ? 29 : 2 * j)) | dead;
          auxm[t & 3][0] = __builtin_amdgcn_raw_buffer_load_b8(rax, mo, 0, 0);
          auxm[t & 3][1] = __builtin_amdgcn_raw_buffer_load_b8(rax, mo, 1, 0);
        }
      }
      // tap (a, c) reads input pixel (y' - a, x' - c) = bordered (y' + 1 - a, x' + 1 - c)
      const int toff = ((1 - (t4 >> 1)) * kTcuGrid + 1 - (t4 & 1)) * 16;
      bg_bf16x8 fb[2][3];
      load_b(fb[0], buf, 0, toff);
#pragma unroll
      for (int t = 0; t < 16; ++t) {   // (py = 1 runs a sixteenth, phantom row: no branch here)
        if (t + 1 < 16) load_b(fb[(t + 1) & 1], buf, t + 1, toff);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};   // smallest terms first
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t4][c][PA[pr]], fb[t & 1][PB[pr]], acc[t][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (EPI == REPO_EPI_MUL_DRELU) auxb[3] = compress();
    if (EPI == REPO_EPI_MUL_CMASK) auxb[3] = compress_mask();
  };
  auto epilogue = [&](auto ec, int img, unsigned(&auxb)[4]) __attribute__((always_inline)) {
    constexpr int EPI = decltype(ec)::value;
#ifdef TCU_NO_EPI   // ablation build: one store per lane and image keeps the accumulators alive
    float sacc = 0.f;
    for (int t = 0; t < 16; ++t) for (int c = 0; c < 2; ++c) { for (int r = 0; r < 4; ++r) sacc += acc[t][c][r]; acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    p.out[(size_t)img * 32 * 961 + tid] = sacc;
    return;
#endif
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const unsigned vo = pair_off(t, img);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v0 = acc[t][0][r] + bv[r], v1 = acc[t][1][r] + bv[r];
        // lane 15: (x = 29 from lane 14's second column, x = 30 = its own first)
        const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v1), 0x111, 0xf, 0xf, false));
        float a = j == 15 ? up : v0, b = j == 15 ? v0 : v1;
        if (EPI == REPO_EPI_MUL_DRELU || EPI == REPO_EPI_MUL_CMASK) {
          const unsigned bits = auxb[t >> 2] >> ((t & 3) * 8 + 2 * r);
          a = bits & 1 ? a : 0.f, b = bits & 2 ? b : 0.f;
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bg_u32x2, f32x2{a, b}), rout, vo, r * 961 * 4, 0);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // one chunk: products (+ the aux prefetch of epilogue kind PC, -1 = none) and, after an image's second chunk, its epilogue
  // (gfx950's vmcnt counts stores: a weight load issued after the epilogue's 64 stores cannot be waited for before they
  // have all been acknowledged -- so the NEXT chunk's weights, all four taps, are requested before the epilogue)
  auto chunk = [&](auto pc, auto ec, const char* buf, int h, int img) __attribute__((always_inline)) {
    f32x2 auxv[4][4];
    unsigned auxb[4];
    compute(pc, buf, h, img, auxv, auxb);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) load_a(fa[t4], t4, h ^ 1);
    if (h) epilogue(ec, img, auxb);
  };

  if (img_beg >= img_end) return;
  __syncthreads();   // the zero fill
  gload(img_beg, 0);
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) load_a(fa[t4], t4, 0);
  lstore(tcu_lds);
  __syncthreads();
  int c = 0;
  for (int img = img_beg; img < img_end; ++img) {
#pragma unroll
    for (int h = 0; h < 2; ++h, ++c) {
      gload(h ? img + 1 : img, h ^ 1);                      // the next chunk
      const char* buf = tcu_lds + (c & 1) * kTcuBuf;
      using N1 = std::integral_constant<int, -1>;
      using KE = std::integral_constant<int, KEPI>;
      if (!h) chunk(N1{}, N1{}, buf, 0, img);
      else chunk(std::integral_constant<int, (KEPI == REPO_EPI_MUL_DRELU || KEPI == REPO_EPI_MUL_CMASK) ? KEPI : -1>{}, KE{}, buf, 1, img);
      lstore(tcu_lds + ((c + 1) & 1) * kTcuBuf);            // (an image past the range loaded zeros: harmless)
      __syncthreads();
    }
  }
}

inline int launch_tconv_up_pack(const float* w, char* pack, hipStream_t s) {
  hipLaunchKernelGGL(tconv_up_pack_kernel, dim3(16), dim3(256), 0, s, w, pack);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

inline int launch_tconv_up(const float* small, const char* pack, const float* bias, const void* aux, float* out, int64_t nimg,
                           int epi, hipStream_t s) {
  const int ipw = (int)((nimg + 255) / 256);
  const int nwg = (int)((nimg + ipw - 1) / ipw);
  TcuArgs a{small, pack, bias, aux, out, (int)nimg, epi, ipw, (unsigned)(nimg * 64 * 196 * sizeof(float))};
  hipError_t e;
  if (epi == REPO_EPI_MUL_CMASK) {
    e = hipFuncSetAttribute((const void*)tconv_up_kernel<REPO_EPI_MUL_CMASK>, hipFuncAttributeMaxDynamicSharedMemorySize, kTcuLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tconv_up_kernel<REPO_EPI_MUL_CMASK>, dim3((unsigned)nwg), dim3(256), kTcuLds, s, a);
  } else if (epi == REPO_EPI_MUL_DRELU) {
    e = hipFuncSetAttribute((const void*)tconv_up_kernel<REPO_EPI_MUL_DRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, kTcuLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tconv_up_kernel<REPO_EPI_MUL_DRELU>, dim3((unsigned)nwg), dim3(256), kTcuLds, s, a);
  } else {
    e = hipFuncSetAttribute((const void*)tconv_up_kernel<REPO_EPI_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, kTcuLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tconv_up_kernel<REPO_EPI_NONE>, dim3((unsigned)nwg), dim3(256), kTcuLds, s, a);
  }
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
