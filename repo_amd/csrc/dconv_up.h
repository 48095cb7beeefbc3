// Direct (patch-resident) stride-2 TRANSPOSED convolution on the fp32 matrix cores of gfx950: the decoder's forward
// and the encoder's data gradient with register accumulators, as dconv.h's `down` kernel has them.
//
//   up : big[img][cb][2cy+py][2cx+px] = sum_{cs,ty,tx} small[img][cs][cy-ty][cx-tx] * w[cs][cb][py+2ty][px+2tx]
//
// For one output parity class (py, px) that is a stride-1 correlation of `small` with a J x J sub-kernel
// (J = ceil(KS/2)); all four classes read the SAME input taps, so they are merged on M:
//   M = (py, cb, px), px fastest: 4*CB rows       (weights: A operand; a fragment-ready pack [m][k], dconv_up_pack)
//   N = class-plane pixels (img, cy, cx), cy < NY = (HB+1)/2, cx < NX (rows of `small` beyond HS-1 read as zero)
//   K = (cs, u, v), u, v < J:  B(k, n) = Xpad[cs][cy+u][cx+v],  Xpad = `small` with J-1 zero rows / columns around it
//                              A(m, k) = w[cs][cb][py + 2(J-1-u)][px + 2(J-1-v)]   (0 where the index is >= KS)
// The workgroup copies the rows of `small` its class pixels need into a ZERO-BORDERED patch in LDS once per channel
// chunk (contiguous 16-byte global loads, element-wise LDS stores at the padded addresses: the borders are zeroed once
// and never written), the weight chunk beside it; both MFMA operands are read with immediate offsets, exactly as in
// dconv_down_kernel, and the pipeline is the same (two LDS buffers, one barrier per chunk).
// Why beside uconv.h's scatter kernel: that one accumulates into LDS class planes (read-add-write per tap) and drains
// them in a separate phase -- 0.48 of peak on the 31 x 31 planes (27 % of the time in the drain), and planes beyond
// 80 KB (63 x 63, 64 x 64 outputs) do not fit at all (they took the gather engine at 0.38-0.45).
// Epilogue: px pairs are adjacent accumulator rows, so a lane turns (through a per-wave LDS strip) two rows x two
// pixels into FOUR consecutive output floats: one 16-byte store (+ one 16-byte load of the ReLU operand).
//
// Reference: nn.ConvTranspose2d forward in (TIA)VisualObservationModel (models/decoder.py:43-47,170-173) and
// autograd's input gradient of nn.Conv2d in VisualEncoder (models/encoder.py:35-38).
#pragma once
#include "dconv.h"

namespace repo {

template <class G>
struct UpGeo {
  static constexpr int J = (G::KS + 1) / 2, JJ = J * J;
  static constexpr int NY = (G::HB + 1) / 2, NX = (G::WB + 1) / 2, PSC = NY * NX;
  static constexpr int M = 4 * G::CB, K = G::CS * JJ;
  static constexpr size_t PACK_FLOATS = (size_t)M * K;
  // (an odd remainder of the forward conv -- 31 = 2*14 + 4 - 2 + 1 -- leaves one more class row / column, all zero)
  static_assert((NY == G::HS + J - 1 || NY == G::HS + J) && (NX == G::WS + J - 1 || NX == G::WS + J),
                "class planes = input + J - 1 (+ 1 for an uncovered last row / column)");
};

struct UpPackArgs {
  const float* w;
  float* wp;
};
// wp[m][k], m = (py*CB + cb)*2 + px, k = (cs*J + u)*J + v
template <class G>
__global__ __launch_bounds__(256) void dconv_up_pack_kernel(UpPackArgs p) {
  using U = UpGeo<G>;
  const int total = U::M * U::K;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int k = i % U::K, m = i / U::K;
    const int px = m & 1, cb = (m >> 1) % G::CB, py = (m >> 1) / G::CB;
    const int v = k % U::J, u = (k / U::J) % U::J, cs = k / U::JJ;
    const int ky = py + 2 * (U::J - 1 - u), kx = px + 2 * (U::J - 1 - v);
    p.wp[i] = (ky < G::KS && kx < G::KS) ? p.w[((cs * G::CB + cb) * G::KS + ky) * G::KS + kx] : 0.f;
  }
}

struct UpArgs {
  const float* small;
  const float* wp;
  const float* bias;
  const float* aux;
  float* out;
  int nimg, epi;
  unsigned small_bytes, w_bytes;
};

// T: DTile<BM, BN, CK, WM, WN, LDWPAD> (CK = channels of `small` per chunk)
template <class G, class T>
__global__ __launch_bounds__(T::NT) void dconv_up_kernel(UpArgs p) {
  using U = UpGeo<G>;
  constexpr int BM = T::BM, BN = T::BN, CK = T::CK, NT = T::NT, TM = T::TM, TN = T::TN;
  constexpr int J = U::J, JJ = U::JJ, NY = U::NY, NX = U::NX, PSC = U::PSC, MM = U::M;
  constexpr int KSL = CK * JJ;  // k per chunk
  constexpr int NSL = G::CS / CK;
  static_assert(G::CS % CK == 0 && KSL % 4 == 0, "channel chunk must divide CS and give k % 4 == 0");
  constexpr int LDW = BM + T::LDWPAD;
  constexpr int WPP = NX + J - 1;  // patch row pitch (= WS + 2 (J - 1))
  constexpr int NIMG_MAX = (BN - 1) / PSC + 2;
  constexpr int ROWS_MAX = BN / NX + 2 + NIMG_MAX * (J - 1);      // patch rows of a tile, all its images
  constexpr int PLMAX = ROWS_MAX * WPP + 1;                        // (+1: odd pitch between channels)
  constexpr int PLV = (ROWS_MAX * G::WS + 3 * NIMG_MAX + 3) / 4;   // 16-byte vectors of the copied spans, per channel
  constexpr int W_NV = BM * KSL / 4, W_KV = KSL / 4;
  constexpr int W_PER = (W_NV + NT - 1) / NT, P_PER = (CK * PLV + NT - 1) / NT;
  constexpr int NBUF = NSL > 1 ? 2 : 1;
  constexpr int EP = 36;
  constexpr int LDS_K = NBUF * KSL * LDW + NBUF * CK * PLMAX;
  __shared__ __attribute__((aligned(16))) float lds[cmax(LDS_K, (NT / 64) * 32 * EP)];
  float* Wl = lds;
  float* Pl = lds + NBUF * KSL * LDW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / T::WN, wn = wid % T::WN;
  const int li = lane & 31, lh = lane >> 5;

  const int Ntot = p.nimg * PSC;
  const int n0 = xcd_tile(blockIdx.x, gridDim.x) * BN, m0 = blockIdx.y * BM;
  const int nlast = min(n0 + BN, Ntot) - 1;
  const int ia = n0 / PSC, ib = nlast / PSC;
  const int fa = (n0 % PSC) / NX, lb = (nlast % PSC) / NX;
  auto fcy = [&](int i) __attribute__((always_inline)) { return i == ia ? fa : 0; };
  auto lcy = [&](int i) __attribute__((always_inline)) { return i == ib ? lb : NY - 1; };
  // patch rows of image i start at pstart(i) (floats, within a channel's plane)
  auto pstart = [&](int i) __attribute__((always_inline)) {
    int s = 0;
#pragma unroll
    for (int t = 0; t < NIMG_MAX - 1; ++t)
      if (ia + t < i) s += (lcy(ia + t) - fcy(ia + t) + J) * WPP;
    return s;
  };

  const __amdgpu_buffer_rsrc_t rsm = make_rsrc(p.small, p.small_bytes), rw = make_rsrc(p.wp, p.w_bytes);

  // ---- zero the patch buffers once: the staging stores below only ever write interior cells
  for (int i = tid; i < NBUF * CK * PLMAX; i += NT) Pl[i] = 0.f;

  // ---- staging roles
  unsigned woff[W_PER];
  int wlds[W_PER];
#pragma unroll
  for (int j = 0; j < W_PER; ++j) {
    const int v = tid + j * NT, kv = v % W_KV, m = v / W_KV;
    const bool act = (W_NV % NT == 0) || v < W_NV;
    woff[j] = act ? 4u * (unsigned)(min(m0 + m, MM - 1) * U::K + kv * 4) : kOobOffset;
    wlds[j] = act ? (kv * 4) * LDW + m : -1;
  }
  unsigned poff[P_PER];
  int plds[P_PER][4];
#pragma unroll
  for (int j = 0; j < P_PER; ++j) {
    const int v = tid + j * NT, c = v / PLV;
    int q = (v % PLV) * 4;
    // which image's span: spans are concatenated, each rounded up to whole vectors
    int img = -1, xr0 = 0, len = 0, fy = 0, ps = 0;
#pragma unroll
    for (int t = 0; t < NIMG_MAX; ++t) {
      const int i = ia + t;
      if (img < 0 && i <= ib) {
        const int r0 = max(0, fcy(i) - (J - 1)), r1 = min(G::HS - 1, lcy(i));
        const int ln = (r1 - r0 + 1) * G::WS, ln4 = (ln + 3) & ~3;
        if (q < ln4) {
          img = i;
          xr0 = r0;
          len = ln;
          fy = fcy(i);
          ps = pstart(i);
        } else {
          q -= ln4;
        }
      }
    }
    const bool act = c < CK && img >= 0;
    poff[j] = act ? 4u * (unsigned)((img * G::CS + c) * G::PS + xr0 * G::WS + q) : kOobOffset;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ee = q + e;
      const int xrow = xr0 + ee / G::WS, col = ee % G::WS;
      plds[j][e] = (act && ee < len) ? c * PLMAX + ps + (xrow + (J - 1) - fy) * WPP + col + (J - 1) : -1;
    }
  }
  constexpr unsigned W_STEP = 4u * KSL, P_STEP = 4u * CK * G::PS;

  // ---- per-lane LDS bases of the B (class pixel) fragments
  int bbase[TN], bbase1[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = min(n0 + (wn * TN + j) * 32 + li, Ntot - 1);
    const int i = n / PSC, pix = n % PSC;
    bbase[j] = pstart(i) + (pix / NX - fcy(i)) * WPP + pix % NX;
    bbase1[j] = bbase[j] + lh;
  }
  const int abase = lh * LDW + wm * (TM * 32) + li;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 rwv[W_PER], rpv[P_PER];
  auto gload = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < W_PER; ++j)
      rwv[j] = VecLoad<4>::load(rw, woff[j] == kOobOffset ? kOobOffset : woff[j] + (unsigned)t * W_STEP);
#pragma unroll
    for (int j = 0; j < P_PER; ++j)
      rpv[j] = VecLoad<4>::load(rsm, poff[j] == kOobOffset ? kOobOffset : poff[j] + (unsigned)t * P_STEP);
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
#ifdef REPO_UP_SKIP_STAGE
    if (buf >= 0 && p.nimg > 0) return;
#endif
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
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (plds[j][e] >= 0) pl[plds[j][e]] = rpv[j][e];
  };
  auto koff = [](int k) constexpr { return (k / JJ) * PLMAX + ((k % JJ) / J) * WPP + (k % JJ) % J; };
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
#ifdef REPO_UP_SKIP_MFMA
          acc[i][j][0] += av[i] * bv[j];
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
#endif
    }
  };

  // ---- epilogue roles (computed before the last chunk's MFMAs so that the ReLU operand's loads fly under them):
  // item (pass, lane) of tile (i, j) = accumulator row pair a = 4 pass + lane / 16, class-pixel pair q = lane % 16
  const unsigned out_elems = (unsigned)p.nimg * G::CB * G::PB;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(p.out, 4u * out_elems);
  const __amdgpu_buffer_rsrc_t raux = make_rsrc(p.aux ? p.aux : p.out, 4u * out_elems);
  unsigned eo[TM][TN][4];  // byte offset of the item's first output float, or kOobOffset = nothing to write
  unsigned evec = 0;       // bit ((i*TN + j)*4 + pass): the item is four consecutive floats of one output row
  f32x4 a4[TM][TN][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int a = pass * 4 + (lane >> 4), q = lane & 15;
        const int m = m0 + (wm * TM + i) * 32 + 2 * a, n = n0 + (wn * TN + j) * 32 + 2 * q;
        const int mc = min(m, MM - 2), nc = min(n, Ntot - 1);
        const int py = (mc >> 1) / G::CB, cb = (mc >> 1) % G::CB;
        const int img = nc / PSC, pix = nc % PSC;
        const int cx = pix % NX, y = 2 * (pix / NX) + py, x = 2 * cx;
        const bool ok = m < MM && n < Ntot;  // (row y may be off the output while the pair's second pixel is not)
        eo[i][j][pass] = ok ? 4u * (unsigned)(((img * G::CB + cb) * G::HB + y) * G::WB + x) : kOobOffset;
        if (ok && y < G::HB && cx + 1 < NX && x + 3 < G::WB && n + 1 < Ntot) evec |= 1u << ((i * TN + j) * 4 + pass);
        a4[i][j][pass] = f32x4{1.f, 1.f, 1.f, 1.f};
      }
  static_assert(TM * TN * 4 <= 32, "evec bits");
  auto aux_prefetch = [&]() __attribute__((always_inline)) {
    if (p.epi == REPO_EPI_MUL_DRELU) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int pass = 0; pass < 4; ++pass)
            if ((evec >> ((i * TN + j) * 4 + pass)) & 1u)
              a4[i][j][pass] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, eo[i][j][pass], 0, 0));
    }
  };

  __syncthreads();  // the zeroed borders are in place before the first staging store
  gload(0);
  lstore(0);
  __syncthreads();
  if (NSL == 1) {
    aux_prefetch();
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
  } else {
    int buf = 0;
    for (int t = 0; t + 1 < NSL; ++t) {
      gload(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(buf);
      __builtin_amdgcn_sched_barrier(0);
      lstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    aux_prefetch();
    __builtin_amdgcn_sched_barrier(0);
    compute(buf);
  }

  // ---- epilogue.  Accumulator rows 2a, 2a+1 of a tile are (py, cb, px = 0 / 1): with two horizontally adjacent class
  // pixels they are four consecutive floats of output row 2 cy + py.  Each wave turns its 32 x 32 tile through a
  // private LDS strip (the K loop's buffers are free by now): item = (row pair a, pixel pair q), 16 x 16 items per tile,
  // the 16 lanes of a row pair write 256 contiguous bytes.
  __syncthreads();
#ifdef REPO_UP_SKIP_EPI
  if (p.nimg > 0) {
    float sacc = 0.f;
    for (int i = 0; i < TM; ++i)
      for (int j = 0; j < TN; ++j)
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 12345.678f) p.out[0] = sacc;
    return;
  }
#endif
  float* strip = lds + wid * 32 * EP;
  auto fin = [&](float x, float bv, float a) __attribute__((always_inline)) {
    x += bv;
    if (p.epi == REPO_EPI_RELU) x = fmaxf(x, 0.f);
    else if (p.epi == REPO_EPI_MUL_DRELU) x = a > 0.f ? x : 0.f;
    return x;
  };
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + li] = acc[i][j][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int a = pass * 4 + (lane >> 4), q = lane & 15;
        const float2 lo = *reinterpret_cast<const float2*>(strip + (2 * a) * EP + 2 * q);      // px = 0: pixels n, n+1
        const float2 hi = *reinterpret_cast<const float2*>(strip + (2 * a + 1) * EP + 2 * q);  // px = 1
        const unsigned o = eo[i][j][pass];
        if (o != kOobOffset) {
          const int m = m0 + (wm * TM + i) * 32 + 2 * a;
          const int cb = (m >> 1) % G::CB;
          const float bv = p.bias ? p.bias[cb] : 0.f;
          if ((evec >> ((i * TN + j) * 4 + pass)) & 1u) {
            const f32x4 av = a4[i][j][pass];
            f32x4 v = {fin(lo.x, bv, av[0]), fin(hi.x, bv, av[1]), fin(lo.y, bv, av[2]), fin(hi.y, bv, av[3])};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rout, o, 0, 0);
          } else {
            // a pair that runs over the end of a class row / the output row / the tensor: element by element
            const int n = n0 + (wn * TN + j) * 32 + 2 * q;
            const int py = (m >> 1) / G::CB;
            const float vals[4] = {lo.x, hi.x, lo.y, hi.y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ne = n + (e >> 1);
              if (ne < Ntot) {
                const int ie = ne / PSC, pe = ne % PSC;
                const int ye = 2 * (pe / NX) + py, xe = 2 * (pe % NX) + (e & 1);
                if (ye < G::HB && xe < G::WB) {
                  const int oe = ((ie * G::CB + cb) * G::HB + ye) * G::WB + xe;
                  p.out[oe] = fin(vals[e], bv, p.epi == REPO_EPI_MUL_DRELU ? p.aux[oe] : 1.f);
                }
              }
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
}

template <class G>
inline int launch_dconv_up_pack(const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
  using U = UpGeo<G>;
  if (!ws || ws_bytes < U::PACK_FLOATS * sizeof(float)) return REPO_E_WS_TOO_SMALL;
  UpPackArgs a{w, (float*)ws};
  const int blocks = (int)((U::PACK_FLOATS + 255) / 256);
  hipLaunchKernelGGL((dconv_up_pack_kernel<G>), dim3(blocks > 512 ? 512 : blocks), dim3(256), 0, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// `packed` != 0: ws already holds this layer's pack (repo_conv_up_pack); else it is written first.
template <class G, class T>
inline int launch_dconv_up(const float* small, const float* w, const float* bias, const float* aux, float* out,
                           int64_t nimg, int epi, int packed, void* ws, size_t ws_bytes, hipStream_t s) {
  using U = UpGeo<G>;
  if (!ws || ws_bytes < U::PACK_FLOATS * sizeof(float)) return REPO_E_WS_TOO_SMALL;
  if (!packed) {
    const int rc = launch_dconv_up_pack<G>(w, ws, ws_bytes, s);
    if (rc) return rc;
  }
  UpArgs a{small, (const float*)ws, bias, aux, out, (int)nimg, epi,
           (unsigned)(nimg * G::CS * G::PS * sizeof(float)), (unsigned)(U::PACK_FLOATS * sizeof(float))};
  const long gx = (nimg * U::PSC + T::BN - 1) / T::BN, gy = (U::M + T::BM - 1) / T::BM;
  hipLaunchKernelGGL((dconv_up_kernel<G, T>), dim3((unsigned)gx, (unsigned)gy), dim3(T::NT), 0, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

}  // namespace repo
