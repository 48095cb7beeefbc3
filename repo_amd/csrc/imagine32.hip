// Persistent imagination rollout on 32-row tiles and the bf16 matrix pipe (rowtile32.h); same inputs, outputs and saved
// activations as the 16-row fp32-MFMA engine of imagine16.hip, fp32-level accuracy (every product is the exact
// three-way bf16 split's six partial products, accumulated in fp32).
//
// One workgroup (8 waves) owns 32 start states for the whole rollout: 77 workgroups at 2450 rows.  Per step: the actor
// trunk (4 ELU layers + head), the tanh-Normal sample, fc_embed_state_action, the GRU and the prior head, back to back;
// activations live in LDS as three bf16 planes (split once, by the epilogue that produces them), weights stream from
// L2 as pre-split fragment-ready packs, a wave owns ONE 32-column tile of a layer (7 tiles of a 200-wide layer on 8
// waves).  The GRU keeps the four gate accumulators of its column tile in one wave; the new belief waits in registers
// for the barrier behind which nobody reads the old one (one feature tile instead of two: LDS is what bounds the tile).
//
// Reference: TransitionModel.imagine + ActorModel.get_action (models/rssm.py:148-184, models/actor_critic.py:76-102).
#include <atomic>
#include <type_traits>

#include "rowtile32.h"

namespace repo {

constexpr int kPD32 = 4;
#ifdef RT_NO_STORE  // ablation build
#define GST32(...)
#else
#define GST32 __builtin_amdgcn_raw_buffer_store_b128
#endif
typedef WWin32<kPD32> Win32;

struct Img32FwdArgs {
  int Hm, N, A, D, Hd, S, C;
  const char* wpack;
  unsigned wbytes;
  unsigned aW[5], aB[5];
  unsigned Wsa, Wg[3], Wbp, Wsp;   // Wg[g]: [W_ih | W_hh] of gate g along K (each padded to BW blocks)
  unsigned Bsa, Br, Bz, Bin, Bhn, Bbp, Bsp;  // Br / Bz: b_ih + b_hh of the gate (summed by the pack kernel)
  const float *belief0, *state0, *cond;
  NoiseSrc eps_act, eps_prior;
  float min_std, a_min_std, a_init_std, a_mean_scale;
  float *featx, *prior_mean, *prior_std, *a_hidden, *a_raw, *a_mean, *a_std, *xsa, *e, *gates, *hp;
  size_t a_layer_rows;
  unsigned gates_bytes;
};

template <int NBLK>
__device__ __forceinline__ void dopen32(Win32& w, __amdgpu_buffer_rsrc_t rw, unsigned W, unsigned B, int N, int wave,
                                        int lane) {
  const int nt = (N + 31) >> 5;
  wopen32<NBLK, kPD32>(w, rw, wave < nt, W, B, N, min(wave, nt - 1), lane);
}

// BF / BW / BX: 16-k blocks of F = D + S (+ C), of D and Hd, of X = S + A (+ C)
template <int BF, int BW, int BX>
__global__ __launch_bounds__(512) void imagine32_fwd_kernel(Img32FwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char lds32[];
  const int Hm = p.Hm, N = p.N, A = p.A, D = p.D, Hd = p.Hd, S = p.S, C = p.C;
  const int F = D + S, X = S + A, XL = X + C;
  constexpr int psF = 1024 * BF, psH = 1024 * BW, psX = 1024 * BX;
  char* Fp = lds32;            // [belief | state | condition]
  char* HA = Fp + 3 * psF;
  char* HB = HA + 3 * psH;
  char* XS = HB + 3 * psH;     // [state | action | condition]
  float* SM = reinterpret_cast<float*>(XS + 3 * psX);  // fp32 [64 columns][32 rows]: head outputs
  constexpr int lds_bytes = 3 * psF + 6 * psH + 3 * psX + 64 * kR32 * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * kR32;
  const int nr = min(kR32, N - r0);
  const size_t rowsAll = p.a_layer_rows;
  const __amdgpu_buffer_rsrc_t rw = wrsrc(reinterpret_cast<const float*>(p.wpack), p.wbytes);
  // the gate tensors leave from the accumulators: a lane without a row stores out of range (dropped by the buffer's
  // range check: no branch, no exec mask around the stores)
  const __amdgpu_buffer_rsrc_t q_gates = wrsrc(p.gates, p.gates_bytes);
  const int xf = xfrag32(lane);
  const int c0 = wave * 32 + 4 * lh;  // first column of this lane's quad 0 in a layer tile
  const unsigned gvoff = li < nr ? 4u * ((unsigned)(r0 + li) * 4u * (unsigned)D + (unsigned)c0) : 0x80000000u;

  for (int i = tid; i < lds_bytes / 16; i += 512) reinterpret_cast<f32x4v*>(lds32)[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int i = tid; i < kR32 * F; i += 512) {
    const int row = i / F, f = i % F;
    if (row < nr) {
      const float v = f < D ? p.belief0[(size_t)(r0 + row) * D + f] : p.state0[(size_t)(r0 + row) * S + (f - D)];
      p.featx[(size_t)(r0 + row) * F + f] = v;
      st1_32(Fp, psF, f, row, v);
    }
  }
  // the condition: nothing below writes a column >= F of the feature tile
  for (int i = tid; i < kR32 * C; i += 512) {
    const int row = i / C, c = i % C;
    if (row < nr) st1_32(Fp, psF, F + c, row, p.cond[(size_t)(r0 + row) * C + c]);
  }
  __syncthreads();

  // A 200-wide ELU layer: the accumulator starts from the bias (requested with the weight window, a layer ahead); the
  // epilogue splits the activations into the next layer's planes.  A quad exists in the tile iff its first column is
  // below the tile's padded width: a wave-uniform test (columns [N, width) hold elu(0) = 0 and meet zero weights).
  auto dense = [&](auto nb, const char* Xt, int psx, Win32& w, char* dst) __attribute__((always_inline)) {
    if (w.act) {
      f32x16v c = bias_acc(w);
      wrun32<decltype(nb)::value, kPD32>(c, Xt + xf, psx, w, rw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wave * 32 + 8 * i < 16 * BW) {
          const f32x4v a = quad(c, i);
          f32x4v v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = elu(a[r]);
          stq32(dst, psH, c0 + 8 * i, li, v);
        }
      }
    }
  };
  // a narrow head: fp32 values into SM (64 columns: every quad of the one or two tiles fits)
  auto head = [&](Win32& w, const char* Xt) __attribute__((always_inline)) {
    if (w.act) {
      f32x16v c = bias_acc(w);
      wrun32<BW, kPD32>(c, Xt + xf, psH, w, rw);
#pragma unroll
      for (int e = 0; e < 16; ++e) SM[(c0 + 8 * (e >> 2) + (e & 3)) * kR32 + li] = c[e];
    }
  };
  // The saved fp32 copy of a tile is written by the store wave (the eighth wave has no column tile in a 200-wide
  // layer) behind the layer's barrier, while the other waves run the next layer.
  const bool swave = wave == kW - 1;
  auto save_tile = [&](const char* T, int ps, int ncols, const float* dst, unsigned ld, size_t first_row)
                       __attribute__((always_inline)) {
    if (swave) store_tile32(T, ps, ncols, dst + first_row * ld, (unsigned)r0, nr, ld, 0u, lane);
  };
  // ... and behind a NARROW layer (the actor's head: one column tile, the prior head's output: two) every wave without a
  // tile takes a share: one wave's 1.5 us would outlast the layer
  auto save_tile_shared = [&](const char* T, int ps, int ncols, const float* dst, unsigned ld, size_t first_row,
                              int busy) __attribute__((always_inline)) {
    if (wave >= busy) store_tile32(T, ps, ncols, dst + first_row * ld, (unsigned)r0, nr, ld, 0u, lane, wave - busy, kW - busy);
  };
  const std::integral_constant<int, BF> nbF{};
  const std::integral_constant<int, BW> nbW{};
  const std::integral_constant<int, BX> nbX{};

  const int ntD = (D + 31) >> 5;
  const bool gact = wave < ntD;
  const int gtile = min(wave, ntD - 1);
  Win32 wa, wb;
  dopen32<BF>(wa, rw, p.aW[0], p.aB[0], Hd, wave, lane);
  for (int t = 0; t < Hm; ++t) {
    const size_t rb = (size_t)t * N + r0;
    const size_t tN = (size_t)t * N;
    // ---------------- actor trunk
    dopen32<BW>(wb, rw, p.aW[1], p.aB[1], Hd, wave, lane);
    dense(nbF, Fp, psF, wa, HA);
    lds_barrier();
    save_tile(HA, psH, Hd, p.a_hidden, (unsigned)Hd, 0 * rowsAll + tN);
    dopen32<BW>(wa, rw, p.aW[2], p.aB[2], Hd, wave, lane);
    dense(nbW, HA, psH, wb, HB);
    lds_barrier();
    save_tile(HB, psH, Hd, p.a_hidden, (unsigned)Hd, 1 * rowsAll + tN);
    dopen32<BW>(wb, rw, p.aW[3], p.aB[3], Hd, wave, lane);
    dense(nbW, HB, psH, wa, HA);
    lds_barrier();
    save_tile(HA, psH, Hd, p.a_hidden, (unsigned)Hd, 2 * rowsAll + tN);
    dopen32<BW>(wa, rw, p.aW[4], p.aB[4], 2 * A, wave, lane);
    dense(nbW, HA, psH, wb, HB);
    lds_barrier();
    save_tile_shared(HB, psH, Hd, p.a_hidden, (unsigned)Hd, 3 * rowsAll + tN, 1);   // beside the head (wave 0)
    dopen32<BX>(wb, rw, p.Wsa, p.Bsa, D, wave, lane);
    head(wa, HB);
    lds_barrier();
    // ---------------- tanh-Normal action sample; x = [state, action, condition]
    for (int i = tid; i < kR32 * XL; i += 512) {
      const int row = i / XL, k = i % XL;
      float v;
      if (k < S) {
        v = ld1_32(Fp, psF, D + k, row);
      } else if (k >= X) {
        v = ld1_32(Fp, psF, F + k - X, row);
      } else {
        const int a = k - S;
        const float raw_m = SM[a * kR32 + row], raw_s = SM[(A + a) * kR32 + row];
        const float mu = p.a_mean_scale * tanh_fast(raw_m / p.a_mean_scale);
        const float sd = softplus(raw_s + p.a_init_std) + p.a_min_std;
        const float ep = row < nr ? p.eps_act.at((rb + row) * A + a) : 0.f;
        v = tanh_fast(fmaf(sd, ep, mu));
        if (row < nr) {
          p.a_mean[(rb + row) * A + a] = mu;
          p.a_std[(rb + row) * A + a] = sd;
          p.a_raw[(rb + row) * 2 * A + a] = raw_m;
          p.a_raw[(rb + row) * 2 * A + A + a] = raw_s;
        }
      }
      st1_32(XS, psX, k, row, v);
      if (row < nr) p.xsa[(rb + row) * XL + k] = v;
    }
    lds_barrier();
    // ---------------- e = elu(W_sa x + b); the GRU's first stream is opened behind it
    wopen32<2 * BW, kPD32>(wa, rw, gact, p.Wg[0], p.Br, D, gtile, lane);
    dense(nbX, XS, psX, wb, HA);
    lds_barrier();
    save_tile(HA, psH, D, p.e, (unsigned)D, tN);
    // ---------------- GRU: one column tile per wave, gate by gate (three live accumulators at most).  A gate's two
    //                  products are ONE stream: the pack is [W_ih | W_hh] along K, the activations [e | belief]; the
    //                  n gate's stay apart (n = tanh(gi + r * gh)): its pack is read as two streams of BW blocks
    f32x16v hnew = zero16();
    {
      const unsigned gs = (unsigned)(4u * tN * 4 * D);
      f32x16v rg, gin, ghn;
      wopen32<BW, kPD32>(wb, rw, gact, p.Wg[2], p.Bin, D, gtile, lane);
      if (gact) {
        rg = bias_acc(wa);
        wrun32<2 * BW, kPD32, BW>(rg, HA + xf, psH, wa, rw, Fp + xf, psF);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4v v = quad(rg, i);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = sigmoidf(v[r]);
          set_quad(rg, i, v);
          if (wave * 32 + 8 * i < D)
            GST32(__builtin_bit_cast(u32x4v, v), q_gates, gvoff + 32u * i, gs, 0);
        }
      }
      wopen32<BW, kPD32>(wa, rw, gact, p.Wg[2] + BW * 96u * (unsigned)pad32(D), p.Bhn, D, gtile, lane);
      if (gact) {
        gin = bias_acc(wb);
        wrun32<BW, kPD32>(gin, HA + xf, psH, wb, rw);
      }
      wopen32<2 * BW, kPD32>(wb, rw, gact, p.Wg[1], p.Bz, D, gtile, lane);
      if (gact) {
        ghn = bias_acc(wa);
        wrun32<BW, kPD32>(ghn, Fp + xf, psF, wa, rw);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4v gh = quad(ghn, i), r_ = quad(rg, i);
          f32x4v ng = quad(gin, i);
#pragma unroll
          for (int r = 0; r < 4; ++r) ng[r] = tanh_fast(ng[r] + r_[r] * gh[r]);
          set_quad(gin, i, ng);
          if (wave * 32 + 8 * i < D) {
            GST32(__builtin_bit_cast(u32x4v, gh), q_gates, gvoff + 32u * i,
                                                   gs + 12u * D, 0);
            GST32(__builtin_bit_cast(u32x4v, ng), q_gates, gvoff + 32u * i,
                                                   gs + 8u * D, 0);
          }
        }
      }
      dopen32<BW>(wa, rw, p.Wbp, p.Bbp, Hd, wave, lane);  // the prior head's first layer
      if (gact) {
        f32x16v az = bias_acc(wb);
        wrun32<2 * BW, kPD32, BW>(az, HA + xf, psH, wb, rw, Fp + xf, psF);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (wave * 32 + 8 * i < D) {
            f32x4v zg = quad(az, i);
#pragma unroll
            for (int r = 0; r < 4; ++r) zg[r] = sigmoidf(zg[r]);
            GST32(__builtin_bit_cast(u32x4v, zg), q_gates, gvoff + 32u * i,
                                                   gs + 4u * D, 0);
            const f32x4v hprev = ldq32(Fp, psF, c0 + 8 * i, li);
            const f32x4v ng = quad(gin, i);
            f32x4v hn;
#pragma unroll
            for (int r = 0; r < 4; ++r) hn[r] = (1.f - zg[r]) * ng[r] + zg[r] * hprev[r];
            set_quad(hnew, i, hn);
          }
        }
      }
    }
    lds_barrier();  // nobody reads the old belief any more
    if (gact) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (wave * 32 + 8 * i < D) stq32(Fp, psF, c0 + 8 * i, li, quad(hnew, i));
    }
    lds_barrier();
    save_tile(Fp, psF, D, p.featx, (unsigned)F, tN + N);  // the new belief: columns [0, D) of featx[t + 1]
    // ---------------- prior head (its first layer reads the belief columns of the feature tile: the rows of the
    //                  pack behind K = D are zero)
    dopen32<BW>(wb, rw, p.Wsp, p.Bsp, 2 * S, wave, lane);
    dense(nbW, Fp, psF, wa, HB);
    lds_barrier();
    save_tile_shared(HB, psH, Hd, p.hp, (unsigned)Hd, tN, 2);   // beside the prior head's output layer (waves 0, 1)
    dopen32<BF>(wa, rw, p.aW[0], p.aB[0], Hd, wave, lane);  // the next step's first layer
    head(wb, HB);
    lds_barrier();
    for (int i = tid; i < kR32 * S; i += 512) {
      const int row = i / S, s = i % S;
      const float mu = SM[s * kR32 + row];
      const float sd = softplus(SM[(S + s) * kR32 + row]) + p.min_std;
      float smp = mu;
      if (row < nr) {
        const size_t o = (rb + row) * S + s;
        smp = fmaf(sd, p.eps_prior.at(o), mu);
        p.prior_mean[o] = mu;
        p.prior_std[o] = sd;
        p.featx[((size_t)(t + 1) * N + r0 + row) * F + D + s] = smp;
      }
      st1_32(Fp, psF, D + s, row, smp);
    }
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------ reverse pass
// The reverse rollout through the frozen world model (dreamer.py:357) for the same 32 rows per workgroup.  Gradient
// CARRIES (d belief, d state) stay fp32 in LDS; what multiplies a weight matrix is written as bf16 planes by the phase
// that produces it.  LDS holds two 208-column plane tiles: the four gate deltas pass through them in two rounds
// (r, z; then n * r, n), the accumulators of the two products (into belief_t and into e) wait in registers in between.
constexpr int kPDb = 3;   // (the epilogues' operands -- saved gates, belief_t -- are requested ahead too: registers)
typedef WWin32<kPDb> WinB;
struct Img32BwdArgs {
  int Hm, N, A, D, Hd, S, C;
  const char* wpack;   // transposed packs: W'(n = input index, k = output index)
  unsigned wbytes;
  unsigned Wsp, Wbp, Whh_rz, Wih_rz, Whh_n, Wih_n, Wsa;
  NoiseSrc eps_act, eps_prior;
  float min_std, a_min_std, a_mean_scale;
  const float *featx, *prior_std, *a_mean, *a_std, *xsa, *e, *gates, *hp;
  const float *dfeat, *dprior_mean, *dprior_std;
  float *d_araw, *dfeat0;
};

template <int BW, int BS2>   // BS2: 16-k blocks of 2 S
__global__ __launch_bounds__(512) void imagine32_bwd_kernel(Img32BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char lds32[];
  const int Hm = p.Hm, N = p.N, A = p.A, D = p.D, Hd = p.Hd, S = p.S;
  const int F = D + S, X = S + A, XL = X + p.C;
  constexpr int psH = 1024 * BW, psS = 1024 * BS2;
  constexpr int WC = 16 * BW;                         // columns of a wide tile
  char* XA = lds32;
  char* XB = XA + 3 * psH;
  char* SMp = XB + 3 * psH;                           // planes: d [prior mean | raw std]
  float* Gb = reinterpret_cast<float*>(SMp + 3 * psS);  // fp32 quads [WC x 32]: d belief carry
  float* Gs = Gb + WC * kR32;                         // [32 x 32]: d state
  float* SMa = Gs + 32 * kR32;                        // [48 x 32]: d [state | action] of this step
  constexpr int lds_bytes = 6 * psH + 3 * psS + (WC + 32 + 48) * kR32 * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * kR32;
  const int nr = min(kR32, N - r0);
  const __amdgpu_buffer_rsrc_t rw = wrsrc(reinterpret_cast<const float*>(p.wpack), p.wbytes);
  const __amdgpu_buffer_rsrc_t q_hp = arsrc(p.hp), q_e = arsrc(p.e), q_gates = arsrc(p.gates), q_featx = arsrc(p.featx);
  const __amdgpu_buffer_rsrc_t q_dfeat =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dfeat), 0, 0x7fffffff, 0x00020000);  // 2 GB window: 0x80000000 reads 0
  const unsigned lrow = (unsigned)(r0 + min(li, nr - 1));
  const int xf = xfrag32(lane);
  const int c0 = wave * 32 + 4 * lh;

  for (int i = tid; i < lds_bytes / 16; i += 512) reinterpret_cast<f32x4v*>(lds32)[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int ntD = (D + 31) >> 5, ntH = (Hd + 31) >> 5, ntX = (X + 31) >> 5;
  // three windows: W_sp's (4 blocks) has one of its own, the other six streams of a step alternate between two
  WinB wa, wb, wc;
  wopen32<BS2, kPDb, false>(wc, rw, wave < ntH, p.Wsp, 0, Hd, min(wave, ntH - 1), lane);
  const unsigned lrow_inv = lrow;
  const int c0_inv = c0, tid_inv = tid;
  for (int t = Hm - 1; t >= 0; --t) {
    const size_t rb = (size_t)t * N + r0;
    const unsigned tN = (unsigned)(t * N);
    // per-step copies the compiler cannot see through: the dozens of global byte offsets derived from them are then
    // recomputed where they are used (a multiply-add each) instead of being hoisted out of the step loop, spilled to
    // scratch memory and re-loaded -- one exposed memory round trip per address -- in front of every use
    unsigned lrow = lrow_inv;
    int c0 = c0_inv, tid = tid_inv;
    asm volatile("" : "+v"(lrow), "+v"(c0), "+v"(tid));
    // ---- G += dfeat[t]; the prior head's sample / mean / std gradients -> d [mean | raw_std].  Everything the two
    //      phases read from global memory is requested first (one exposed round trip instead of one per loop
    //      iteration); a thread takes quads of dfeat with consecutive lanes on consecutive rows (LDS: contiguous)
    {
      constexpr int NQ = 4;  // quads of dfeat per thread: 32 rows x (F / 4 rounded up) <= 2048
      f32x4v dq[NQ];
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int u = tid + 512 * j, row = u & 31, qc = u >> 5;
        const bool ok = 4 * qc < F && row < nr;
        dq[j] = bldq(q_dfeat, ok ? 4u * ((unsigned)(r0 + row) * F + 4u * qc) : 0x80000000u, 4u * tN * F);
      }
      float pstd[2], pe[2], pdm[2], pds[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = tid + 512 * j, row = u & 31, sidx = u >> 5;
        const bool ok = sidx < S && row < nr;
        const size_t o = ok ? (rb + row) * S + sidx : 0;
        pstd[j] = ok ? p.prior_std[o] : p.min_std;
        pe[j] = ok ? p.eps_prior.at(o) : 0.f;
        pdm[j] = (ok && p.dprior_mean) ? p.dprior_mean[o] : 0.f;
        pds[j] = (ok && p.dprior_std) ? p.dprior_std[o] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int u = tid + 512 * j, row = u & 31, qc = u >> 5;
        if (4 * qc < F) {
          float* g = 4 * qc < D ? Gb + fq(4 * qc, row) : Gs + fq(4 * qc - D, row);
          f32x4v v = *reinterpret_cast<f32x4v*>(g);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (4 * qc + e < F) ? dq[j][e] : 0.f;  // (a quad straddling the row's end)
          *reinterpret_cast<f32x4v*>(g) = v;
        }
      }
      lds_barrier();
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = tid + 512 * j, row = u & 31, sidx = u >> 5;
        if (sidx < S) {
          const float ds = row < nr ? Gs[fq(sidx, row)] : 0.f;
          const float gm = ds + pdm[j];
          const float gr = fmaf(ds, pe[j], pds[j]) * (-expm1f(-(pstd[j] - p.min_std)));
          st1_32(SMp, psS, sidx, row, row < nr ? gm : 0.f);
          st1_32(SMp, psS, S + sidx, row, row < nr ? gr : 0.f);
        }
      }
    }
    lds_barrier();
    // ---- XB = (d out @ W_sp) * elu'(hp)
    wopen32<BW, kPDb, false>(wa, rw, wave < ntD, p.Wbp, 0, D, min(wave, ntD - 1), lane);
    if (wc.act) {
      f32x4v h[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) h[i] = bldq(q_hp, 4u * (lrow * Hd + (unsigned)min(c0 + 8 * i, Hd - 4)), 4u * tN * Hd);
      f32x16v c = zero16();
      wrun32<BS2, kPDb>(c, SMp + xf, psS, wc, rw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wave * 32 + 8 * i < WC) {
          const f32x4v a = quad(c, i);
          f32x4v v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (c0 + 8 * i < Hd) ? a[r] * elu_grad_from_out(h[i][r]) : 0.f;
          stq32(XB, psH, c0 + 8 * i, li, v);
        }
      }
    }
    lds_barrier();
    // ---- d belief_{t+1} = Gb + XB @ W_bp, and the GRU's gate deltas from it in the same registers (a lane's quads:
    //      4 columns of its row): g_r -> XA at once, g_z -> XB behind the barrier (XB is this product's operand);
    //      g_n and g_n * r wait in registers for the second round; Gb <- d * z.  The saved gates and belief_t are
    //      requested before the MFMA chain.
    f32x16v gz16 = zero16(), gn16 = zero16(), gnr16 = zero16();
    wopen32<2 * BW, kPDb, false>(wb, rw, wave < ntD, p.Whh_rz, 0, D, min(wave, ntD - 1), lane);
    if (wa.act) {
      f32x4v rq[4], zq[4], nq[4], hq[4], pq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned col = (unsigned)min(c0 + 8 * i, D - 4);
        const unsigned gv = 4u * (lrow * 4 * D + col), gs = 4u * tN * 4 * D;
        rq[i] = bldq(q_gates, gv, gs);
        zq[i] = bldq(q_gates, gv + 4u * D, gs);
        nq[i] = bldq(q_gates, gv + 8u * D, gs);
        hq[i] = bldq(q_gates, gv + 12u * D, gs);
        pq[i] = bldq(q_featx, 4u * (lrow * F + col), 4u * tN * F);
      }
      f32x16v c = zero16();
      wrun32<BW, kPDb>(c, XB + xf, psH, wa, rw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wave * 32 + 8 * i < WC) {
          const int o = fq(c0 + 8 * i, li);
          const bool in = c0 + 8 * i < D && li < nr;
          const f32x4v d = quad(c, i) + *reinterpret_cast<const f32x4v*>(Gb + o);
          f32x4v g_r, g_z, g_n, g_nr, dhp;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float rg = rq[i][r], zg = zq[i][r], ng = nq[i][r];
            g_n[r] = in ? d[r] * (1.f - zg) * (1.f - ng * ng) : 0.f;
            g_z[r] = in ? d[r] * (pq[i][r] - ng) * zg * (1.f - zg) : 0.f;
            g_r[r] = g_n[r] * hq[i][r] * rg * (1.f - rg);
            g_nr[r] = g_n[r] * rg;
            dhp[r] = in ? d[r] * zg : 0.f;
          }
          stq32(XA, psH, c0 + 8 * i, li, g_r);
          *reinterpret_cast<f32x4v*>(Gb + o) = dhp;
          set_quad(gz16, i, g_z);
          set_quad(gn16, i, g_n);
          set_quad(gnr16, i, g_nr);
        }
      }
    }
    lds_barrier();  // every wave has read XB
    if (wa.act) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (wave * 32 + 8 * i < WC) stq32(XB, psH, c0 + 8 * i, li, quad(gz16, i));
    }
    lds_barrier();
    // ---- through W_hh into belief_t (ah) and through W_ih into e (ae): [r | z] first ...
    f32x16v ah = zero16(), ae = zero16();
    wopen32<2 * BW, kPDb, false>(wa, rw, wave < ntD, p.Wih_rz, 0, D, min(wave, ntD - 1), lane);
    if (wb.act) wrun32<2 * BW, kPDb, BW>(ah, XA + xf, psH, wb, rw, XB + xf, psH);
    wopen32<BW, kPDb, false>(wb, rw, wave < ntD, p.Whh_n, 0, D, min(wave, ntD - 1), lane);
    if (wa.act) wrun32<2 * BW, kPDb, BW>(ae, XA + xf, psH, wa, rw, XB + xf, psH);
    wopen32<BW, kPDb, false>(wa, rw, wave < ntD, p.Wih_n, 0, D, min(wave, ntD - 1), lane);
    lds_barrier();  // every wave has read g_r, g_z
    // ---- ... second round: XA <- g_n * r, XB <- g_n
    if (wa.act) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wave * 32 + 8 * i < WC) {
          stq32(XA, psH, c0 + 8 * i, li, quad(gnr16, i));
          stq32(XB, psH, c0 + 8 * i, li, quad(gn16, i));
        }
      }
    }
    lds_barrier();
    {
      f32x4v ev[4];
      if (wb.act) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ev[i] = bldq(q_e, 4u * (lrow * D + (unsigned)min(c0 + 8 * i, D - 4)), 4u * tN * D);
        wrun32<BW, kPDb>(ah, XA + xf, psH, wb, rw);
      }
      wopen32<BW, kPDb, false>(wb, rw, wave < ntX, p.Wsa, 0, X, min(wave, ntX - 1), lane);
      if (wa.act) wrun32<BW, kPDb>(ae, XB + xf, psH, wa, rw);
      lds_barrier();  // every wave has read the gate tiles
      if (wa.act) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (wave * 32 + 8 * i < WC) {
            const int o = fq(c0 + 8 * i, li);
            *reinterpret_cast<f32x4v*>(Gb + o) = *reinterpret_cast<const f32x4v*>(Gb + o) + quad(ah, i);  // new carry
            const f32x4v a = quad(ae, i);
            f32x4v v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (c0 + 8 * i < D) ? a[r] * elu_grad_from_out(ev[i][r]) : 0.f;
            stq32(XA, psH, c0 + 8 * i, li, v);  // d pre-activation of fc_embed_state_action
          }
        }
      }
    }
    lds_barrier();
    // ---- d [state_t | action_t] = XA @ W_sa (the sample backward's operands are requested ahead of it)
    wopen32<BS2, kPDb, false>(wc, rw, wave < ntH, p.Wsp, 0, Hd, min(wave, ntH - 1), lane);  // the next (earlier) step's first layer
    const int prow = tid & 31, pa_ = tid >> 5;
    const bool pok = pa_ < A && prow < nr;
    const size_t po = pok ? (rb + prow) * A + pa_ : 0;
    const float p_act = pok ? p.xsa[(rb + prow) * XL + S + pa_] : 0.f;
    const float p_mean = pok ? p.a_mean[po] : 0.f, p_std = pok ? p.a_std[po] : p.a_min_std;
    const float p_eps = pok ? p.eps_act.at(po) : 0.f;
    if (wb.act) {
      f32x16v c = zero16();
      wrun32<BW, kPDb>(c, XA + xf, psH, wb, rw);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = c0 + 8 * (e >> 2) + (e & 3);
        if (n < S) Gs[fq(n, li)] = c[e];
        else if (n < X) SMa[fq(n, li)] = c[e];
      }
    }
    lds_barrier();
    // ---- tanh-Normal sample backward -> gradient at the actor trunk's output of step t
    if (pok) {
      const float du = SMa[fq(S + pa_, prow)] * (1.f - p_act * p_act);
      const float tm = p_mean / p.a_mean_scale;
      p.d_araw[(rb + prow) * 2 * A + pa_] = du * (1.f - tm * tm);
      p.d_araw[(rb + prow) * 2 * A + A + pa_] = du * p_eps * (-expm1f(-(p_std - p.a_min_std)));
    }
    lds_barrier();
  }
  if (p.dfeat0) {
    for (int i = tid; i < kR32 * F; i += 512) {
      const int row = i / F, f = i % F;
      if (row < nr) p.dfeat0[(size_t)(r0 + row) * F + f] = f < D ? Gb[fq(f, row)] : Gs[fq(f - D, row)];
    }
  }
}

// ------------------------------------------------------------------------------------------ host side
constexpr int kBF32 = 15, kBW32 = 13, kBX32 = 3;
#ifndef RT32_DEFAULT
#define RT32_DEFAULT 1
#endif
static thread_local int t_rowtile32_enabled = RT32_DEFAULT;   // thread-local: see api.hip
bool rowtile32_enabled() { return t_rowtile32_enabled != 0; }

bool imagine32_ok(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, int n_actor_layers, int64_t C) {
  auto blk = [](int64_t k) { return pad16((int)k) >> 4; };
  return rowtile32_enabled() && n_actor_layers == 5 && D % 4 == 0 && Hd % 4 == 0 && D == Hd && blk(D) == kBW32 &&
         blk(D + S) == kBF32 && blk(D + S + C) == kBF32 && blk(S + A) == kBX32 && blk(S + A + C) == kBX32 && C >= 0 &&
         2 * S <= 64 && S % 2 == 0 && 2 * A <= 32 && (Hm + 1) * N * 4 * D < kMaxIdx;
}

size_t imagine32_fwd_ws_bytes(int64_t A, int64_t D, int64_t Hd, int64_t S) {
  return pack32_bytes(Hd, D + S + 16) + 3 * pack32_bytes(Hd, Hd) + pack32_bytes(2 * A, Hd) + pack32_bytes(D, S + A + 16) +
         6 * pack32_bytes(D, D) + pack32_bytes(Hd, D) + pack32_bytes(2 * S, Hd) + 5 * vec32_bytes(Hd) +
         vec32_bytes(2 * A) + 5 * vec32_bytes(D) + vec32_bytes(2 * S) + 256;
}

int imagine32_fwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, const float* const* rp,
                  const float* const* ap, const float* belief0, const float* state0, const float* cond, int64_t C,
                  NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std, float a_init_std,
                  float a_mean_scale, float* featx, float* prior_mean, float* prior_std, float* a_hidden,
                  int64_t a_layer_rows, float* a_raw, float* a_mean, float* a_std, float* xsa, float* e, float* gates,
                  float* hp, void* ws, hipStream_t stream) {
  const int F = (int)(D + S + C), X = (int)(S + A + C);
  char* w = (char*)ws;
  char* const w_begin = w;
  Img32FwdArgs a;
  a.Hm = (int)Hm, a.N = (int)N, a.A = (int)A, a.D = (int)D, a.Hd = (int)Hd, a.S = (int)S, a.C = (int)C;
  a.cond = cond;
  Pack32Args pa;
  pa.njobs = 0;
  auto mat = [&](const float* src, int Nn, int K, int sn, int sk) {
    pa.job[pa.njobs++] = Pack32Job{src, w, Nn, K, sn, sk, 0, nullptr};
    const unsigned r = (unsigned)(w - w_begin);
    w += pack32_bytes(Nn, K);
    return r;
  };
  auto vec2 = [&](const float* src, const float* src2, int Nn) {
    pa.job[pa.njobs++] = Pack32Job{src, w, Nn, 0, 0, 0, 1, src2};
    const unsigned r = (unsigned)(w - w_begin);
    w += vec32_bytes(Nn);
    return r;
  };
  auto vec = [&](const float* src, int Nn) { return vec2(src, nullptr, Nn); };
  const int kin[5] = {F, (int)Hd, (int)Hd, (int)Hd, (int)Hd};
  const int nout[5] = {(int)Hd, (int)Hd, (int)Hd, (int)Hd, (int)(2 * A)};
  for (int l = 0; l < 5; ++l) {
    a.aW[l] = mat(ap[2 * l], nout[l], kin[l], kin[l], 1);
    a.aB[l] = vec(ap[2 * l + 1], nout[l]);
  }
  a.Wsa = mat(rp[0], (int)D, X, X, 1);
  for (int g = 0; g < 3; ++g) {  // [W_ih | W_hh] of a gate: two packs of the same column count back to back = one
    a.Wg[g] = mat(rp[2] + (size_t)g * D * D, (int)D, (int)D, (int)D, 1);   // stream of 2 BW blocks
    mat(rp[3] + (size_t)g * D * D, (int)D, (int)D, (int)D, 1);
  }
  a.Wbp = mat(rp[6], (int)Hd, (int)D, (int)D, 1);
  a.Wsp = mat(rp[8], (int)(2 * S), (int)Hd, (int)Hd, 1);
  a.Bsa = vec(rp[1], (int)D);
  a.Br = vec2(rp[4], rp[5], (int)D);
  a.Bz = vec2(rp[4] + D, rp[5] + D, (int)D);
  a.Bin = vec(rp[4] + 2 * D, (int)D);
  a.Bhn = vec(rp[5] + 2 * D, (int)D);
  a.Bbp = vec(rp[7], (int)Hd);
  a.Bsp = vec(rp[9], (int)(2 * S));
  a.wpack = w_begin;
  a.wbytes = (unsigned)(w - w_begin);
  int rc = launch_pack32(pa, stream);
  if (rc) return rc;
  a.belief0 = belief0, a.state0 = state0, a.eps_act = eps_act, a.eps_prior = eps_prior;
  a.min_std = min_std, a.a_min_std = a_min_std, a.a_init_std = a_init_std, a.a_mean_scale = a_mean_scale;
  a.featx = featx, a.prior_mean = prior_mean, a.prior_std = prior_std, a.a_hidden = a_hidden, a.a_raw = a_raw;
  a.a_mean = a_mean, a.a_std = a_std, a.xsa = xsa, a.e = e, a.gates = gates, a.hp = hp;
  a.a_layer_rows = (size_t)a_layer_rows;
  a.gates_bytes = (unsigned)((size_t)Hm * N * 4 * D * sizeof(float));
  constexpr int lds_b = 3 * 1024 * kBF32 + 6 * 1024 * kBW32 + 3 * 1024 * kBX32 + 64 * kR32 * 4;
  auto kern = imagine32_fwd_kernel<kBF32, kBW32, kBX32>;
  hipError_t he = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL(kern, dim3((unsigned)((N + kR32 - 1) / kR32)), dim3(512), lds_b, stream, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

size_t imagine32_bwd_ws_bytes(int64_t A, int64_t D, int64_t Hd, int64_t S) {
  return pack32_bytes(Hd, 2 * S) + pack32_bytes(D, Hd) + 6 * pack32_bytes(D, D) + pack32_bytes(S + A, D) + 256;
}

int imagine32_bwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, const float* const* rp, int64_t C,
                  NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std, float a_mean_scale,
                  const float* featx, const float* prior_std, const float* a_mean, const float* a_std, const float* xsa,
                  const float* e, const float* gates, const float* hp, const float* dfeat, const float* dprior_mean,
                  const float* dprior_std, float* d_araw, float* dfeat0, void* ws, hipStream_t stream) {
  const int X = (int)(S + A);
  char* w = (char*)ws;
  char* const w_begin = w;
  Img32BwdArgs a;
  a.Hm = (int)Hm, a.N = (int)N, a.A = (int)A, a.D = (int)D, a.Hd = (int)Hd, a.S = (int)S, a.C = (int)C;
  Pack32Args pa;
  pa.njobs = 0;
  // W'(n = input index, k = output index) = native[k * ld + n]
  auto mat = [&](const float* src, int Nin, int Kout, int ld) {
    pa.job[pa.njobs++] = Pack32Job{src, w, Nin, Kout, 1, ld, 0, nullptr};
    const unsigned r = (unsigned)(w - w_begin);
    w += pack32_bytes(Nin, Kout);
    return r;
  };
  a.Wsp = mat(rp[8], (int)Hd, (int)(2 * S), (int)Hd);
  a.Wbp = mat(rp[6], (int)D, (int)Hd, (int)D);
  a.Whh_rz = mat(rp[3], (int)D, (int)D, (int)D);                 // [r | z] along K: two packs back to back
  mat(rp[3] + (size_t)D * D, (int)D, (int)D, (int)D);
  a.Wih_rz = mat(rp[2], (int)D, (int)D, (int)D);
  mat(rp[2] + (size_t)D * D, (int)D, (int)D, (int)D);
  a.Whh_n = mat(rp[3] + (size_t)2 * D * D, (int)D, (int)D, (int)D);
  a.Wih_n = mat(rp[2] + (size_t)2 * D * D, (int)D, (int)D, (int)D);
  a.Wsa = mat(rp[0], X, (int)D, X + (int)C);  // the rows of d [state|action] only: the condition takes no gradient
  a.wpack = w_begin;
  a.wbytes = (unsigned)(w - w_begin);
  int rc = launch_pack32(pa, stream);
  if (rc) return rc;
  a.eps_act = eps_act, a.eps_prior = eps_prior;
  a.min_std = min_std, a.a_min_std = a_min_std, a.a_mean_scale = a_mean_scale;
  a.featx = featx, a.prior_std = prior_std, a.a_mean = a_mean, a.a_std = a_std, a.xsa = xsa, a.e = e;
  a.gates = gates, a.hp = hp, a.dfeat = dfeat, a.dprior_mean = dprior_mean, a.dprior_std = dprior_std;
  a.d_araw = d_araw, a.dfeat0 = dfeat0;
  constexpr int kBS2 = 4;
  constexpr int lds_b = 6 * 1024 * kBW32 + 3 * 1024 * kBS2 + (16 * kBW32 + 32 + 48) * kR32 * 4;
  auto kern = imagine32_bwd_kernel<kBW32, kBS2>;
  hipError_t he = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL(kern, dim3((unsigned)((N + kR32 - 1) / kR32)), dim3(512), lds_b, stream, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

}  // namespace repo

extern "C" int repo_debug_rowtile32(int enable) {
  const int prev = repo::t_rowtile32_enabled;
  repo::t_rowtile32_enabled = enable ? 1 : 0;
  return prev;
}
