// Column-split, WEIGHT-STATIONARY observe scan on the fp32 matrix cores (north_star: "MFMA ... for the
// dense hidden-to-hidden matmuls inside the GRU", "LDS staging of the deterministic/stochastic state across the
// L-step scan").
//
// rssm.hip's scan gives a workgroup 1-2 batch rows and streams ALL in-scan weights (1.2 MB) from L2 every step:
// 17-20 us per step whatever B is.  Here the weights never move: NW = ceil(D/16) workgroups each own 16 belief /
// hidden columns -- their slices of W_ih, W_hh (6 x D x 16), W_bq (D x 16) and the two small replicated layers (W_sa:
// (S+A) -> D, W_sq: Hd -> 2S) stay in REGISTERS for all T steps, each lane holding exactly the MFMA fragments it feeds
// (LDS holds activation tiles only, 57 / 67 KB, so the CU stays open to the convolutions of the other lanes: +0.9 % on the
// update against LDS-resident slices, which were 4 % faster alone) -- and every workgroup carries the same 16
// batch rows (one v_mfma_f32_16x16x4_f32 row tile; more rows = more independent groups of NW workgroups).  Per step:
//   A  e = elu(W_sa x + b)                    all D columns, replicated                     (13 tiles x 3 blocks)
//   B  GRU gates of the OWN 16 columns        K = D over e and over belief                  (6 x 13 blocks)
//   X1 all-gather of the new belief           own slice published, NW slices gathered
//   C  hq = elu(W_bq[:, :D] h + eemb + b)     own 16 columns, K split over the 4 waves      (13 blocks)
//   X2 all-gather of hq
//   D  posterior (mean, std) = W_sq hq + b    all 2S columns, replicated; sample            (4 tiles x 13 blocks)
// An all-gather is the hand-off form measured by tools/probe/scan_exchange.hip (MI355X_MICROARCH.md's table, row 1):
// ONE wave stores the slice with sc1 (write-through) 16-byte stores, waits vmcnt(0) and stores the step's epoch into
// the workgroup's flag (sc1); one lane per flag polls with sc1 loads; after a workgroup barrier all threads gather
// with sc1 16-byte loads.  2.6 us per exchange on an idle chip, 7-10 us beside a streaming kernel
// (profiles/r03_scan_exchange_probe.txt).  The forward scan replaced the flag hop by DATA-TAGGED granules (below:
// ~1.5 us per exchange).  With INDEPENDENT 16-row groups every exchange stays at the 16-row price, and the engine is
// the default for every B <= 64 (ops.py): a step costs 11 / 13.5 us instead of 17-20 for a 7-row shard of a
// strong-scaling job and, measured inside the N = 1 update at B = 50, 577 / 721 us per scan against 901 / 1722.
// Failure mode: the exchanges are cross-workgroup spin-waits and a plain launch does not guarantee co-residency of a
// group's 13 workgroups (DESIGN.md section 5 has the argument why they are resident in practice).  A poll that does
// not see its peers within `spin_limit` rounds (debug knob: repo_debug_scan_spin_limit) therefore gives up instead of
// hanging: it ORs REPO_SCAN_STATUS_{FWD,BWD}_TIMEOUT into the caller's sticky `status` word (an argument of
// repo_rssm_observe_fwd / _bwd; the agents read it back inside their ONE per-update scalar copy and raise
// RepoHipError), the whole group leaves the time loop, and NaN is written into its outputs (the last belief slot /
// the first d e row) so that nothing plausible is left behind.
//
// Prior head: not here (it is off the recurrence: repo_rssm_prior_head evaluates it for all steps at once).
// Everything the reverse scan needs is written exactly as rssm.hip's forward writes it.
// Reference: TransitionModel.observe, /root/reference/algorithms/repo/models/rssm.py:34-64,76-146.
#include <algorithm>

#include "rowtile.h"
#include "scan_cs.h"

namespace repo {

constexpr int kSC1 = 16;  // cache-policy bit of the raw buffer builtins on gfx940+: sc1

struct CsFwdArgs {
  int T, B, A, D, Hd, S;
  const float *Wsa, *Wih, *Whh, *Wbq, *Wsq;  // pack16 layout [k/4][N][4], K padded to 16
  const float *bsa, *bih, *bhh, *bbq, *bsq;
  const float *prev_belief, *prev_state, *actions, *nonterms, *eemb;
  NoiseSrc eps_post;
  float *featx, *post_mean, *post_std, *xsa, *e, *gates, *hq;
  float min_std;
  float* xbuf;      // [group][kind 2][rotation 4][KP*16]
  unsigned* flags;  // [group][NW] on 128-byte lines
  unsigned* err;
  unsigned* status;  // caller's sticky status word (nullable): REPO_SCAN_STATUS_* bits are OR-ed in on an abort
  int spin_limit;
};

__device__ __forceinline__ f32x4v ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kSC1));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, const f32x4v& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, off, 0, kSC1);
}
// The forward scan's all-gathers are data-tagged (no vmcnt(0) wait, no flag hop: 590 -> 539 us); the reverse scan's
// four-tile gather would poll 13 granules per thread and gained nothing that way (663 vs 655 us): it keeps epoch flags.
// a bounded poll ran out: the per-launch word in the workspace (peers of the group read nothing from it; kept for
// debuggers) and the caller's sticky status word, which the host reads back with its per-update scalar copy
__device__ __forceinline__ void raise_status(unsigned* err, unsigned* status, unsigned bit) {
  __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (status) __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ f32x4v kSentinel4() {
  const float n = __builtin_bit_cast(float, 0xffffffffu);
  return f32x4v{n, n, n, n};
}
__device__ __forceinline__ bool has_sentinel(const f32x4v& v) {
  return __builtin_bit_cast(unsigned, v[0]) == 0xffffffffu || __builtin_bit_cast(unsigned, v[1]) == 0xffffffffu ||
         __builtin_bit_cast(unsigned, v[2]) == 0xffffffffu || __builtin_bit_cast(unsigned, v[3]) == 0xffffffffu;
}
__device__ __forceinline__ f32x4v mfma4(const f32x4v& w, const f32x4v& x, f32x4v acc) {
#pragma unroll
  for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u], x[u], acc, 0, 0, 0);
  return acc;
}

// KBX / KBD / KBH: 16-k blocks of X = S + A, of D, of Hd
template <int KBX, int KBD, int KBH>
__global__ __launch_bounds__(256) void observe_cs_fwd_kernel(CsFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int DP = KBD * 16, HP = KBH * 16, XP = KBX * 16;
  constexpr int KP = DP > HP ? DP : HP;
  const int T = p.T, B = p.B, A = p.A, D = p.D, Hd = p.Hd, S = p.S;
  const int X = S + A, F = D + S;
  float* XS = lds;                    // tiles, k4-interleaved: x, e, belief, hq
  float* ES = XS + XP * 16;
  float* HS = ES + DP * 16;
  float* QS = HS + DP * 16;
  float* G4 = QS + HP * 16;           // [4][16][16]: r, z, W_in e + b, W_hn h + b
  float* PART = G4 + 4 * 256;         // [4 waves][16][16]
  float* RAW = PART + 4 * 256;        // [16][64]
  float* ST = RAW + 16 * 64;          // [16][32]
  __shared__ int s_abort;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int w = blockIdx.x, NW = gridDim.x, grp = blockIdx.y;
  const int c0 = 16 * w;
  const int b0 = 16 * grp, nr = min(16, B - b0);
  const int erow = tid >> 4, ecol = tid & 15;  // element role of the pointwise phases
  const int ec = c0 + ecol;

  // ---- zero the activation tiles (their K padding must stay finite), load the stationary weights
  for (int i = tid; i < XP * 16 + 2 * DP * 16 + HP * 16 + 2 * 4 * 256 + 16 * 64 + 16 * 32; i += 256) XS[i] = 0.f;
  if (tid == 0) s_abort = 0;
  // ---- every weight this workgroup multiplies by is REGISTER-stationary: a lane keeps exactly the fragments it feeds
  //      the MFMAs with (column `own` = its column of the own 16, k = 16 kb + 4 lq + 0..3) -- 196 VGPRs; LDS holds the
  //      activation tiles only (57 KB), so the CU stays open to the other lanes' convolution workgroups
  const int own = min(c0 + li, D - 1), ownh = min(c0 + li, Hd - 1);
  constexpr int KH = (KBD + 1) / 2;   // blocks of a half of the K = D reduction
  f32x4v WBf[KBD], WBh[KH], WC[(KBD + 3) / 4];
  {
    // stage B: wave 0: W_ir, 1: W_hr, 2: W_iz, 3: W_hz (whole K); then W_in halves on waves 0 / 1, W_hn halves on 2 / 3
    const f32x4v* full = reinterpret_cast<const f32x4v*>((wave & 1) ? p.Whh : p.Wih);
    const int gfull = wave >> 1;
#pragma unroll
    for (int kb = 0; kb < KBD; ++kb) WBf[kb] = full[(size_t)(kb * 4 + lq) * 3 * D + gfull * D + own];
    const f32x4v* half = reinterpret_cast<const f32x4v*>(wave < 2 ? p.Wih : p.Whh);
    const int kbh = (wave & 1) * KH;
#pragma unroll
    for (int x = 0; x < KH; ++x) WBh[x] = half[(size_t)(min(kbh + x, KBD - 1) * 4 + lq) * 3 * D + 2 * D + own];
    // stage C: this wave's blocks of the K = D reduction of W_bq
    const int kc0 = (wave * KBD) / 4;
#pragma unroll
    for (int x = 0; x < (KBD + 3) / 4; ++x)
      WC[x] = reinterpret_cast<const f32x4v*>(p.Wbq)[(size_t)(min(kc0 + x, KBD - 1) * 4 + lq) * Hd + ownh];
  }
  // register-stationary: W_sa tiles wave, wave+4, ... (stage A), W_sq tile `wave` (stage D)
  constexpr int NTD = KBD;  // column tiles of D
  f32x4v WA[4][KBX], WD[KBH], bA[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tA = wave + 4 * j, col = min(tA * 16 + li, D - 1);
#pragma unroll
    for (int kb = 0; kb < KBX; ++kb)
      WA[j][kb] = reinterpret_cast<const f32x4v*>(p.Wsa)[(size_t)(kb * 4 + lq) * D + col];
#pragma unroll
    for (int r = 0; r < 4; ++r) bA[j][r] = p.bsa[min(tA * 16 + 4 * lq + r, D - 1)];
  }
#pragma unroll
  for (int kb = 0; kb < KBH; ++kb)
    WD[kb] = reinterpret_cast<const f32x4v*>(p.Wsq)[(size_t)(kb * 4 + lq) * 2 * S + min(wave * 16 + li, 2 * S - 1)];
  f32x4v bD;  // stage D bias quad
#pragma unroll
  for (int r = 0; r < 4; ++r) bD[r] = p.bsq[min(wave * 16 + 4 * lq + r, 2 * S - 1)];
  const int ecc = min(ec, D - 1);  // gate biases of the pointwise role's column
  const float bg_r = p.bih[ecc] + p.bhh[ecc], bg_z = p.bih[D + ecc] + p.bhh[D + ecc];
  const float bg_in = p.bih[2 * D + ecc], bg_hn = p.bhh[2 * D + ecc];
  const float b_q = p.bbq[min(ec, Hd - 1)];
  __syncthreads();

  // ---- slot 0: carried belief / state into the tiles; featx[0]
  for (int i = tid; i < 16 * D; i += 256) {
    const int row = i / D, c = i % D;
    const float v = row < nr ? p.prev_belief[(size_t)(b0 + row) * D + c] : 0.f;
    HS[ai(c, row)] = v;
    if (row < nr && (c >> 4) == w) p.featx[(size_t)(b0 + row) * F + c] = v;
  }
  for (int i = tid; i < 16 * S; i += 256) {
    const int row = i / S, s = i % S;
    const float v = row < nr ? p.prev_state[(size_t)(b0 + row) * S + s] : 0.f;
    ST[row * 32 + s] = v;
    if (row < nr && w == 0) p.featx[(size_t)(b0 + row) * F + D + s] = v;
  }

  const __amdgpu_buffer_rsrc_t rx = wrsrc(p.xbuf + (size_t)grp * 8 * KP * 16, 4u * 8u * KP * 16);
  unsigned* flags = p.flags + (size_t)grp * NW * 32;
  // x-vector roles of this thread: elements tid, tid + 256, ... of the 16 x X tile
  constexpr int XPER = (16 * XP + 255) / 256;
  float xin[XPER];
  auto load_x = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XPER; ++j) {
      const int i = tid + 256 * j, row = i / X, k = i % X;
      const size_t gr = (size_t)t * B + b0 + row;
      xin[j] = (i < 16 * X && row < nr) ? (k < S ? p.nonterms[gr] : p.actions[gr * A + (k - S)]) : 0.f;
    }
  };
  float em_next = 0.f;
  auto load_em = [&](int t) __attribute__((always_inline)) {
    em_next = (erow < nr && ec < Hd) ? p.eemb[((size_t)t * B + b0 + erow) * Hd + ec] : 0.f;
  };
  if (T > 0) {
    load_x(0);
    load_em(0);
  }
  __syncthreads();

  // all-gather of one own slice (already in `tile`, k-groups c0/4 .. c0/4+3): kind 0 = belief, 1 = hq
  auto exchange = [&](float* tile, int kind, int t, int kp) __attribute__((always_inline)) -> bool {
    // data-tagged hand-off: four rotating buffers per kind, all cells start as the sentinel (a NaN no activation can be);
    // the producer stores its slice into buffer t % 4 and re-arms its slice of buffer (t + 2) % 4 (every consumer is
    // done with it: a workgroup publishes exchange t only after gathering t - 1, which needed every peer's publish of
    // t - 1, which each peer issued after gathering t - 2 -- and the re-arm is acknowledged before this workgroup's
    // NEXT publish, its own gather in between waits for vmcnt(0)); a consumer re-loads a granule until it holds no
    // sentinel.  No vmcnt(0) wait, no flag hop: 2.6 -> ~1.5 us per exchange.
    const unsigned base = 4u * (unsigned)((kind * 4 + (t & 3)) * KP * 16);
    const unsigned rearm = 4u * (unsigned)((kind * 4 + ((t + 2) & 3)) * KP * 16);
    if (wave == 0) {
      const unsigned o = 16u * (unsigned)((c0 >> 2) * 16 + lane);
      st_sc1(rx, base + o, *reinterpret_cast<const f32x4v*>(tile + ((c0 >> 2) * 16 + lane) * 4));
      st_sc1(rx, rearm + o, kSentinel4());
    }
    constexpr int NV = (KP * 4 + 255) / 256;
    f32x4v g[NV];
    bool okv[NV];
    bool all = false;
#pragma unroll
    for (int j = 0; j < NV; ++j) okv[j] = (tid + 256 * j) >= kp * 4;
    for (int n = 0; !all; ++n) {
      all = true;
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (!okv[j]) g[j] = ld_sc1(rx, base + 16u * (unsigned)(tid + 256 * j));
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (!okv[j]) {
          okv[j] = !has_sentinel(g[j]);
          all = all && okv[j];
        }
      if (!all && n >= p.spin_limit) {
        s_abort = 1;
        raise_status(p.err, p.status, REPO_SCAN_STATUS_FWD_TIMEOUT);
        break;
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + 256 * j;
      if (i < kp * 4) reinterpret_cast<f32x4v*>(tile)[i] = g[j];
    }
    __syncthreads();
    return !s_abort;
  };

  // a group that gave up on its peers (error word raised) must not leave plausible numbers behind
  auto poison = [&]() __attribute__((always_inline)) {
    if (erow < nr && ec < D) p.featx[((size_t)T * B + b0 + erow) * F + ec] = __builtin_bit_cast(float, 0x7fc00000u);
  };
  for (int t = 0; t < T; ++t) {
    const size_t row0 = (size_t)t * B + b0;
    const float em = em_next;
    // ---- x = [state * nonterm, action]
#pragma unroll
    for (int j = 0; j < XPER; ++j) {
      const int i = tid + 256 * j, row = i / X, k = i % X;
      if (i < 16 * X) {
        const float v = row < nr ? (k < S ? ST[row * 32 + k] * xin[j] : xin[j]) : 0.f;
        XS[ai(k, row)] = v;
        if (w == 0 && row < nr) p.xsa[(row0 + row) * X + k] = v;
      }
    }
    if (t + 1 < T) {
      load_x(t + 1);
      load_em(t + 1);
    }
    __syncthreads();
    // ---- A: e = elu(W_sa x + b), every column tile (replicated in all workgroups); the own tile is saved
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tA = wave + 4 * j;
      if (tA < NTD) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KBX; ++kb)
          acc = mfma4(WA[j][kb], *reinterpret_cast<const f32x4v*>(XS + ((kb * 4 + lq) * 16 + li) * 4), acc);
        f32x4v v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = elu(acc[r] + bA[j][r]);
        const int n0 = tA * 16 + 4 * lq;
        stq(ES, n0, li, v);
        if (tA == w && li < nr) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < D) p.e[(row0 + li) * D + n0 + r] = v[r];
        }
      }
    }
    __syncthreads();
    // ---- B: GRU pre-activations of the own columns, 6 products x KBD blocks spread evenly over the 4 waves:
    //         wave 0: W_ir e, 1: W_hr h, 2: W_iz e, 3: W_hz h, then W_in e split over waves 0 / 1 and W_hn h over 2 / 3
    //         (the sums meet in the pointwise phase: 78-80 MFMAs per wave instead of 104 on two of them)
    {
      f32x4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const float* xfull = (wave & 1) == 0 ? ES : HS;
#pragma unroll
      for (int kb = 0; kb < KBD; ++kb)
        a0 = mfma4(WBf[kb], *reinterpret_cast<const f32x4v*>(xfull + ((kb * 4 + lq) * 16 + li) * 4), a0);
      const float* xhalf = wave < 2 ? ES : HS;
      const int kb0 = (wave & 1) * KH;
#pragma unroll
      for (int x = 0; x < KH; ++x)
        if (kb0 + x < KBD)
          a1 = mfma4(WBh[x], *reinterpret_cast<const f32x4v*>(xhalf + (((kb0 + x) * 4 + lq) * 16 + li) * 4), a1);
      // G4[0..3] = W_ir e, W_hr h, W_iz e, W_hz h;  PART[0..3] = the halves of W_in e (0, 1) and of W_hn h (2, 3)
      *reinterpret_cast<f32x4v*>(G4 + wave * 256 + li * 16 + 4 * lq) = a0;
      *reinterpret_cast<f32x4v*>(PART + wave * 256 + li * 16 + 4 * lq) = a1;
    }
    __syncthreads();
    {
      const int o = erow * 16 + ecol;
      const float rg = sigmoidf(G4[o] + G4[256 + o] + bg_r), zg = sigmoidf(G4[512 + o] + G4[768 + o] + bg_z);
      const float gin = PART[o] + PART[256 + o] + bg_in, ghn = PART[512 + o] + PART[768 + o] + bg_hn;
      const float ng = tanh_fast(gin + rg * ghn);
      const float hprev = HS[ai(ec, erow)];
      const float hn = (1.f - zg) * ng + zg * hprev;
      HS[ai(ec, erow)] = hn;  // own slice only: every product over the old belief is done (barrier above)
      if (erow < nr && ec < D) {
        float* g = p.gates + (row0 + erow) * 4 * D;
        g[ec] = rg;
        g[D + ec] = zg;
        g[2 * D + ec] = ng;
        g[3 * D + ec] = ghn;
        p.featx[((size_t)(t + 1) * B + b0 + erow) * F + ec] = hn;
      }
    }
    __syncthreads();
    if (!exchange(HS, 0, t, DP)) return poison();
    // ---- C: hq = elu(W_bq[:, :D] h + eemb + b), own columns, K split over the waves
    {
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
      const int kb0 = (wave * KBD) / 4, kb1 = ((wave + 1) * KBD) / 4;
#pragma unroll
      for (int x = 0; x < (KBD + 3) / 4; ++x)
        if (kb0 + x < kb1)
          acc = mfma4(WC[x], *reinterpret_cast<const f32x4v*>(HS + (((kb0 + x) * 4 + lq) * 16 + li) * 4), acc);
      *reinterpret_cast<f32x4v*>(PART + wave * 256 + li * 16 + 4 * lq) = acc;
    }
    __syncthreads();
    {
      const int o = erow * 16 + ecol;
      const float v = elu(PART[o] + PART[256 + o] + PART[512 + o] + PART[768 + o] + em + b_q);
      QS[ai(ec, erow)] = v;
      if (erow < nr && ec < Hd) p.hq[(row0 + erow) * Hd + ec] = v;
    }
    __syncthreads();
    if (!exchange(QS, 1, t, HP)) return poison();
    // ---- D: posterior (mean | raw std) = W_sq hq + b, replicated; softplus + sample
    {
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KBH; ++kb)
        acc = mfma4(WD[kb], *reinterpret_cast<const f32x4v*>(QS + ((kb * 4 + lq) * 16 + li) * 4), acc);
      *reinterpret_cast<f32x4v*>(RAW + li * 64 + wave * 16 + 4 * lq) = acc + bD;
    }
    __syncthreads();
    for (int i = tid; i < 16 * S; i += 256) {
      const int row = i / S, s = i % S;
      const float mean = RAW[row * 64 + s];
      const float sd = softplus(RAW[row * 64 + S + s]) + p.min_std;
      float smp = 0.f;
      if (row < nr) {
        const size_t o = (row0 + row) * S + s;
        smp = fmaf(sd, p.eps_post.at(o), mean);
        if (w == 0) {
          p.post_mean[o] = mean;
          p.post_std[o] = sd;
          p.featx[((size_t)(t + 1) * B + b0 + row) * F + D + s] = smp;
        }
      }
      ST[row * 32 + s] = smp;
    }
    __syncthreads();
  }
}

static size_t cs_pack_floats(int64_t A, int64_t D, int64_t Hd, int64_t S) {
  return pack_floats(D, S + A) + 2 * pack_floats(3 * D, D) + pack_floats(Hd, D) + pack_floats(2 * S, Hd);
}

bool scan_cs_ok(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S) {
  (void)T;
  // one instantiation: the reference shapes (belief = hidden = 200 -> 13 blocks, S + A <= 48, 2 S <= 64)
  return B > 0 && pad16((int)D) == 208 && pad16((int)Hd) == 208 && pad16((int)(S + A)) == 48 && 2 * S <= 64 && S <= 32 &&
         D % 4 == 0;
}

size_t scan_cs_fwd_ws_floats(int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S) {
  const int64_t G = (B + 15) / 16, NW = (std::max(D, Hd) + 15) / 16, KP = pad16((int)std::max(D, Hd));
  return cs_pack_floats(A, D, Hd, S) + (size_t)(G * 8 * KP * 16) + (size_t)(G * NW * 32) + 32;
}

int scan_cs_fwd(const ScanCsFwd& q, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!scan_cs_ok(q.T, q.B, q.A, q.D, q.Hd, q.S)) return REPO_E_SHAPE;
  if (!ws || ws_bytes < scan_cs_fwd_ws_floats(q.B, q.A, q.D, q.Hd, q.S) * sizeof(float)) return REPO_E_WS_TOO_SMALL;
  const int d = (int)q.D, h = (int)q.Hd, X = (int)(q.S + q.A), s2 = (int)(2 * q.S);
  const float* const* P = q.params;
  float* w = (float*)ws;
  float* Wsa = w;  w += pack_floats(d, X);
  float* Wih = w;  w += pack_floats(3 * d, d);
  float* Whh = w;  w += pack_floats(3 * d, d);
  float* Wbq = w;  w += pack_floats(h, d);
  float* Wsq = w;  w += pack_floats(s2, h);
  const int64_t G = (q.B + 15) / 16, NW = (std::max(q.D, q.Hd) + 15) / 16, KP = pad16((int)std::max(q.D, q.Hd));
  float* xbuf = w;  w += G * 8 * KP * 16;
  unsigned* flags = (unsigned*)w;  w += G * NW * 32;
  unsigned* err = (unsigned*)w;
  PackArgs pa;
  pa.njobs = 0;
  pa.job[pa.njobs++] = PackJob{P[0], Wsa, d, X, X, 1};
  pa.job[pa.njobs++] = PackJob{P[2], Wih, 3 * d, d, d, 1};
  pa.job[pa.njobs++] = PackJob{P[3], Whh, 3 * d, d, d, 1};
  pa.job[pa.njobs++] = PackJob{P[10], Wbq, h, d, (int)(q.D + q.E), 1};
  pa.job[pa.njobs++] = PackJob{P[12], Wsq, s2, h, h, 1};
  // the exchange state is armed by two FILL jobs of the same launch (flags + error word = 0, every cell of the exchange
  // buffer = the sentinel), NOT by hipMemsetAsync: inside a captured HIP graph (the acting path) the memset nodes did not
  // reliably take effect before the scan kernel that follows them -- replays gathered the PREVIOUS replay's cells as if
  // they were this step's (round 6: tools/act_graph_debug.py, tools/probe/graph_memset.hip)
  hipError_t he = hipSuccess;
#ifdef CS_MEMSET_NODES   // (the round 3-5 form, for A/B)
  int rc = launch_pack(pa, s);
  if (rc) return rc;
  he = hipMemsetAsync(flags, 0, (size_t)(G * NW * 32 + 32) * sizeof(unsigned), s);
  if (he != hipSuccess) return (int)he;
  he = hipMemsetAsync(xbuf, 0xff, (size_t)(G * 8 * KP * 16) * sizeof(float), s);  // every cell = the sentinel
  if (he != hipSuccess) return (int)he;
#else
  pa.job[pa.njobs++] = fill_job(flags, (size_t)(G * NW * 32 + 32), 0u);
  pa.job[pa.njobs++] = fill_job(xbuf, (size_t)(G * 8 * KP * 16), 0xFFFFFFFFu);
  int rc = launch_pack(pa, s);
  if (rc) return rc;
#endif
  if (q.T == 0) return REPO_OK;
  CsFwdArgs a;
  a.T = (int)q.T; a.B = (int)q.B; a.A = (int)q.A; a.D = d; a.Hd = h; a.S = (int)q.S;
  a.Wsa = Wsa; a.Wih = Wih; a.Whh = Whh; a.Wbq = Wbq; a.Wsq = Wsq;
  a.bsa = P[1]; a.bih = P[4]; a.bhh = P[5]; a.bbq = P[11]; a.bsq = P[13];
  a.prev_belief = q.prev_belief; a.prev_state = q.prev_state; a.actions = q.actions; a.nonterms = q.nonterms;
  a.eemb = q.eemb; a.eps_post = q.eps_post;
  a.featx = q.featx; a.post_mean = q.post_mean; a.post_std = q.post_std; a.xsa = q.xsa; a.e = q.e; a.gates = q.gates;
  a.hq = q.hq; a.min_std = q.min_std;
  a.xbuf = xbuf; a.flags = flags; a.err = err; a.status = q.status;
  a.spin_limit = scan_cs_spin_limit();
  constexpr int DP = 208, HP = 208, XP = 48;
  const size_t lds_b = (size_t)(XP * 16 + 2 * DP * 16 + HP * 16 + 2 * 4 * 256 + 16 * 64 + 16 * 32) * sizeof(float);
  he = hipFuncSetAttribute((const void*)observe_cs_fwd_kernel<3, 13, 13>, hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)lds_b);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL((observe_cs_fwd_kernel<3, 13, 13>), dim3((unsigned)NW, (unsigned)G), dim3(256), lds_b, s, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

// ======================================================================================= reverse scan
// The same column-split layout, walked backwards.  Workgroup w owns belief / hidden columns [16w, 16w+16):
//   a  posterior output deltas (mean, raw std) from d state (upstream + carried), KL gradients    pointwise, replicated
//   1  dhq = (W_sq^T d out) * elu'(hq)          all Hd columns, replicated (W_sq^T in registers)   13 tiles x 4 blocks
//   2  d belief = carried + upstream + prior share + W_bq^T dhq   own columns, K split over waves  13 blocks
//   d  GRU pointwise: d gates of the own columns; carried d belief = d belief * z
//   X1 all-gather of (g_r, g_z, g_n, g_hn)      four tiles
//   3  d belief_{t-1} += W_hh^T d gh ; d e = W_ih^T d gi          own columns, K = 3D, K split     78 blocks
//   X2 all-gather of d e_pre = d e * elu'(e)
//   4  d state_{t-1} = (W_sa[:, :S]^T d e_pre) * nonterm          replicated (W_sa^T in registers) 2 tiles x 13 blocks
// Every weight fragment is register-stationary (a lane keeps what it feeds the MFMAs with); LDS: the activation tiles.
struct CsBwdArgs {
  int T, B, A, D, Hd, S;
  const float *WsqT, *WbqT, *WhhT, *WihT, *WsaT;  // pack16: [k/4][N][4]; WhhT / WihT: 3 packs (one per gate) back to back
  const float *nonterms, *featx, *post_std, *e, *gates, *hq;
  NoiseSrc eps_post;
  const float *dfeat, *dqm, *dqs, *dbx;
  float *doutq, *dhq, *dgi, *dgh, *de, *dprev_belief, *dprev_state;
  float min_std;
  float* xbuf;      // [group][20 tiles of KP*16]: [rotation 4][g_r, g_z, g_n, g_hn], then [rotation 4] d e_pre
  unsigned* flags;
  unsigned* err;
  unsigned* status;  // caller's sticky status word (nullable): REPO_SCAN_STATUS_* bits are OR-ed in on an abort
  int spin_limit;
};

template <int KBD, int KBH>
__global__ __launch_bounds__(256) void observe_cs_bwd_kernel(CsBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int DP = KBD * 16, HP = KBH * 16, KP = DP > HP ? DP : HP, TS = KP * 16;  // TS: floats of a tile
  constexpr int KBO = 4;  // 16-k blocks of 2S (<= 64)
  const int T = p.T, B = p.B, D = p.D, Hd = p.Hd, S = p.S;
  const int F = D + S;
  float* G = lds;                // [4][TS]: g_r, g_z, g_n, g_hn tiles; G[3] doubles as the dhq tile, G[0] as d e_pre
  float* DO = G + 4 * TS;        // [64 x 16] posterior output deltas
  float* PART = DO + 64 * 16;    // [2][4][256]
  float* DST = PART + 2 * 4 * 256;  // [16][32] carried d state
  __shared__ int s_abort;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int w = blockIdx.x, NW = gridDim.x, grp = blockIdx.y;
  const int c0 = 16 * w;
  const int b0 = 16 * grp, nr = min(16, B - b0);
  const int erow = tid >> 4, ecol = tid & 15, ec = c0 + ecol;
  const bool e_ok = erow < nr && ec < D, h_ok = erow < nr && ec < Hd;

  for (int i = tid; i < 4 * TS + 64 * 16 + 2 * 4 * 256 + 16 * 32; i += 256) G[i] = 0.f;
  if (tid == 0) s_abort = 0;
  // stage 3's weights (the own-column slices of W_hh^T and W_ih^T, K = 3 D as (gate, block) pairs split over the
  // waves) are register-stationary like everything else: LDS holds the activation tiles only (67 KB)
  constexpr int NB3 = 3 * KBD, MX3 = (NB3 + 3) / 4;
  const int x3_0 = (wave * NB3) / 4, x3_1 = ((wave + 1) * NB3) / 4;
  f32x4v WH3[MX3], WI3[MX3];
  {
    const int own = min(c0 + li, D - 1);
#pragma unroll
    for (int i = 0; i < MX3; ++i) {
      const int x = min(x3_0 + i, NB3 - 1), g = x / KBD, kb = x % KBD;
      WH3[i] = reinterpret_cast<const f32x4v*>(p.WhhT + (size_t)g * DP * D)[(size_t)(kb * 4 + lq) * D + own];
      WI3[i] = reinterpret_cast<const f32x4v*>(p.WihT + (size_t)g * DP * D)[(size_t)(kb * 4 + lq) * D + own];
    }
  }
  // register-stationary weights
  constexpr int NTH = KBH;
  f32x4v W1[4][KBO];  // stage 1: W_sq^T, hidden-column tiles wave, wave + 4, ...
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kb = 0; kb < KBO; ++kb)
      W1[j][kb] = reinterpret_cast<const f32x4v*>(p.WsqT)[(size_t)(kb * 4 + lq) * Hd + min((wave + 4 * j) * 16 + li, Hd - 1)];
  constexpr int KB2 = (KBH + 3) / 4;  // stage 2: this wave's blocks of the K = Hd reduction
  const int kb2_0 = (wave * KBH) / 4, kb2_1 = ((wave + 1) * KBH) / 4;
  f32x4v W2[KB2];
#pragma unroll
  for (int x = 0; x < KB2; ++x)
    W2[x] = reinterpret_cast<const f32x4v*>(p.WbqT)[(size_t)((min(kb2_0 + x, KBH - 1)) * 4 + lq) * D + min(c0 + li, D - 1)];
  // stage 4: d state tile (wave & 1), K half (wave >> 1)
  constexpr int KB4 = (KBD + 1) / 2;
  const int kb4_0 = (wave >> 1) * KB4;
  f32x4v W4[KB4];
#pragma unroll
  for (int x = 0; x < KB4; ++x)
    W4[x] = reinterpret_cast<const f32x4v*>(p.WsaT)[(size_t)(min(kb4_0 + x, KBD - 1) * 4 + lq) * S + min((wave & 1) * 16 + li, S - 1)];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rx = wrsrc(p.xbuf + (size_t)grp * 20 * TS, 4u * 20u * TS);
  unsigned* flags = p.flags + (size_t)grp * NW * 32;
  const __amdgpu_buffer_rsrc_t rhq = arsrc(p.hq);

  // ---- per-step operands, fetched one step ahead
  constexpr int SPER = 2;  // (row, s) roles of the pointwise output stage: 16 * S <= 512
  struct StepIn {
    float dsm[SPER], dm[SPER], dsd[SPER], sd[SPER], nt[SPER];
    f32x4v hq4[4];
    float dfb, g_r, g_z, g_n, g_hn, hprev, ev;
  } in;
  auto load_step = [&](int t) __attribute__((always_inline)) {
    const size_t row0 = (size_t)t * B + b0;
#pragma unroll
    for (int j = 0; j < SPER; ++j) {
      const int i = tid + 256 * j, row = i / S, s = i % S;
      const bool ok = i < 16 * S && row < nr;
      const size_t o = (row0 + row) * S + s;
      in.dsm[j] = (ok && p.dfeat) ? p.dfeat[(row0 + row) * F + D + s] : 0.f;
      in.dm[j] = (ok && p.dqm) ? p.dqm[o] : 0.f;
      in.dsd[j] = (ok && p.dqs) ? p.dqs[o] : 0.f;
      in.sd[j] = ok ? p.post_std[o] : 1.f;
      in.nt[j] = ok ? p.nonterms[row0 + row] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n0 = (wave + 4 * j) * 16 + 4 * lq;
      const bool ok = (wave + 4 * j) < NTH && li < nr && n0 + 3 < Hd;
      in.hq4[j] = ok ? bldq(rhq, 4u * (unsigned)(li * Hd + n0), 4u * (unsigned)(row0 * Hd)) : f32x4v{0.f, 0.f, 0.f, 0.f};
      if (!ok && (wave + 4 * j) < NTH && li < nr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) in.hq4[j][r] = n0 + r < Hd ? p.hq[(row0 + li) * Hd + n0 + r] : 0.f;
      }
    }
    const float* g = p.gates + (row0 + erow) * 4 * D;
    in.dfb = e_ok ? ((p.dfeat ? p.dfeat[(row0 + erow) * F + ec] : 0.f) + (p.dbx ? p.dbx[(row0 + erow) * D + ec] : 0.f)) : 0.f;
    in.g_r = e_ok ? g[ec] : 0.f;
    in.g_z = e_ok ? g[D + ec] : 0.f;
    in.g_n = e_ok ? g[2 * D + ec] : 0.f;
    in.g_hn = e_ok ? g[3 * D + ec] : 0.f;
    in.hprev = e_ok ? p.featx[((size_t)t * B + b0 + erow) * F + ec] : 0.f;
    in.ev = e_ok ? p.e[(row0 + erow) * D + ec] : 0.f;
  };
  if (T > 0) load_step(T - 1);
  float dh = 0.f;  // carried d belief of element (erow, ec)

  // all-gather of `nt` own slices starting at tile index `t0` of this step's parity block
  // kind 0: the four gate tiles (xbuf tiles [rot 4][4]); kind 1: d e_pre ([rot 4][1] behind them).  `seq` = step.
  auto exchange = [&](int kind, float* tiles, int seq) __attribute__((always_inline)) -> bool {
    const int nt = kind == 0 ? 4 : 1;
    const unsigned epoch = (unsigned)(2 * seq + kind + 1);
    const unsigned base = 4u * (unsigned)((kind == 0 ? (seq & 1) * 4 : 16 + (seq & 1)) * TS);
    if (wave == 0) {
      for (int x = 0; x < nt; ++x) {
        const f32x4v v = *reinterpret_cast<const f32x4v*>(tiles + x * TS + ((c0 >> 2) * 16 + lane) * 4);
        st_sc1(rx, base + 4u * (unsigned)(x * TS) + 16u * (unsigned)((c0 >> 2) * 16 + lane), v);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(flags + 32 * w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane < NW) {
        int n = 0;
        while (__hip_atomic_load(flags + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++n > p.spin_limit) {
            s_abort = 1;
            raise_status(p.err, p.status, REPO_SCAN_STATUS_BWD_TIMEOUT);
            break;
          }
        }
      }
    }
    __syncthreads();
    if (s_abort) return false;
    const int nv = nt * TS / 4;
    for (int i0 = 0; i0 < nv; i0 += 256 * 8) {
      f32x4v g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + tid + 256 * j;
        g[j] = ld_sc1(rx, i < nv ? base + 16u * (unsigned)i : 0xfffffff0u);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + tid + 256 * j;
        if (i < nv) reinterpret_cast<f32x4v*>(tiles)[i] = g[j];
      }
    }
    __syncthreads();
    return true;
  };

  // a group that gave up on its peers (error word raised) must not leave plausible numbers behind
  auto poison = [&]() __attribute__((always_inline)) {
    if (e_ok) p.de[(size_t)(b0 + erow) * D + ec] = __builtin_bit_cast(float, 0x7fc00000u);
  };
  for (int t = T - 1; t >= 0; --t) {
    const size_t row0 = (size_t)t * B + b0;
    const int step = T - 1 - t;
    // ---- a: posterior output deltas
#pragma unroll
    for (int j = 0; j < SPER; ++j) {
      const int i = tid + 256 * j, row = i / S, s = i % S;
      if (i < 16 * S) {
        float dm = 0.f, draw = 0.f;
        if (row < nr) {
          const size_t o = (row0 + row) * S + s;
          const float dsmp = in.dsm[j] + DST[row * 32 + s];
          dm = in.dm[j] + dsmp;
          const float dsd = fmaf(dsmp, p.eps_post.at(o), in.dsd[j]);
          draw = dsd * (-expm1f(-(in.sd[j] - p.min_std)));
          if (w == 0) {
            p.doutq[(row0 + row) * 2 * S + s] = dm;
            p.doutq[(row0 + row) * 2 * S + S + s] = draw;
          }
        }
        DO[ai(s, row)] = dm;
        DO[ai(S + s, row)] = draw;
      }
    }
    const float nt0 = in.nt[0], nt1 = in.nt[1];
    __syncthreads();
    // ---- 1: dhq = (W_sq^T d out) * elu'(hq), every hidden tile (replicated); the own tile is saved.  Into G[3].
    float* QD = G + 3 * TS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tA = wave + 4 * j;
      if (tA < NTH) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KBO; ++kb)
          acc = mfma4(W1[j][kb], *reinterpret_cast<const f32x4v*>(DO + ((kb * 4 + lq) * 16 + li) * 4), acc);
        f32x4v v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = li < nr ? acc[r] * elu_grad_from_out(in.hq4[j][r]) : 0.f;
        const int n0 = tA * 16 + 4 * lq;
        stq(QD, n0, li, v);
        if (tA == w && li < nr) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < Hd) p.dhq[(row0 + li) * Hd + n0 + r] = v[r];
        }
      }
    }
    __syncthreads();
    // ---- 2: d belief (own columns) = carried + upstream + W_bq^T dhq, K split over the waves
    {
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int x = 0; x < KB2; ++x)
        if (kb2_0 + x < kb2_1)
          acc = mfma4(W2[x], *reinterpret_cast<const f32x4v*>(QD + (((kb2_0 + x) * 4 + lq) * 16 + li) * 4), acc);
      *reinterpret_cast<f32x4v*>(PART + wave * 256 + li * 16 + 4 * lq) = acc;
    }
    __syncthreads();
    // ---- d: GRU pointwise on element (erow, ec)
    {
      const int o = erow * 16 + ecol;
      const float db_ = dh + in.dfb + PART[o] + PART[256 + o] + PART[512 + o] + PART[768 + o];
      float g_r = 0.f, g_z = 0.f, g_n = 0.f, g_hn = 0.f, dhprev = 0.f;
      if (e_ok) {
        const float rg = in.g_r, zg = in.g_z, ng = in.g_n, ghn = in.g_hn;
        const float dn = db_ * (1.f - zg), dz = db_ * (in.hprev - ng);
        dhprev = db_ * zg;
        g_n = dn * (1.f - ng * ng);
        g_hn = g_n * rg;
        g_r = g_n * ghn * rg * (1.f - rg);
        g_z = dz * zg * (1.f - zg);
        float* gi = p.dgi + (row0 + erow) * 3 * D;
        float* gh = p.dgh + (row0 + erow) * 3 * D;
        gi[ec] = g_r;
        gi[D + ec] = g_z;
        gi[2 * D + ec] = g_n;
        gh[ec] = g_r;
        gh[D + ec] = g_z;
        gh[2 * D + ec] = g_hn;
      }
      dh = dhprev;
      const int a_ = ai(ec, erow);
      // (G[3] held dhq: every read of it is behind the barrier above)
      G[a_] = g_r;
      G[TS + a_] = g_z;
      G[2 * TS + a_] = g_n;
      G[3 * TS + a_] = g_hn;
    }
    __syncthreads();
    if (!exchange(0, G, step)) return poison();
    // ---- 3: d belief_{t-1} += W_hh^T (g_r, g_z, g_hn);  d e = W_ih^T (g_r, g_z, g_n): own columns, K = 3 D split
    //         over the waves as (gate, block) pairs
    {
      f32x4v ah = {0.f, 0.f, 0.f, 0.f}, ae = ah;
#pragma unroll
      for (int i = 0; i < MX3; ++i) {
        const int x = x3_0 + i;
        if (x < x3_1) {
          const int g = x / KBD, kb = x % KBD;
          const int o = ((kb * 4 + lq) * 16 + li) * 4;
          const f32x4v gi4 = *reinterpret_cast<const f32x4v*>(G + g * TS + o);
          const f32x4v gh4 = g < 2 ? gi4 : *reinterpret_cast<const f32x4v*>(G + 3 * TS + o);
          ah = mfma4(WH3[i], gh4, ah);
          ae = mfma4(WI3[i], gi4, ae);
        }
      }
      *reinterpret_cast<f32x4v*>(PART + wave * 256 + li * 16 + 4 * lq) = ah;
      *reinterpret_cast<f32x4v*>(PART + 1024 + wave * 256 + li * 16 + 4 * lq) = ae;
    }
    __syncthreads();
    {
      const int o = erow * 16 + ecol;
      dh += PART[o] + PART[256 + o] + PART[512 + o] + PART[768 + o];
      const float ae = PART[1024 + o] + PART[1280 + o] + PART[1536 + o] + PART[1792 + o];
      const float v = e_ok ? ae * elu_grad_from_out(in.ev) : 0.f;
      if (e_ok) p.de[(row0 + erow) * D + ec] = v;
      G[ai(ec, erow)] = v;  // G[0] now carries d e_pre (the gate tiles are consumed: barrier above)
    }
    if (t > 0) load_step(t - 1);
    __syncthreads();
    if (!exchange(1, G, step)) return poison();
    // ---- 4: d state_{t-1} = (W_sa[:, :S]^T d e_pre) * nonterm, replicated
    {
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int x = 0; x < KB4; ++x)
        if (kb4_0 + x < KBD)
          acc = mfma4(W4[x], *reinterpret_cast<const f32x4v*>(G + (((kb4_0 + x) * 4 + lq) * 16 + li) * 4), acc);
      *reinterpret_cast<f32x4v*>(PART + wave * 256 + li * 16 + 4 * lq) = acc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SPER; ++j) {
      const int i = tid + 256 * j, row = i / S, s = i % S;
      if (i < 16 * S) {
        const int tl = s >> 4, o = row * 16 + (s & 15);
        const float v = PART[tl * 256 + o] + PART[(tl + 2) * 256 + o];
        DST[row * 32 + s] = row < nr ? v * (j == 0 ? nt0 : nt1) : 0.f;
      }
    }
    __syncthreads();
  }
  if (p.dprev_belief && e_ok) p.dprev_belief[(size_t)(b0 + erow) * D + ec] = dh;
  if (p.dprev_state && w == 0)
    for (int i = tid; i < nr * S; i += 256) p.dprev_state[(size_t)(b0 + i / S) * S + i % S] = DST[(i / S) * 32 + i % S];
}

static size_t cs_bwd_pack_floats(int64_t D, int64_t Hd, int64_t S) {
  return pack_floats(Hd, 2 * S) + pack_floats(D, Hd) + 6 * pack_floats(D, D) + pack_floats(S, D);
}

size_t scan_cs_bwd_ws_floats(int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S) {
  (void)A;
  const int64_t G = (B + 15) / 16, NW = (std::max(D, Hd) + 15) / 16, KP = pad16((int)std::max(D, Hd));
  return cs_bwd_pack_floats(D, Hd, S) + (size_t)(G * 20 * KP * 16) + (size_t)(G * NW * 32) + 32;
}

int scan_cs_bwd(const ScanCsBwd& q, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!scan_cs_ok(q.T, q.B, q.A, q.D, q.Hd, q.S) || q.T <= 0) return REPO_E_SHAPE;
  if (!ws || ws_bytes < scan_cs_bwd_ws_floats(q.B, q.A, q.D, q.Hd, q.S) * sizeof(float)) return REPO_E_WS_TOO_SMALL;
  const int d = (int)q.D, h = (int)q.Hd, X = (int)(q.S + q.A), s2 = (int)(2 * q.S);
  const float* const* P = q.params;
  float* w = (float*)ws;
  float* WsqT = w;  w += pack_floats(h, s2);
  float* WbqT = w;  w += pack_floats(d, h);
  float* WhhT = w;  w += 3 * pack_floats(d, d);
  float* WihT = w;  w += 3 * pack_floats(d, d);
  float* WsaT = w;  w += pack_floats(q.S, d);
  const int64_t G = (q.B + 15) / 16, NW = (std::max(q.D, q.Hd) + 15) / 16, KP = pad16((int)std::max(q.D, q.Hd));
  float* xbuf = w;  w += G * 20 * KP * 16;
  unsigned* flags = (unsigned*)w;  w += G * NW * 32;
  unsigned* err = (unsigned*)w;
  // transposed products: W(n = input feature, k = output feature) = native[k * ld + n]
  PackArgs pa;
  pa.njobs = 0;
  pa.job[pa.njobs++] = PackJob{P[12], WsqT, h, s2, 1, h};
  pa.job[pa.njobs++] = PackJob{P[10], WbqT, d, h, 1, (int)(q.D + q.E)};
  for (int g = 0; g < 3; ++g) {
    pa.job[pa.njobs++] = PackJob{P[3] + (size_t)g * d * d, WhhT + g * pack_floats(d, d), d, d, 1, d};
    pa.job[pa.njobs++] = PackJob{P[2] + (size_t)g * d * d, WihT + g * pack_floats(d, d), d, d, 1, d};
  }
  pa.job[pa.njobs++] = PackJob{P[0], WsaT, (int)q.S, d, 1, X};
  hipError_t he = hipSuccess;
#ifdef CS_MEMSET_NODES
  int rc = launch_pack(pa, s);
  if (rc) return rc;
  he = hipMemsetAsync(flags, 0, (size_t)(G * NW * 32 + 32) * sizeof(unsigned), s);
  if (he != hipSuccess) return (int)he;
  he = hipMemsetAsync(xbuf, 0xff, (size_t)(G * 20 * KP * 16) * sizeof(float), s);  // every cell = the sentinel
  if (he != hipSuccess) return (int)he;
#else   // armed inside the pack launch (see scan_cs_fwd)
  pa.job[pa.njobs++] = fill_job(flags, (size_t)(G * NW * 32 + 32), 0u);
  pa.job[pa.njobs++] = fill_job(xbuf, (size_t)(G * 20 * KP * 16), 0xFFFFFFFFu);
  int rc = launch_pack(pa, s);
  if (rc) return rc;
#endif
  CsBwdArgs a;
  a.T = (int)q.T; a.B = (int)q.B; a.A = (int)q.A; a.D = d; a.Hd = h; a.S = (int)q.S;
  a.WsqT = WsqT; a.WbqT = WbqT; a.WhhT = WhhT; a.WihT = WihT; a.WsaT = WsaT;
  a.nonterms = q.nonterms; a.featx = q.featx; a.post_std = q.post_std; a.e = q.e; a.gates = q.gates; a.hq = q.hq;
  a.eps_post = q.eps_post;
  a.dfeat = q.dfeat; a.dqm = q.dqm; a.dqs = q.dqs; a.dbx = q.dbx;
  a.doutq = q.doutq; a.dhq = q.dhq; a.dgi = q.dgi; a.dgh = q.dgh; a.de = q.de;
  a.dprev_belief = q.dprev_belief; a.dprev_state = q.dprev_state; a.min_std = q.min_std;
  a.xbuf = xbuf; a.flags = flags; a.err = err; a.status = q.status;
  a.spin_limit = scan_cs_spin_limit();
  constexpr int DP = 208, TS = 208 * 16;
  const size_t lds_b = (size_t)(4 * TS + 64 * 16 + 2 * 4 * 256 + 16 * 32) * sizeof(float);
  he = hipFuncSetAttribute((const void*)observe_cs_bwd_kernel<13, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL((observe_cs_bwd_kernel<13, 13>), dim3((unsigned)NW, (unsigned)G), dim3(256), lds_b, s, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

}  // namespace repo
