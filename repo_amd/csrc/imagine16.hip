// Persistent, row-tiled imagination rollout (forward and reverse pass) on 16-row tiles.
//
// Imagined rows are independent across ALL H-1 steps (each row is the child of one posterior state), so one
// workgroup owns a tile of rows for the whole rollout: no kernel boundary, no grid synchronisation, activations
// of a step never leave the CU.  Per step a workgroup runs the actor trunk (5 dense layers), the tanh-Normal
// sample, fc_embed_state_action, the GRU and the prior head back to back.
//
// Round 1 used 32-row tiles on v_mfma_f32_32x32x2_f32: 2450 start states -> 77 workgroups on 256 CUs, 107 us per
// step.  The rollout is a chain of 11 dependent layers x 14 steps, so its duration is (steps x per-tile latency),
// not work / chip: here a tile is 16 rows on v_mfma_f32_16x16x4_f32 -- 154 workgroups, half the MFMA work per
// tile and step.  What makes 16 rows affordable is the operand traffic: per FLOP a 16-row tile streams twice the
// weights of a 32-row one, so both MFMA operands move 16 bytes per lane per 4 k-steps:
//   * activations live in LDS "k4-interleaved": element (k, row) at ((k/4)*16 + row)*4 + k%4, so the A fragments
//     of four consecutive k-steps are ONE ds_read_b128 (k-step j of a 16-k block feeds lane quarter q the k
//     16b + 4q + j -- any assignment of the block's 16 k to (quarter, step) is a valid MFMA schedule as long as
//     both operands use it);
//   * weights are packed the same way ([k/4][n][4], zero-padded to a multiple of 16 k; ONE pack launch per call
//     for all ten matrices): the B fragments of four k-steps are one 16-byte global load, issued for the whole
//     K range of a column-tile pair before its first MFMA;
//   * a wave owns column tiles w and w+8 of a layer and runs them as two interleaved accumulator chains
//     (the GRU pairs r with z and the two n products);
//   * the product is formed transposed (weights = the MFMA's A operand, rowtile.h): a lane's accumulator is a
//     "quad", 4 consecutive columns of one row -- one ds_write_b128 into the next layer's tile, one 16-byte raw-buffer
//     store per saved activation (lane offset invariant over the steps, step offset scalar: per-site 64-bit
//     addresses would be hoisted out of the step loop and spilled), biases read as quads from a packed copy behind
//     the weights.
// Padding columns of the activation tiles (K -> multiple of 16) only ever hold finite values and meet zero rows of
// the packed weights.
// (S + A) may be odd (ManiSkill's A = 7): K is padded, not paired.
//
// Reference: TransitionModel.imagine + ActorModel.get_action (models/rssm.py:148-184,
// models/actor_critic.py:76-102) and autograd's backward through them (dreamer.py:357).
#include "rowtile.h"

namespace repo {

struct ImgDims {
  int Hm, N, A, D, Hd, S;
  int C;  // condition width (multitask: the task one-hot, ConditionalTransitionModel.imagine, models/rssm.py:221-249); 0 = none
};

struct ImgFwdArgs {
  ImgDims d;
  const float* wpack;  // all packs, contiguous
  unsigned wbytes;
  unsigned aW[5], aB[5];  // byte offsets of the packed actor weights / biases
  unsigned Wsa, Wih, Whh, Wbp, Wsp;
  unsigned Bsa, Bih, Bhh, Bbp, Bsp;
  const float *belief0, *state0;
  const float* cond;  // (N, C): constant over the rollout; columns [F, F+C) of the actor's input, [X, X+C) of W_sa's
  NoiseSrc eps_act, eps_prior;
  float min_std, a_min_std, a_init_std, a_mean_scale;
  float *featx, *prior_mean, *prior_std, *a_hidden, *a_raw, *a_mean, *a_std, *xsa, *e, *gates, *hp;
  size_t a_layer_rows;
};

// BF / BW / BX / BS: 16-k blocks of F = D + S, of D and Hd, of X = S + A, of 2 S
template <int BF, int BW, int BX, int BS>
__global__ __launch_bounds__(512) void imagine_fwd_kernel(ImgFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Hm = p.d.Hm, N = p.d.N, A = p.d.A, D = p.d.D, Hd = p.d.Hd, S = p.d.S, C = p.d.C;
  const int F = D + S, X = S + A;
  const int XL = X + C;  // row length of the saved [state|action|condition] rows = K of W_sa
  const int FP = pad16(F + C), WP = pad16(max(D, Hd)), XP = pad16(XL), SP = pad16(max(2 * A, 2 * S));
  float* Fa = lds;               // [FP x 16]  current [belief|state] (| condition: columns F.., written once)
  float* Fb = Fa + FP * kR;      // next
  float* HA = Fb + FP * kR;      // [WP x 16]
  float* HB = HA + WP * kR;
  float* XS = HB + WP * kR;      // [XP x 16]  [state|action]
  float* SM = XS + XP * kR;      // [SP x 16]
  const int lds_floats = (2 * FP + 2 * WP + XP + SP) * kR;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane >> 4, lm = lane & 15;  // a lane's quad: tile row lm, columns 16 tile + 4 lq .. + 3
  const int r0 = blockIdx.x * kR;
  const int nr = min(kR, N - r0);
  const size_t rowsAll = p.a_layer_rows;  // row stride between the saved actor layers
  const __amdgpu_buffer_rsrc_t rw = wrsrc(p.wpack, p.wbytes);
  // saved activations, written as quads: lane part of the offset (row r0 + lm) per row stride, step part scalar
  const __amdgpu_buffer_rsrc_t q_hid = arsrc(p.a_hidden), q_raw = arsrc(p.a_raw), q_e = arsrc(p.e),
                               q_gates = arsrc(p.gates), q_featx = arsrc(p.featx), q_hp = arsrc(p.hp);
  const unsigned lrow = (unsigned)(r0 + min(lm, nr - 1));
  const bool lst = lm < nr;  // this lane's row exists

  for (int i = tid; i < lds_floats / 4; i += 512) reinterpret_cast<f32x4v*>(lds)[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // ---- slot 0: start states (row-major global -> tile), also echoed to featx[0]
  for (int i = tid; i < kR * F; i += 512) {
    const int row = i / F, f = i % F;
    if (row < nr) {
      const float v = f < D ? p.belief0[(size_t)(r0 + row) * D + f] : p.state0[(size_t)(r0 + row) * S + (f - D)];
      p.featx[(size_t)(r0 + row) * F + f] = v;
      Fa[ai(f, row)] = v;
    }
  }
  // the condition: nothing below writes a column >= F of either feature tile (belief quads end at D, D % 4 == 0; the
  // state is written element-wise), so it is placed once and meets the actor's fc1 columns F.. on every step
  for (int i = tid; i < kR * C; i += 512) {
    const int row = i / C, c = i % C;
    if (row < nr) Fa[ai(F + c, row)] = Fb[ai(F + c, row)] = p.cond[(size_t)(r0 + row) * C + c];
  }
  __syncthreads();

  float* Fc = Fa;
  float* Fn = Fb;
  // Weight windows.  Every stream is opened one stage before it is used (before the barrier / during the MFMAs of
  // the previous stream); w0 / w1 alternate.
  WWin w0, w1, wc;
  dense_open<BF>(w0, rw, p.aW[0], Hd, wave, lane);
  for (int t = 0; t < Hm; ++t) {
    const size_t rb = (size_t)t * N + r0;  // first global row of this tile at step t
    const unsigned tN = (unsigned)(t * N);  // rows before this step
    // ---------------- actor trunk: 4 ELU layers + linear head
    auto hidden_epi = [&](float* dst, unsigned bias, int layer) {
      return [=](bool ok, int n0, const f32x4v& acc) {
        if (!ok) return;
        const f32x4v bv = ldbias(rw, bias, n0);
        f32x4v v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = elu(acc[r] + bv[r]);
        stq(dst, n0, lm, v);
        if (lst) bstq(q_hid, 4u * (lrow * Hd + n0), 4u * (unsigned)((layer * rowsAll + tN) * Hd), v, 4);
      };
    };
    dense_open<BW>(w1, rw, p.aW[1], Hd, wave, lane);
    dense_run<BF>(Fc, w0, rw, Hd, wave, lane, hidden_epi(HA, p.aB[0], 0));
    __syncthreads();
    dense_open<BW>(w0, rw, p.aW[2], Hd, wave, lane);
    dense_run<BW>(HA, w1, rw, Hd, wave, lane, hidden_epi(HB, p.aB[1], 1));
    __syncthreads();
    dense_open<BW>(w1, rw, p.aW[3], Hd, wave, lane);
    dense_run<BW>(HB, w0, rw, Hd, wave, lane, hidden_epi(HA, p.aB[2], 2));
    __syncthreads();
    dense_open<BW>(w0, rw, p.aW[4], 2 * A, wave, lane);
    dense_run<BW>(HA, w1, rw, Hd, wave, lane, hidden_epi(HB, p.aB[3], 3));
    __syncthreads();
    dense_open<BX>(w1, rw, p.Wsa, D, wave, lane);
    dense_run<BW>(HB, w0, rw, 2 * A, wave, lane, [=](bool ok, int n0, const f32x4v& acc) {
      if (!ok) return;
      const f32x4v v = acc + ldbias(rw, p.aB[4], n0);
      stq(SM, n0, lm, v);
      if (lst) bstq(q_raw, 4u * (lrow * 2 * A + n0), 4u * tN * 2 * A, v, 2 * A - n0);
    });
    __syncthreads();
    // ---------------- tanh-Normal action sample; x = [state, action]
    for (int i = tid; i < kR * XL; i += 512) {
      const int row = i / XL, k = i % XL;
      float v;
      if (k < S) {
        v = Fc[ai(D + k, row)];
      } else if (k >= X) {
        v = Fc[ai(F + k - X, row)];  // pseudo-action = [action | condition]
      } else {
        const int a = k - S;
        const float mu = p.a_mean_scale * tanh_fast(SM[ai(a, row)] / p.a_mean_scale);
        const float sd = softplus(SM[ai(A + a, row)] + p.a_init_std) + p.a_min_std;
        const float ep = row < nr ? p.eps_act.at((rb + row) * A + a) : 0.f;
        v = tanh_fast(fmaf(sd, ep, mu));
        if (row < nr) {
          p.a_mean[(rb + row) * A + a] = mu;
          p.a_std[(rb + row) * A + a] = sd;
        }
      }
      XS[ai(k, row)] = v;
      if (row < nr) p.xsa[(rb + row) * XL + k] = v;
    }
    __syncthreads();
    // ---------------- e = elu(W_sa x + b); meanwhile the GRU's first stream and the prior head's are opened
    const int gtiles = (D + 15) >> 4;
    const int gcol0 = min(wave * 16 + (lane & 15), D - 1), gcol1 = min((wave + kW) * 16 + (lane & 15), D - 1);
    const bool g0 = wave < gtiles, g1 = wave + kW < gtiles;
    wopen<BW>(w0, rw, g0, p.Wih, 3 * D, gcol0, p.Wih, 3 * D, D + gcol0, true, lane);
    dense_run<BX>(XS, w1, rw, D, wave, lane, [=](bool ok, int n0, const f32x4v& acc) {
      if (!ok) return;
      const f32x4v bv = ldbias(rw, p.Bsa, n0);
      f32x4v v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = elu(acc[r] + bv[r]);
      stq(HA, n0, lm, v);
      if (lst) bstq(q_e, 4u * (lrow * D + n0), 4u * tN * D, v, 4);
    });
    __syncthreads();
    // ---------------- GRU: the gate pre-activations of a column tile stay in one wave.  r and z only need
    //                  gi + gh: both products accumulate into one tile.  Streams per tile: (W_ih r, z), (W_hh r, z),
    //                  (W_ih n, W_hh n); window roles swap between the wave's two tiles.
    auto gru_tile = [&](WWin& x, WWin& y, int col, int n, bool more, int ncol) __attribute__((always_inline)) {
      f32x4v ar = {0.f, 0.f, 0.f, 0.f}, az = ar, gin = ar, ghn_ = ar;
      wopen<BW>(y, rw, true, p.Whh, 3 * D, col, p.Whh, 3 * D, D + col, true, lane);
      wrun<BW>(ar, az, HA, HA, x, rw, lane);
      wopen<BW>(x, rw, true, p.Wih, 3 * D, 2 * D + col, p.Whh, 3 * D, 2 * D + col, true, lane);
      wrun<BW>(ar, az, Fc, Fc, y, rw, lane);
      // behind the last stream of this tile: the next tile's first stream, or the prior head's first layer
      if (more) wopen<BW>(y, rw, true, p.Wih, 3 * D, ncol, p.Wih, 3 * D, D + ncol, true, lane);
      else dense_open<BW>(y, rw, p.Wbp, Hd, wave, lane);
      wrun<BW>(gin, ghn_, HA, Fc, x, rw, lane);
      if (n < D) {  // n = the quad's first column.  One gate at a time: few live registers beside the windows
        constexpr int nv = 4;  // D % 4 == 0 (imagine_fused_ok): whole quads
        const unsigned gv = 4u * (lrow * 4 * D + n), gs = 4u * tN * 4 * D;
        const bool st = lst;
        f32x4v rg = ar + ldbias(rw, p.Bih, n) + ldbias(rw, p.Bhh, n);
#pragma unroll
        for (int r = 0; r < 4; ++r) rg[r] = sigmoidf(rg[r]);
        if (st) bstq(q_gates, gv, gs, rg, nv);
        const f32x4v ghn = ghn_ + ldbias(rw, p.Bhh + 8u * D, n);
        if (st) bstq(q_gates, gv + 12u * D, gs, ghn, nv);
        f32x4v ng = gin + ldbias(rw, p.Bih + 8u * D, n);
#pragma unroll
        for (int r = 0; r < 4; ++r) ng[r] = tanh_fast(ng[r] + rg[r] * ghn[r]);
        if (st) bstq(q_gates, gv + 8u * D, gs, ng, nv);
        f32x4v zg = az + ldbias(rw, p.Bih + 4u * D, n) + ldbias(rw, p.Bhh + 4u * D, n);
#pragma unroll
        for (int r = 0; r < 4; ++r) zg[r] = sigmoidf(zg[r]);
        if (st) bstq(q_gates, gv + 4u * D, gs, zg, nv);
        const f32x4v hprev = ldq(Fc, n, lm);
        f32x4v hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) hn[r] = (1.f - zg[r]) * ng[r] + zg[r] * hprev[r];
        stq(Fn, n, lm, hn);
        if (st) bstq(q_featx, 4u * (lrow * F + n), 4u * (tN + N) * F, hn, nv);
      }
    };
    if (g0) gru_tile(w0, w1, gcol0, wave * 16 + 4 * lq, g1, gcol1);
    else dense_open<BW>(w1, rw, p.Wbp, Hd, wave, lane);
    if (g1) gru_tile(w1, w0, gcol1, (wave + kW) * 16 + 4 * lq, false, gcol1);
    wc = g1 ? w0 : w1;  // where the prior head's stream was opened
    __syncthreads();
    // ---------------- prior head
    dense_open<BW>(w1, rw, p.Wsp, 2 * S, wave, lane);
    dense_run<BW>(Fn, wc, rw, Hd, wave, lane, [=](bool ok, int n0, const f32x4v& acc) {
      if (!ok) return;
      const f32x4v bv = ldbias(rw, p.Bbp, n0);
      f32x4v v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = elu(acc[r] + bv[r]);
      stq(HB, n0, lm, v);
      if (lst) bstq(q_hp, 4u * (lrow * Hd + n0), 4u * tN * Hd, v, 4);
    });
    __syncthreads();
    dense_open<BF>(w0, rw, p.aW[0], Hd, wave, lane);  // the next step's first layer
    dense_run<BW>(HB, w1, rw, 2 * S, wave, lane, [=](bool ok, int n0, const f32x4v& acc) {
      if (!ok) return;
      stq(SM, n0, lm, acc + ldbias(rw, p.Bsp, n0));
    });
    __syncthreads();
    for (int i = tid; i < kR * S; i += 512) {
      const int row = i / S, s = i % S;
      const float mu = SM[ai(s, row)];
      const float sd = softplus(SM[ai(S + s, row)]) + p.min_std;
      float smp = mu;
      if (row < nr) {
        const size_t o = (rb + row) * S + s;
        smp = fmaf(sd, p.eps_prior.at(o), mu);
        p.prior_mean[o] = mu;
        p.prior_std[o] = sd;
        p.featx[((size_t)(t + 1) * N + r0 + row) * F + D + s] = smp;
      }
      Fn[ai(D + s, row)] = smp;
    }
    __syncthreads();
    float* tmp = Fc;
    Fc = Fn;
    Fn = tmp;
  }
}

// ------------------------------------------------------------------------------------------ reverse pass
struct ImgBwdArgs {
  ImgDims d;
  // packs with the reduction over the layer's OUTPUT index: W'(n = input index, k = output index)
  const float* wpack;
  unsigned wbytes;
  unsigned Wsp, Wbp, Whh[3], Wih[3], Wsa;
  NoiseSrc eps_act, eps_prior;
  float min_std, a_min_std, a_mean_scale;
  const float *featx, *prior_std, *a_mean, *a_std, *xsa, *e, *gates, *hp;
  const float *dfeat, *dprior_mean, *dprior_std;
  float *d_araw, *dfeat0;
};

template <int BF, int BW, int BX, int BS>
__global__ __launch_bounds__(512) void imagine_bwd_kernel(ImgBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Hm = p.d.Hm, N = p.d.N, A = p.d.A, D = p.d.D, Hd = p.d.Hd, S = p.d.S;
  const int F = D + S, X = S + A;
  const int DP = pad16(D), SP = pad16(max(2 * S, X)), WP = pad16(max(D, Hd));
  float* Gb = lds;              // [DP x 16]  grad on belief_{t+1} (carry + dfeat)
  float* Gs = Gb + DP * kR;     // [pad16(S) x 16]  grad on state_{t+1}
  float* SM = Gs + pad16(S) * kR;  // [SP x 16]   d prior-head outputs, later d [state|action]
  float* X1 = SM + SP * kR;
  float* X2 = X1 + WP * kR;
  float* X3 = X2 + WP * kR;
  float* X4 = X3 + WP * kR;
  const int lds_floats = (DP + pad16(S) + SP + 4 * WP) * kR;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane >> 4, lm = lane & 15;  // a lane's quad: tile row lm, columns 16 tile + 4 lq .. + 3
  const int r0 = blockIdx.x * kR;
  const int nr = min(kR, N - r0);
  const __amdgpu_buffer_rsrc_t rw = wrsrc(p.wpack, p.wbytes);
  const __amdgpu_buffer_rsrc_t q_hp = arsrc(p.hp), q_e = arsrc(p.e);  // saved activations, read as quads
  const unsigned lrow = (unsigned)(r0 + min(lm, nr - 1));

  for (int i = tid; i < lds_floats / 4; i += 512) reinterpret_cast<f32x4v*>(lds)[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  WWin w0, w1;
  dense_open<BS>(w0, rw, p.Wsp, Hd, wave, lane);
  const int gtiles = (D + 15) >> 4;  // <= 2 * kW: results of tiles w, w+8 are held across a barrier
  const bool g0 = wave < gtiles, g1 = wave + kW < gtiles;
  const int gcol0 = min(wave * 16 + (lane & 15), D - 1), gcol1 = min((wave + kW) * 16 + (lane & 15), D - 1);
  for (int t = Hm - 1; t >= 0; --t) {
    const size_t rb = (size_t)t * N + r0;
    const unsigned tN = (unsigned)(t * N);
    // ---- G += dfeat[t]
    for (int i = tid; i < kR * F; i += 512) {
      const int row = i / F, f = i % F;
      if (row < nr) {
        float* g = f < D ? &Gb[ai(f, row)] : &Gs[ai(f - D, row)];
        *g += p.dfeat[(rb + row) * F + f];
      }
    }
    __syncthreads();
    // ---- prior head: sample / mean / std gradients -> d [mean | raw_std]
    for (int i = tid; i < kR * S; i += 512) {
      const int row = i / S, s = i % S;
      float gm = 0.f, gr = 0.f;
      if (row < nr) {
        const size_t o = (rb + row) * S + s;
        const float ds = Gs[ai(s, row)];
        gm = ds + (p.dprior_mean ? p.dprior_mean[o] : 0.f);
        const float gs = fmaf(ds, p.eps_prior.at(o), p.dprior_std ? p.dprior_std[o] : 0.f);
        gr = gs * (-expm1f(-(p.prior_std[o] - p.min_std)));
      }
      SM[ai(s, row)] = gm;
      SM[ai(S + s, row)] = gr;
    }
    __syncthreads();
    // ---- X1 = (d out @ W_sp) * elu'(hp)
    dense_open<BW>(w1, rw, p.Wbp, D, wave, lane);
    dense_run<BS>(SM, w0, rw, Hd, wave, lane, [=](bool ok, int n0, const f32x4v& acc) {
      if (!ok) return;
      const f32x4v h = bldq(q_hp, 4u * (lrow * Hd + n0), 4u * tN * Hd);
      f32x4v v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[r] * elu_grad_from_out(h[r]);
      stq(X1, n0, lm, v);
    });
    __syncthreads();
    // ---- X2 = d belief_{t+1} = Gb + X1 @ W_bp
    wopen<BW>(w0, rw, g0, p.Whh[0], D, gcol0, p.Wih[0], D, gcol0, true, lane);  // the gate products' first stream
    dense_run<BW>(X1, w1, rw, D, wave, lane, [=](bool ok, int n0, const f32x4v& acc) {
      if (!ok) return;
      stq(X2, n0, lm, acc + ldq(Gb, n0, lm));
    });
    __syncthreads();
    // ---- GRU gates (element-wise): X2 <- g_r, X1 <- g_z, X3 <- g_n, X4 <- g_n * r, Gb <- d * z
    for (int i = tid; i < kR * D; i += 512) {
      const int row = i / D, n = i % D;
      float g_r = 0.f, g_z = 0.f, g_n = 0.f, g_hn = 0.f, dhp = 0.f;
      if (row < nr) {
        const float* g = p.gates + (rb + row) * 4 * D;
        const float rg = g[n], zg = g[D + n], ng = g[2 * D + n], ghn = g[3 * D + n];
        const float hprev = p.featx[((size_t)t * N + r0 + row) * F + n];
        const float d = X2[ai(n, row)];
        g_n = d * (1.f - zg) * (1.f - ng * ng);
        g_z = d * (hprev - ng) * zg * (1.f - zg);
        g_r = g_n * ghn * rg * (1.f - rg);
        g_hn = g_n * rg;
        dhp = d * zg;
      }
      X2[ai(n, row)] = g_r;
      X1[ai(n, row)] = g_z;
      X3[ai(n, row)] = g_n;
      X4[ai(n, row)] = g_hn;
      Gb[ai(n, row)] = dhp;
    }
    __syncthreads();
    // ---- through W_hh into belief_t (new carry: ah) and through W_ih into e (ae); both from the same tiles.
    //      Streams per tile: (W_hh r, W_ih r) on g_r, (z, z) on g_z, (W_hh n on g_n*r, W_ih n on g_n).
    {
      f32x4v ah0 = {0.f, 0.f, 0.f, 0.f}, ae0 = ah0, ah1 = ah0, ae1 = ah0;
      auto gate_tile = [&](WWin& x, WWin& y, f32x4v& ah, f32x4v& ae, int col, bool more, int ncol) __attribute__((always_inline)) {
        wopen<BW>(y, rw, true, p.Whh[1], D, col, p.Wih[1], D, col, true, lane);
        wrun<BW>(ah, ae, X2, X2, x, rw, lane);
        wopen<BW>(x, rw, true, p.Whh[2], D, col, p.Wih[2], D, col, true, lane);
        wrun<BW>(ah, ae, X1, X1, y, rw, lane);
        wopen<BW>(y, rw, more, p.Whh[0], D, ncol, p.Wih[0], D, ncol, true, lane);
        wrun<BW>(ah, ae, X4, X3, x, rw, lane);
      };
      if (g0) gate_tile(w0, w1, ah0, ae0, gcol0, g1, gcol1);
      if (g1) gate_tile(w1, w0, ah1, ae1, gcol1, false, gcol1);
      // both windows are free again: open the last layer of this step
      dense_open<BW>(w0, rw, p.Wsa, X, wave, lane);
      __syncthreads();  // every wave has finished reading X1..X4
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n0 = (wave + i * kW) * 16 + 4 * lq;
        const f32x4v& ah = i ? ah1 : ah0;
        const f32x4v& ae = i ? ae1 : ae0;
        if ((i ? g1 : g0) && n0 < D) {
          stq(Gb, n0, lm, ldq(Gb, n0, lm) + ah);
          const f32x4v ev = bldq(q_e, 4u * (lrow * D + n0), 4u * tN * D);
          f32x4v v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ae[r] * elu_grad_from_out(ev[r]);
          stq(X4, n0, lm, v);  // d pre-activation of fc_embed_state_action
        }
      }
    }
    __syncthreads();
    // ---- d [state_t | action_t] = X4 @ W_sa
    dense_open<BS>(w1, rw, p.Wsp, Hd, wave, lane);  // the next (earlier) step's first layer
    dense_run<BW>(X4, w0, rw, X, wave, lane, [=](bool ok, int n0, const f32x4v& acc) {
      if (!ok) return;
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // a quad may straddle the state | action boundary
        const int n = n0 + r;
        if (n < S) Gs[ai(n, lm)] = acc[r];
        else if (n < X) SM[ai(n, lm)] = acc[r];
      }
    });
    __syncthreads();
    // ---- tanh-Normal sample backward -> gradient at the actor trunk's output of step t
    for (int i = tid; i < kR * A; i += 512) {
      const int row = i / A, a = i % A;
      if (row < nr) {
        const size_t o = (rb + row) * A + a;
        const float act = p.xsa[(rb + row) * (X + p.d.C) + S + a];
        const float du = SM[ai(S + a, row)] * (1.f - act * act);
        const float tm = p.a_mean[o] / p.a_mean_scale;
        p.d_araw[(rb + row) * 2 * A + a] = du * (1.f - tm * tm);
        p.d_araw[(rb + row) * 2 * A + A + a] = du * p.eps_act.at(o) * (-expm1f(-(p.a_std[o] - p.a_min_std)));
      }
    }
    __syncthreads();
    // window roles for the next iteration: its first layer was opened into w1
    {
      WWin tmp = w0;
      w0 = w1;
      w1 = tmp;
    }
  }
  if (p.dfeat0) {
    for (int i = tid; i < kR * F; i += 512) {
      const int row = i / F, f = i % F;
      if (row < nr) p.dfeat0[(size_t)(r0 + row) * F + f] = f < D ? Gb[ai(f, row)] : Gs[ai(f - D, row)];
    }
  }
}

// ------------------------------------------------------------------------------------------ host side
// The kernels are instantiated for the reference architecture's block counts: belief = hidden = 200 (13 blocks of
// 16 k), belief + state = 230 (15), state + action 33..48 (3: A = 6 and ManiSkill's A = 7), 2 * state = 60 (4);
// any other size runs the per-step engine (imagine.hip).
constexpr int kBF = 15, kBW = 13, kBX = 3, kBS = 4;
bool imagine_fused_ok(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, int n_actor_layers, int64_t C) {
  auto blk = [](int64_t k) { return pad16((int)k) >> 4; };
  // a condition of C columns rides in the K padding of the two layers it widens: 230 + C <= 240, S + A + C <= 48
  return n_actor_layers == 5 && D % 4 == 0 && Hd % 4 == 0 && blk(D) == kBW && blk(Hd) == kBW && blk(D + S) == kBF &&
         blk(D + S + C) == kBF && blk(S + A) == kBX && blk(S + A + C) == kBX && C >= 0 &&
         blk(2 * S) == kBS && 2 * A <= 16 && (Hm + 1) * N * 4 * D < kMaxIdx;
}

size_t imagine_fused_fwd_ws_floats(int64_t A, int64_t D, int64_t Hd, int64_t S) {
  // (K is padded to 16: a condition inside the padding -- imagine_fused_ok -- does not change these)
  return pack_floats(Hd, D + S) + 3 * pack_floats(Hd, Hd) + pack_floats(2 * A, Hd) + pack_floats(D, S + A) +
         2 * pack_floats(3 * D, D) + pack_floats(Hd, D) + pack_floats(2 * S, Hd) +
         // bias vectors: 4 actor hidden + head, fc_embed_state_action, the GRU's two, prior hidden, prior out
         5 * pack_floats(1, Hd) + pack_floats(1, 2 * A) + pack_floats(1, D) + 2 * pack_floats(1, 3 * D) +
         pack_floats(1, 2 * S);
}
size_t imagine_fused_bwd_ws_floats(int64_t A, int64_t D, int64_t Hd, int64_t S) {
  return pack_floats(Hd, 2 * S) + pack_floats(D, Hd) + 6 * pack_floats(D, D) + pack_floats(S + A, D);
}

int imagine_fused_fwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S,
                      const float* const* rp, const float* const* ap, const float* belief0, const float* state0,
                      const float* cond, int64_t C,
                      NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std, float a_init_std,
                      float a_mean_scale, float* featx, float* prior_mean, float* prior_std, float* a_hidden,
                      int64_t a_layer_rows, float* a_raw, float* a_mean, float* a_std, float* xsa, float* e,
                      float* gates, float* hp, void* ws, hipStream_t stream) {
  const int F = (int)(D + S + C), X = (int)(S + A + C);  // K of the two widened layers (C = 0: the reference's)
  float* w = (float*)ws;
  ImgFwdArgs a;
  a.d = ImgDims{(int)Hm, (int)N, (int)A, (int)D, (int)Hd, (int)S, (int)C};
  a.cond = cond;
  PackArgs pa;
  pa.njobs = 0;
  float* const w_begin = w;
  auto add = [&](const float* src, int Nn, int K, int sn, int sk) {
    pa.job[pa.njobs++] = PackJob{src, w, Nn, K, sn, sk};
    const unsigned r = (unsigned)((w - w_begin) * sizeof(float));
    w += pack_floats(Nn, K);
    return r;
  };
  const int kin[5] = {F, (int)Hd, (int)Hd, (int)Hd, (int)Hd};
  const int nout[5] = {(int)Hd, (int)Hd, (int)Hd, (int)Hd, (int)(2 * A)};
  for (int l = 0; l < 5; ++l) {
    a.aW[l] = add(ap[2 * l], nout[l], kin[l], kin[l], 1);
    a.aB[l] = add(ap[2 * l + 1], 1, nout[l], 0, 1);  // a vector: copied, zero-padded to 16
  }
  a.Wsa = add(rp[0], (int)D, X, X, 1);
  a.Wih = add(rp[2], (int)(3 * D), (int)D, (int)D, 1);
  a.Whh = add(rp[3], (int)(3 * D), (int)D, (int)D, 1);
  a.Wbp = add(rp[6], (int)Hd, (int)D, (int)D, 1);
  a.Wsp = add(rp[8], (int)(2 * S), (int)Hd, (int)Hd, 1);
  a.Bsa = add(rp[1], 1, (int)D, 0, 1);
  a.Bih = add(rp[4], 1, (int)(3 * D), 0, 1);
  a.Bhh = add(rp[5], 1, (int)(3 * D), 0, 1);
  a.Bbp = add(rp[7], 1, (int)Hd, 0, 1);
  a.Bsp = add(rp[9], 1, (int)(2 * S), 0, 1);
  a.wpack = w_begin;
  a.wbytes = (unsigned)((w - w_begin) * sizeof(float));
  int rc = launch_pack(pa, stream);
  if (rc) return rc;
  a.belief0 = belief0; a.state0 = state0; a.eps_act = eps_act; a.eps_prior = eps_prior;
  a.min_std = min_std; a.a_min_std = a_min_std; a.a_init_std = a_init_std; a.a_mean_scale = a_mean_scale;
  a.featx = featx; a.prior_mean = prior_mean; a.prior_std = prior_std; a.a_hidden = a_hidden; a.a_raw = a_raw;
  a.a_mean = a_mean; a.a_std = a_std; a.xsa = xsa; a.e = e; a.gates = gates; a.hp = hp;
  a.a_layer_rows = (size_t)a_layer_rows;
  const int W = (int)(D > Hd ? D : Hd), SMr = (int)(2 * A > 2 * S ? 2 * A : 2 * S);
  const size_t lds_b = (size_t)(2 * pad16(F) + 2 * pad16(W) + pad16(X) + pad16(SMr)) * kR * sizeof(float);
  hipError_t he = hipFuncSetAttribute((const void*)imagine_fwd_kernel<kBF, kBW, kBX, kBS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_b);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL((imagine_fwd_kernel<kBF, kBW, kBX, kBS>), dim3((unsigned)((N + kR - 1) / kR)), dim3(512), lds_b, stream, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

int imagine_fused_bwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, const float* const* rp,
                      int64_t C, NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std,
                      float a_mean_scale, const float* featx, const float* prior_std, const float* a_mean,
                      const float* a_std, const float* xsa, const float* e, const float* gates, const float* hp,
                      const float* dfeat, const float* dprior_mean, const float* dprior_std, float* d_araw,
                      float* dfeat0, void* ws, hipStream_t stream) {
  const int X = (int)(S + A);
  float* w = (float*)ws;
  ImgBwdArgs a;
  a.d = ImgDims{(int)Hm, (int)N, (int)A, (int)D, (int)Hd, (int)S, (int)C};
  PackArgs pa;
  pa.njobs = 0;
  // W'(n = input index, k = output index) = native[k * ld + n]
  float* const w_begin = w;
  auto add = [&](const float* src, int Nin, int Kout, int ld) {
    pa.job[pa.njobs++] = PackJob{src, w, Nin, Kout, 1, ld};
    const unsigned r = (unsigned)((w - w_begin) * sizeof(float));
    w += pack_floats(Nin, Kout);
    return r;
  };
  a.Wsp = add(rp[8], (int)Hd, (int)(2 * S), (int)Hd);
  a.Wbp = add(rp[6], (int)D, (int)Hd, (int)D);
  for (int g = 0; g < 3; ++g) {
    a.Whh[g] = add(rp[3] + (size_t)g * D * D, (int)D, (int)D, (int)D);
    a.Wih[g] = add(rp[2] + (size_t)g * D * D, (int)D, (int)D, (int)D);
  }
  a.Wsa = add(rp[0], X, (int)D, X + (int)C);  // the rows of d [state|action] only: the condition takes no gradient
  a.wpack = w_begin;
  a.wbytes = (unsigned)((w - w_begin) * sizeof(float));
  int rc = launch_pack(pa, stream);
  if (rc) return rc;
  a.eps_act = eps_act; a.eps_prior = eps_prior;
  a.min_std = min_std; a.a_min_std = a_min_std; a.a_mean_scale = a_mean_scale;
  a.featx = featx; a.prior_std = prior_std; a.a_mean = a_mean; a.a_std = a_std; a.xsa = xsa; a.e = e;
  a.gates = gates; a.hp = hp; a.dfeat = dfeat; a.dprior_mean = dprior_mean; a.dprior_std = dprior_std;
  a.d_araw = d_araw; a.dfeat0 = dfeat0;
  const int W = (int)(D > Hd ? D : Hd), SMr = (int)(2 * S > X ? 2 * S : X);
  const size_t lds_b = (size_t)(pad16((int)D) + pad16((int)S) + pad16(SMr) + 4 * pad16(W)) * kR * sizeof(float);
  hipError_t he = hipFuncSetAttribute((const void*)imagine_bwd_kernel<kBF, kBW, kBX, kBS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_b);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL((imagine_bwd_kernel<kBF, kBW, kBX, kBS>), dim3((unsigned)((N + kR - 1) / kR)), dim3(512), lds_b, stream, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

}  // namespace repo
