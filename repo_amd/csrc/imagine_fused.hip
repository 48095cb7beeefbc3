// Persistent, row-tiled imagination rollout (forward and reverse pass).
//
// Imagined rows are independent across ALL H-1 steps (each row is the child of one posterior
// state), so one workgroup owns 32 rows for the whole rollout: no kernel boundary, no grid
// synchronisation, activations of a step never leave the CU.  Per step a workgroup runs the actor
// trunk (5 dense layers), the tanh-Normal sample, fc_embed_state_action, the GRU and the prior head
// back to back:
//   * activations live in LDS as [feature][row] (row stride 33 dwords): that is directly the A
//     operand layout of v_mfma_f32_32x32x2_f32 (lane = row, conflict-free), and the epilogue's
//     column-per-lane writes (stride 33) are conflict-free too;
//   * weights are streamed from L2-resident [k][n] copies straight into the B operand (lane = n,
//     two 128-byte segments per instruction), double-buffered in registers 8 k-steps ahead;
//   * 8 waves split the output columns of a layer (32 per wave-tile); the three gates of the GRU
//     for a column tile are accumulated by the same wave so the gate math runs in its epilogue.
// This replaces ~13 launches per step of 2450-row GEMMs that ran one wave per SIMD and were bound
// by non-MFMA instruction latency (13 us per launch) rather than by the matrix pipe.
//
// Reference: TransitionModel.imagine + ActorModel.get_action (models/rssm.py:148-184,
// models/actor_critic.py:76-102) and autograd's backward through them (dreamer.py:357).
#include <stdlib.h>

#include "common.h"

namespace repo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kLD = 33;        // LDS row stride (dwords) of a [feature][32 rows] activation tile
constexpr int kRows = 32;      // rows per workgroup
constexpr int kWaves = 8;      // 512 threads
constexpr int kChunk = 10;     // k-steps (of 2) per register buffer of the weight stream (K/2 = 100, 30: no tail)

// acc (32 rows x 32 columns) += A[32 x K] * Wt[K x 32-column slice].
//   A  : LDS, [k][kLD] (k-major), K even
//   Wt : global, element (k, n) at Wt[k*ldw + n]; `col` = this lane's (clamped) column
__device__ __forceinline__ void mma_cols(f32x16& acc, const float* __restrict__ A, int K,
                                         const float* __restrict__ Wt, int ldw, int col, int li, int lh) {
  // Addressing is split into a wave-uniform part (k: scalar registers / immediates) and a
  // per-lane part fixed for the whole call (lh, column/row): ~1 non-MFMA instruction per MFMA
  // in the steady state instead of ~19 (measured: 9.5 VALU + 9.3 SALU per MFMA before).
  const int nks = K >> 1;
  const int nfull = nks / kChunk, rem = nks - nfull * kChunk;
  const int lane_w = lh * ldw + col;
  const int lane_a = lh * kLD + li;
  float b0[kChunk], b1[kChunk];
  auto loadb = [&](float (&b)[kChunk], int chunk) __attribute__((always_inline)) {
    const int cc = min(chunk, nfull - 1);  // prefetch past the end re-reads the last full chunk
    const float* ub = Wt + (size_t)(2 * kChunk * cc) * ldw;
#pragma unroll
    for (int u = 0; u < kChunk; ++u) b[u] = (ub + (size_t)(2 * u) * ldw)[lane_w];
  };
  auto run = [&](const float (&b)[kChunk], int chunk) __attribute__((always_inline)) {
    const float* ua = A + 2 * kChunk * chunk * kLD + lane_a;
    float a[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) a[u] = ua[2 * u * kLD];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  };
  if (nfull > 0) {
    loadb(b0, 0);
    for (int ch = 0; ch < nfull; ch += 2) {
      loadb(b1, ch + 1);  // unconditional (clamped): keeps the vmcnt bookkeeping static
      __builtin_amdgcn_sched_barrier(0);
      run(b0, ch);
      __builtin_amdgcn_sched_barrier(0);
      if (ch + 1 >= nfull) break;
      loadb(b0, ch + 2);
      __builtin_amdgcn_sched_barrier(0);
      run(b1, ch + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (int u = 0; u < rem; ++u) {  // short tail (K/2 not a multiple of kChunk)
    const int ks = nfull * kChunk + u;
    const float bv = Wt[(size_t)(2 * ks) * ldw + lane_w];
    const float av = A[2 * ks * kLD + lane_a];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
}

__device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// Dense layer on the row tile: for each 32-column tile owned by this wave, acc = A * Wt and then
// epi(valid, n, acc) with n = this lane's column, rows m_r = (r&3) + 8*(r>>2) + 4*lh.
template <class Epi>
__device__ __forceinline__ void dense_cols(const float* A, int K, const float* Wt, int ldw, int Nout, int wave, int li,
                                           int lh, Epi epi) {
  const int ntiles = (Nout + 31) >> 5;
  for (int ct = wave; ct < ntiles; ct += kWaves) {
    const int c0 = ct * 32;
    const int valid = min(32, Nout - c0);
    f32x16 acc;
    zero_acc(acc);
    mma_cols(acc, A, K, Wt, ldw, c0 + min(li, valid - 1), li, lh);
    epi(li < valid, c0 + li, acc);
  }
}

struct ImgDims {
  int Hm, N, A, D, Hd, S;
};

struct ImgFwdArgs {
  ImgDims d;
  const float* aWt[5];  // actor weights transposed [k][n]
  const float* ab[5];
  const float *WsaT, *bsa, *WihT, *WhhT, *bih, *bhh, *WbpT, *bbp, *WspT, *bsp;
  const float *belief0, *state0;
  NoiseSrc eps_act, eps_prior;
  float min_std, a_min_std, a_init_std, a_mean_scale;
  float *featx, *prior_mean, *prior_std, *a_hidden, *a_raw, *a_mean, *a_std, *xsa, *e, *gates, *hp;
  size_t a_layer_rows;
};

__global__ __launch_bounds__(512) void imagine_fwd_kernel(ImgFwdArgs p) {
  extern __shared__ float lds[];
  const int Hm = p.d.Hm, N = p.d.N, A = p.d.A, D = p.d.D, Hd = p.d.Hd, S = p.d.S;
  const int F = D + S, X = S + A;
  const int W = max(D, Hd);
  float* Fa = lds;                 // [F][kLD]   current [belief|state]
  float* Fb = Fa + F * kLD;        // [F][kLD]   next
  float* HA = Fb + F * kLD;        // [W][kLD]
  float* HB = HA + W * kLD;        // [W][kLD]
  float* XS = HB + W * kLD;        // [X][kLD]   [state|action]
  float* SM = XS + X * kLD;        // [max(2A,2S)][kLD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * kRows;
  const int nr = min(kRows, N - r0);
  const size_t rowsAll = p.a_layer_rows;  // row stride between the saved actor layers

  // ---- slot 0: start states (row-major global -> [feature][row] LDS), also echoed to featx[0]
  for (int i = tid; i < kRows * F; i += blockDim.x) {
    const int row = i / F, f = i % F;
    float v = 0.f;
    if (row < nr) {
      v = f < D ? p.belief0[(size_t)(r0 + row) * D + f] : p.state0[(size_t)(r0 + row) * S + (f - D)];
      p.featx[(size_t)(r0 + row) * F + f] = v;
    }
    Fa[f * kLD + row] = v;
  }
  __syncthreads();

  float* Fc = Fa;
  float* Fn = Fb;
  for (int t = 0; t < Hm; ++t) {
    const size_t rb = (size_t)t * N + r0;  // first global row of this tile at step t
    // ---------------- actor trunk: 4 ELU layers + linear head
    auto hidden_epi = [&](float* dst, const float* bias, float* save) {
      return [=](bool ok, int n, const f32x16& acc) {
        if (!ok) return;
        const float bv = bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float v = elu(acc[r] + bv);
          dst[n * kLD + m] = v;
          if (m < nr) save[(rb + m) * Hd + n] = v;
        }
      };
    };
    dense_cols(Fc, F, p.aWt[0], Hd, Hd, wave, li, lh, hidden_epi(HA, p.ab[0], p.a_hidden));
    __syncthreads();
    dense_cols(HA, Hd, p.aWt[1], Hd, Hd, wave, li, lh, hidden_epi(HB, p.ab[1], p.a_hidden + rowsAll * Hd));
    __syncthreads();
    dense_cols(HB, Hd, p.aWt[2], Hd, Hd, wave, li, lh, hidden_epi(HA, p.ab[2], p.a_hidden + 2 * rowsAll * Hd));
    __syncthreads();
    dense_cols(HA, Hd, p.aWt[3], Hd, Hd, wave, li, lh, hidden_epi(HB, p.ab[3], p.a_hidden + 3 * rowsAll * Hd));
    __syncthreads();
    dense_cols(HB, Hd, p.aWt[4], 2 * A, 2 * A, wave, li, lh, [=](bool ok, int n, const f32x16& acc) {
      if (!ok) return;
      const float bv = p.ab[4][n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = acc[r] + bv;
        SM[n * kLD + m] = v;
        if (m < nr) p.a_raw[(rb + m) * 2 * A + n] = v;
      }
    });
    __syncthreads();
    // ---------------- tanh-Normal action sample; x = [state, action]
    for (int i = tid; i < kRows * X; i += blockDim.x) {
      const int row = i / X, k = i % X;
      float v;
      if (k < S) {
        v = Fc[(D + k) * kLD + row];
      } else {
        const int a = k - S;
        const float mu = p.a_mean_scale * tanh_fast(SM[a * kLD + row] / p.a_mean_scale);
        const float sd = softplus(SM[(A + a) * kLD + row] + p.a_init_std) + p.a_min_std;
        const float ep = row < nr ? p.eps_act.at((rb + row) * A + a) : 0.f;
        v = tanh_fast(fmaf(sd, ep, mu));
        if (row < nr) {
          p.a_mean[(rb + row) * A + a] = mu;
          p.a_std[(rb + row) * A + a] = sd;
        }
      }
      XS[k * kLD + row] = v;
      if (row < nr) p.xsa[(rb + row) * X + k] = v;
    }
    __syncthreads();
    // ---------------- e = elu(W_sa x + b)
    dense_cols(XS, X, p.WsaT, D, D, wave, li, lh, [=](bool ok, int n, const f32x16& acc) {
      if (!ok) return;
      const float bv = p.bsa[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = elu(acc[r] + bv);
        HA[n * kLD + m] = v;
        if (m < nr) p.e[(rb + m) * D + n] = v;
      }
    });
    __syncthreads();
    // ---------------- GRU: the six gate pre-activations of a column tile stay in one wave
    {
      const int ntiles = (D + 31) >> 5;
      for (int ct = wave; ct < ntiles; ct += kWaves) {
        const int c0 = ct * 32;
        const int valid = min(32, D - c0);
        const int col = c0 + min(li, valid - 1);
        // r and z only need gi+gh: accumulate both products into one tile (4 accumulators, not 6)
        f32x16 ar, az, gin, ghn_;
        zero_acc(ar);
        zero_acc(az);
        zero_acc(gin);
        zero_acc(ghn_);
        mma_cols(ar, HA, D, p.WihT, 3 * D, col, li, lh);
        mma_cols(ar, Fc, D, p.WhhT, 3 * D, col, li, lh);
        mma_cols(az, HA, D, p.WihT + D, 3 * D, col, li, lh);
        mma_cols(az, Fc, D, p.WhhT + D, 3 * D, col, li, lh);
        mma_cols(gin, HA, D, p.WihT + 2 * D, 3 * D, col, li, lh);
        mma_cols(ghn_, Fc, D, p.WhhT + 2 * D, 3 * D, col, li, lh);
        if (li < valid) {
          const int n = c0 + li;
          const float br = p.bih[n] + p.bhh[n], bz = p.bih[D + n] + p.bhh[D + n];
          const float bin = p.bih[2 * D + n], bhn = p.bhh[2 * D + n];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float rg = sigmoidf(ar[r] + br);
            const float zg = sigmoidf(az[r] + bz);
            const float ghn = ghn_[r] + bhn;
            const float ng = tanh_fast(gin[r] + bin + rg * ghn);
            const float hprev = Fc[n * kLD + m];
            const float hn = (1.f - zg) * ng + zg * hprev;
            Fn[n * kLD + m] = hn;
            if (m < nr) {
              float* g = p.gates + (rb + m) * 4 * D;
              g[n] = rg;
              g[D + n] = zg;
              g[2 * D + n] = ng;
              g[3 * D + n] = ghn;
              p.featx[((size_t)(t + 1) * N + r0 + m) * F + n] = hn;
            }
          }
        }
      }
    }
    __syncthreads();
    // ---------------- prior head
    dense_cols(Fn, D, p.WbpT, Hd, Hd, wave, li, lh, [=](bool ok, int n, const f32x16& acc) {
      if (!ok) return;
      const float bv = p.bbp[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = elu(acc[r] + bv);
        HB[n * kLD + m] = v;
        if (m < nr) p.hp[(rb + m) * Hd + n] = v;
      }
    });
    __syncthreads();
    dense_cols(HB, Hd, p.WspT, 2 * S, 2 * S, wave, li, lh, [=](bool ok, int n, const f32x16& acc) {
      if (!ok) return;
      const float bv = p.bsp[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        SM[n * kLD + m] = acc[r] + bv;
      }
    });
    __syncthreads();
    for (int i = tid; i < kRows * S; i += blockDim.x) {
      const int row = i / S, s = i % S;
      const float mu = SM[s * kLD + row];
      const float sd = softplus(SM[(S + s) * kLD + row]) + p.min_std;
      float smp = mu;
      if (row < nr) {
        const size_t o = (rb + row) * S + s;
        smp = fmaf(sd, p.eps_prior.at(o), mu);
        p.prior_mean[o] = mu;
        p.prior_std[o] = sd;
        p.featx[((size_t)(t + 1) * N + r0 + row) * F + D + s] = smp;
      }
      Fn[(D + s) * kLD + row] = smp;
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
  const float *Wsa, *Wih, *Whh, *Wbp, *Wsp;  // native [out][in] layouts = [k][n] for the transposed products
  NoiseSrc eps_act, eps_prior;
  float min_std, a_min_std, a_mean_scale;
  const float *featx, *prior_std, *a_mean, *a_std, *xsa, *e, *gates, *hp;
  const float *dfeat, *dprior_mean, *dprior_std;
  float *d_araw, *dfeat0;
};

__global__ __launch_bounds__(512) void imagine_bwd_kernel(ImgBwdArgs p) {
  extern __shared__ float lds[];
  const int Hm = p.d.Hm, N = p.d.N, A = p.d.A, D = p.d.D, Hd = p.d.Hd, S = p.d.S;
  const int F = D + S, X = S + A;
  const int W = max(D, Hd);
  float* Gb = lds;              // [D][kLD]  grad on belief_{t+1} (carry + dfeat)
  float* Gs = Gb + D * kLD;     // [S][kLD]  grad on state_{t+1}
  float* SM = Gs + S * kLD;     // [max(2S, X)][kLD]   d prior-head outputs, later d [state|action]
  float* X1 = SM + max(2 * S, X) * kLD;
  float* X2 = X1 + W * kLD;
  float* X3 = X2 + W * kLD;
  float* X4 = X3 + W * kLD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * kRows;
  const int nr = min(kRows, N - r0);

  for (int i = tid; i < F * kLD; i += blockDim.x) Gb[i] = 0.f;  // Gb and Gs are contiguous
  __syncthreads();

  for (int t = Hm - 1; t >= 0; --t) {
    const size_t rb = (size_t)t * N + r0;
    // ---- G += dfeat[t]
    for (int i = tid; i < kRows * F; i += blockDim.x) {
      const int row = i / F, f = i % F;
      if (row < nr) Gb[f * kLD + row] += p.dfeat[(rb + row) * F + f];
    }
    __syncthreads();
    // ---- prior head: sample / mean / std gradients -> d [mean | raw_std]
    for (int i = tid; i < kRows * S; i += blockDim.x) {
      const int row = i / S, s = i % S;
      float gm = 0.f, gr = 0.f;
      if (row < nr) {
        const size_t o = (rb + row) * S + s;
        const float ds = Gs[s * kLD + row];
        gm = ds + (p.dprior_mean ? p.dprior_mean[o] : 0.f);
        const float gs = fmaf(ds, p.eps_prior.at(o), p.dprior_std ? p.dprior_std[o] : 0.f);
        gr = gs * (-expm1f(-(p.prior_std[o] - p.min_std)));
      }
      SM[s * kLD + row] = gm;
      SM[(S + s) * kLD + row] = gr;
    }
    __syncthreads();
    // ---- X1 = (d out @ W_sp) * elu'(hp)
    dense_cols(SM, 2 * S, p.Wsp, Hd, Hd, wave, li, lh, [=](bool ok, int n, const f32x16& acc) {
      if (!ok) return;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float h = m < nr ? p.hp[(rb + m) * Hd + n] : 0.f;
        X1[n * kLD + m] = acc[r] * elu_grad_from_out(h);
      }
    });
    __syncthreads();
    // ---- X2 = d belief_{t+1} = Gb + X1 @ W_bp
    dense_cols(X1, Hd, p.Wbp, D, D, wave, li, lh, [=](bool ok, int n, const f32x16& acc) {
      if (!ok) return;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        X2[n * kLD + m] = acc[r] + Gb[n * kLD + m];
      }
    });
    __syncthreads();
    // ---- GRU gates (element-wise): X2 <- g_r, X1 <- g_z, X3 <- g_n, X4 <- g_n * r, Gb <- d * z
    for (int i = tid; i < kRows * D; i += blockDim.x) {
      const int row = i / D, n = i % D;
      float g_r = 0.f, g_z = 0.f, g_n = 0.f, g_hn = 0.f, dhp = 0.f;
      if (row < nr) {
        const float* g = p.gates + (rb + row) * 4 * D;
        const float rg = g[n], zg = g[D + n], ng = g[2 * D + n], ghn = g[3 * D + n];
        const float hprev = p.featx[((size_t)t * N + r0 + row) * F + n];
        const float d = X2[n * kLD + row];
        g_n = d * (1.f - zg) * (1.f - ng * ng);
        g_z = d * (hprev - ng) * zg * (1.f - zg);
        g_r = g_n * ghn * rg * (1.f - rg);
        g_hn = g_n * rg;
        dhp = d * zg;
      }
      X2[n * kLD + row] = g_r;
      X1[n * kLD + row] = g_z;
      X3[n * kLD + row] = g_n;
      X4[n * kLD + row] = g_hn;
      Gb[n * kLD + row] = dhp;
    }
    __syncthreads();
    // ---- through W_hh into belief_t (new carry) and through W_ih into e; both from the same tiles
    {
      const int ntiles = (D + 31) >> 5;  // <= kWaves: one tile per wave, results held across the barrier
      const int ct = wave;
      const bool have = ct < ntiles;
      const int c0 = ct * 32;
      const int valid = have ? min(32, D - c0) : 1;
      const int col = have ? c0 + min(li, valid - 1) : 0;
      f32x16 ah, ae;
      zero_acc(ah);
      zero_acc(ae);
      if (have) {
        mma_cols(ah, X2, D, p.Whh, D, col, li, lh);
        mma_cols(ah, X1, D, p.Whh + (size_t)D * D, D, col, li, lh);
        mma_cols(ah, X4, D, p.Whh + (size_t)2 * D * D, D, col, li, lh);
        mma_cols(ae, X2, D, p.Wih, D, col, li, lh);
        mma_cols(ae, X1, D, p.Wih + (size_t)D * D, D, col, li, lh);
        mma_cols(ae, X3, D, p.Wih + (size_t)2 * D * D, D, col, li, lh);
      }
      __syncthreads();  // every wave has finished reading X1..X4
      if (have && li < valid) {
        const int n = c0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
          Gb[n * kLD + m] += ah[r];
          const float ev = m < nr ? p.e[(rb + m) * D + n] : 0.f;
          X4[n * kLD + m] = ae[r] * elu_grad_from_out(ev);  // d pre-activation of fc_embed_state_action
        }
      }
    }
    __syncthreads();
    // ---- d [state_t | action_t] = X4 @ W_sa
    dense_cols(X4, D, p.Wsa, X, X, wave, li, lh, [=](bool ok, int n, const f32x16& acc) {
      if (!ok) return;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < S) Gs[n * kLD + m] = acc[r];
        else SM[n * kLD + m] = acc[r];
      }
    });
    __syncthreads();
    // ---- tanh-Normal sample backward -> gradient at the actor trunk's output of step t
    for (int i = tid; i < kRows * A; i += blockDim.x) {
      const int row = i / A, a = i % A;
      if (row < nr) {
        const size_t o = (rb + row) * A + a;
        const float act = p.xsa[(rb + row) * X + S + a];
        const float du = SM[(S + a) * kLD + row] * (1.f - act * act);
        const float tm = p.a_mean[o] / p.a_mean_scale;
        p.d_araw[(rb + row) * 2 * A + a] = du * (1.f - tm * tm);
        p.d_araw[(rb + row) * 2 * A + A + a] = du * p.eps_act.at(o) * (-expm1f(-(p.a_std[o] - p.a_min_std)));
      }
    }
    __syncthreads();
  }
  if (p.dfeat0) {
    for (int i = tid; i < kRows * F; i += blockDim.x) {
      const int row = i / F, f = i % F;
      if (row < nr) p.dfeat0[(size_t)(r0 + row) * F + f] = Gb[f * kLD + row];
    }
  }
}

__global__ void transpose2_kernel(const float* __restrict__ src, int rows, int cols, int ld, float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int rr = blockIdx.y * 32, cc = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = rr + i, c = cc + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * ld + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = cc + i, r = rr + threadIdx.x;
    if (c < cols && r < rows) dst[(size_t)c * rows + r] = tile[threadIdx.x][i];
  }
}

static int tr(const float* src, int rows, int cols, float* dst, hipStream_t s) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(32, 8);
  hipLaunchKernelGGL(transpose2_kernel, grid, block, 0, s, src, rows, cols, cols, dst);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

bool imagine_fused_ok(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, int n_actor_layers) {
  return n_actor_layers == 5 && D <= 256 && Hd <= 256 && D % 2 == 0 && Hd % 2 == 0 && (S + A) % 2 == 0 &&
         (D + S) % 2 == 0 && (2 * S) % 2 == 0 && 2 * A <= 64 && 2 * S <= 64 && S + A <= 64 &&
         (Hm + 1) * N * 4 * D < kMaxIdx;
}

size_t imagine_fused_fwd_ws_floats(int64_t A, int64_t D, int64_t Hd, int64_t S) {
  return (size_t)((D + S) * Hd + 3 * Hd * Hd + Hd * 2 * A + (S + A) * D + 2 * D * 3 * D + D * Hd + Hd * 2 * S);
}

static size_t fwd_lds_bytes(int A, int D, int Hd, int S) {
  const int F = D + S, X = S + A, W = D > Hd ? D : Hd, SMr = 2 * A > 2 * S ? 2 * A : 2 * S;
  return (size_t)(2 * F + 2 * W + X + SMr) * kLD * sizeof(float);
}
static size_t bwd_lds_bytes(int A, int D, int Hd, int S) {
  const int X = S + A, W = D > Hd ? D : Hd, SMr = 2 * S > X ? 2 * S : X;
  return (size_t)(D + S + SMr + 4 * W) * kLD * sizeof(float);
}

int imagine_fused_fwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S,
                      const float* const* rp, const float* const* ap, const float* belief0, const float* state0,
                      NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std, float a_init_std,
                      float a_mean_scale, float* featx, float* prior_mean, float* prior_std, float* a_hidden,
                      int64_t a_layer_rows, float* a_raw, float* a_mean, float* a_std, float* xsa, float* e,
                      float* gates, float* hp, void* ws, hipStream_t stream) {
  const int F = (int)(D + S), X = (int)(S + A);
  float* w = (float*)ws;
  ImgFwdArgs a;
  a.d = ImgDims{(int)Hm, (int)N, (int)A, (int)D, (int)Hd, (int)S};
  int rc;
  const int kin[5] = {F, (int)Hd, (int)Hd, (int)Hd, (int)Hd};
  const int nout[5] = {(int)Hd, (int)Hd, (int)Hd, (int)Hd, (int)(2 * A)};
  for (int l = 0; l < 5; ++l) {
    if ((rc = tr(ap[2 * l], nout[l], kin[l], w, stream))) return rc;
    a.aWt[l] = w;
    a.ab[l] = ap[2 * l + 1];
    w += (size_t)nout[l] * kin[l];
  }
  if ((rc = tr(rp[0], (int)D, X, w, stream))) return rc;
  a.WsaT = w; w += (size_t)D * X;
  if ((rc = tr(rp[2], (int)(3 * D), (int)D, w, stream))) return rc;
  a.WihT = w; w += (size_t)3 * D * D;
  if ((rc = tr(rp[3], (int)(3 * D), (int)D, w, stream))) return rc;
  a.WhhT = w; w += (size_t)3 * D * D;
  if ((rc = tr(rp[6], (int)Hd, (int)D, w, stream))) return rc;
  a.WbpT = w; w += (size_t)Hd * D;
  if ((rc = tr(rp[8], (int)(2 * S), (int)Hd, w, stream))) return rc;
  a.WspT = w;
  a.bsa = rp[1]; a.bih = rp[4]; a.bhh = rp[5]; a.bbp = rp[7]; a.bsp = rp[9];
  a.belief0 = belief0; a.state0 = state0; a.eps_act = eps_act; a.eps_prior = eps_prior;
  a.min_std = min_std; a.a_min_std = a_min_std; a.a_init_std = a_init_std; a.a_mean_scale = a_mean_scale;
  a.featx = featx; a.prior_mean = prior_mean; a.prior_std = prior_std; a.a_hidden = a_hidden; a.a_raw = a_raw;
  a.a_mean = a_mean; a.a_std = a_std; a.xsa = xsa; a.e = e; a.gates = gates; a.hp = hp;
  a.a_layer_rows = (size_t)a_layer_rows;
  const size_t lds_b = fwd_lds_bytes((int)A, (int)D, (int)Hd, (int)S);
  hipError_t he = hipFuncSetAttribute((const void*)imagine_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_b);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL(imagine_fwd_kernel, dim3((unsigned)((N + kRows - 1) / kRows)), dim3(512), lds_b, stream, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

int imagine_fused_bwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, const float* const* rp,
                      NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std,
                      float a_mean_scale, const float* featx, const float* prior_std, const float* a_mean,
                      const float* a_std, const float* xsa, const float* e, const float* gates, const float* hp,
                      const float* dfeat, const float* dprior_mean, const float* dprior_std, float* d_araw,
                      float* dfeat0, hipStream_t stream) {
  ImgBwdArgs a;
  a.d = ImgDims{(int)Hm, (int)N, (int)A, (int)D, (int)Hd, (int)S};
  a.Wsa = rp[0]; a.Wih = rp[2]; a.Whh = rp[3]; a.Wbp = rp[6]; a.Wsp = rp[8];
  a.eps_act = eps_act; a.eps_prior = eps_prior;
  a.min_std = min_std; a.a_min_std = a_min_std; a.a_mean_scale = a_mean_scale;
  a.featx = featx; a.prior_std = prior_std; a.a_mean = a_mean; a.a_std = a_std; a.xsa = xsa; a.e = e;
  a.gates = gates; a.hp = hp; a.dfeat = dfeat; a.dprior_mean = dprior_mean; a.dprior_std = dprior_std;
  a.d_araw = d_araw; a.dfeat0 = dfeat0;
  const size_t lds_b = bwd_lds_bytes((int)A, (int)D, (int)Hd, (int)S);
  if (lds_b > 160 * 1024) return REPO_E_SHAPE;
  hipError_t he = hipFuncSetAttribute((const void*)imagine_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_b);
  if (he != hipSuccess) return (int)he;
  hipLaunchKernelGGL(imagine_bwd_kernel, dim3((unsigned)((N + kRows - 1) / kRows)), dim3(512), lds_b, stream, a);
  he = hipGetLastError();
  return he == hipSuccess ? REPO_OK : (int)he;
}

}  // namespace repo
