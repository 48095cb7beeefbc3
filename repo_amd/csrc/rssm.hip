// RSSM observe scan: fused per-timestep GRU + prior/posterior cell, persistent over T.
//
// Reference: TransitionModel.observe / compute_belief / compute_prior_state /
// compute_posterior_state, /root/reference/algorithms/repo/models/rssm.py:34-64,76-146.
//
// Structure (MI355X-first, not a per-op translation):
//  * The posterior's observation half, embed_t @ W_bq[:, D:]^T, does not depend on the
//    recurrence and is hoisted out of the scan as ONE (T*B, E) x (E, Hd) MFMA GEMM.
//  * The scan itself is a latency-bound chain of T dependent steps on B rows.  One
//    persistent workgroup owns R batch rows for all T steps; the deterministic belief and
//    the stochastic state stay in LDS between steps, and each thread owns one output
//    feature of the current stage, so weights stream through coalesced, transposed copies
//    ([k][feature]) that stay L2-resident, with the row vectors broadcast from LDS.
//  * Everything the backward needs is written once per step; weight gradients are NOT
//    formed inside the scan: the reverse scan emits per-step pre-activation deltas and the
//    weight/bias gradients become seven (T*B)-row MFMA GEMMs afterwards.
#include "common.h"
#include "scan_cs.h"

namespace repo {

constexpr int kMaxW = 256;   // max belief / hidden width (one thread per feature)
constexpr int kMaxS2 = 128;  // max 2*state
constexpr int kMaxX = 64;    // max state + action

struct ObsDims {
  int T, B, A, D, Hd, S;
};

// dst[c][r] = src[r*ld + c]   (rows x cols -> cols x rows)
__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, int ld, float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * ld + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < cols && r < rows) dst[(size_t)c * rows + r] = tile[threadIdx.x][i];
  }
}

static int transpose_to(const float* src, int rows, int cols, int ld, float* dst, hipStream_t s) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(32, 8);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, s, src, rows, cols, ld, dst);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// k4-interleaved weights for the scan's GEMV stages: dst[(kg*N + n)*4 + u] = W(n, k = 4kg + u), 0 for k >= K,
// with W(n, k) = src[n*sn + k*sk].  Thread n of a stage then reads FOUR consecutive k of its output with one
// coalesced 16-byte load (1 KB per wave) instead of four dword loads: the scan was bound by the CU's
// vector-memory issue rate (~5.5k wave-level dword loads per step, ~12 cycles each), not by L2 latency.
struct K4Job {
  const float* src;
  float* dst;
  int N, K, sn, sk;
};
constexpr int kMaxK4Jobs = 8;
struct K4Jobs {
  K4Job job[kMaxK4Jobs];
  int n;
};
// one launch for all the matrices of a scan: blockIdx.y = job
__global__ void pack_k4_kernel(K4Jobs js) {
  const K4Job j = js.job[blockIdx.y];
  const int total = ((j.K + 3) >> 2) * 4 * j.N;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int u = i & 3, q = i >> 2;
    const int n = q % j.N, kg = q / j.N;
    const int k = 4 * kg + u;
    j.dst[i] = k < j.K ? j.src[(size_t)n * j.sn + (size_t)k * j.sk] : 0.f;
  }
}

static size_t packed_k4_floats(int64_t N, int64_t K) { return (size_t)((K + 3) / 4) * 4 * N; }

// queue a pack; returns where the next pack may start
static float* pack_k4_add(K4Jobs& js, const float* src, int N, int K, int sn, int sk, float* dst) {
  js.job[js.n++] = K4Job{src, dst, N, K, sn, sk};
  return dst + packed_k4_floats(N, K);
}
static int pack_k4_launch(const K4Jobs& js, hipStream_t s) {
  int mx = 1;
  for (int i = 0; i < js.n; ++i) {
    const int blocks = (int)((packed_k4_floats(js.job[i].N, js.job[i].K) + 255) / 256);
    if (blocks > mx) mx = blocks;
  }
  if (mx > 512) mx = 512;
  hipLaunchKernelGGL(pack_k4_kernel, dim3(mx, js.n), dim3(256), 0, s, js);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

__device__ __forceinline__ float fma4(const float4 w, const float* __restrict__ x, float acc) {
  const float4 v = *reinterpret_cast<const float4*>(x);  // same address in every lane: LDS broadcast
  acc = fmaf(w.x, v.x, acc);
  acc = fmaf(w.y, v.y, acc);
  acc = fmaf(w.z, v.z, acc);
  return fmaf(w.w, v.w, acc);
}

struct ObsFwdArgs {
  ObsDims d;
  // k4-interleaved weights (pack_k4_kernel) and biases
  const float *WsaT, *bsa, *WihT, *WhhT, *bih, *bhh, *WbpT, *bbp, *WspT, *bsp, *WbqT, *bbq, *WsqT, *bsq;
  const float *prev_belief, *prev_state;  // (B,D), (B,S)
  const float *actions, *nonterms;        // (T,B,A), (T,B)
  const float* eemb;                      // (T,B,Hd) hoisted embed contribution (no bias)
  NoiseSrc eps_prior, eps_post;           // (T,B,S)
  float* featx;                           // (T+1,B,D+S): slot 0 = [prev_belief|prev_state], slot t+1 = [belief_t|post_t]
  float *prior_state, *prior_mean, *prior_std, *post_mean, *post_std;  // (T,B,S)
  float *xsa, *e, *gates, *hp, *hq;       // saved for backward: (T,B,S+A) (T,B,D) (T,B,4D) (T,B,Hd) (T,B,Hd)
  float min_std;
  int prior_only;  // the NEXT step is fed the prior sample (observe without observations, rssm.py:118)
  int skip_prior;  // the prior head is not evaluated here (repo_rssm_prior_head does it for all steps at once)
};

// R rows per workgroup, KQ-way split of every reduction (k) range over thread groups of 256:
// thread (kq, j) accumulates feature j over its quarter of k, partials meet in LDS.  The scan is
// bound by the latency of streaming ~1.4 MB of L2-resident weights per step through ONE CU; the
// k-split multiplies the loads in flight (memory-level parallelism), which is what that needs.
template <int R, int KQ>
__global__ __launch_bounds__(256 * KQ) void observe_fwd_kernel(ObsFwdArgs p) {
  const int T = p.d.T, B = p.d.B, A = p.d.A, D = p.d.D, Hd = p.d.Hd, S = p.d.S;
  const int X = S + A, F = D + S;
  __shared__ __attribute__((aligned(16))) float xs[R][kMaxX];
  __shared__ __attribute__((aligned(16))) float es[R][kMaxW];
  __shared__ __attribute__((aligned(16))) float hs[2][R][kMaxW];
  __shared__ __attribute__((aligned(16))) float hps[R][kMaxW];
  __shared__ __attribute__((aligned(16))) float hqs[R][kMaxW];
  __shared__ float outs[R][2 * kMaxS2];  // [0,2S) prior raw, [2S,4S) posterior raw
  __shared__ float st[R][kMaxS2];
  __shared__ float part[KQ][6][R][kMaxW];  // k-split partial sums

  const int tid = threadIdx.x;
  const int j = tid & 255;
  const int kq = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int b0 = blockIdx.x * R;
  int nr = B - b0;
  if (nr > R) nr = R;
  // this thread group's range of k GROUPS (4 consecutive k each) of a K-long reduction
  auto krange = [&](int K, int& g0, int& g1) {
    const int KG = (K + 3) >> 2, per = (KG + KQ - 1) / KQ;
    g0 = kq * per;
    g1 = min(KG, g0 + per);
  };
  // rows of the operand vectors beyond their width are read by the zero-padded last k group
  for (int i = tid; i < R * kMaxX; i += blockDim.x) (&xs[0][0])[i] = 0.f;
  for (int i = tid; i < R * kMaxW; i += blockDim.x) {
    (&es[0][0])[i] = 0.f;
    (&hs[0][0][0])[i] = 0.f;
    (&hs[1][0][0])[i] = 0.f;
    (&hps[0][0])[i] = 0.f;
    (&hqs[0][0])[i] = 0.f;
  }
  __syncthreads();

  // slot 0 of featx and the carried state
  for (int i = tid; i < R * D; i += blockDim.x) {
    const int r = i / D, c = i % D;
    const float v = r < nr ? p.prev_belief[(size_t)(b0 + r) * D + c] : 0.f;
    hs[0][r][c] = v;
    if (r < nr) p.featx[(size_t)(b0 + r) * F + c] = v;
  }
  for (int i = tid; i < R * S; i += blockDim.x) {
    const int r = i / S, c = i % S;
    const float v = r < nr ? p.prev_state[(size_t)(b0 + r) * S + c] : 0.f;
    st[r][c] = v;
    if (r < nr) p.featx[(size_t)(b0 + r) * F + D + c] = v;
  }
  __syncthreads();

  // Every step is a chain of ten dependent stages; a global load issued where its value is needed adds its
  // whole latency (an L2 / HBM round trip, longer when other kernels share the chip) to that chain.  So: the
  // biases live in registers for the whole scan, and the per-step operands (nonterminal / action of the x
  // vector, the hoisted embed contribution) are fetched one step AHEAD, while the previous step computes.
  const bool feat_thr = tid < D, hid_thr = tid < Hd, out_thr = tid < 4 * S;
  const float b_sa = feat_thr ? p.bsa[tid] : 0.f;
  float b_g[6];
#pragma unroll
  for (int g = 0; g < 6; ++g) b_g[g] = feat_thr ? (g < 3 ? p.bih[g * D + tid] : p.bhh[(g - 3) * D + tid]) : 0.f;
  const float b_bp = hid_thr ? p.bbp[tid] : 0.f, b_bq = hid_thr ? p.bbq[tid] : 0.f;
  const float b_out = out_thr ? (tid >= 2 * S ? p.bsq[tid - 2 * S] : p.bsp[tid]) : 0.f;
  // x-vector role of this thread (R * X <= blockDim): element (xr, xk)
  const int xr = tid / X, xk = tid % X;
  const bool x_thr = tid < R * X && xr < nr;
  auto load_x = [&](int t) __attribute__((always_inline)) {
    const size_t row = (size_t)t * B + b0 + xr;
    return x_thr ? (xk < S ? p.nonterms[row] : p.actions[row * A + (xk - S)]) : 0.f;
  };
  float em_next[R];
  auto load_em = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      em_next[r] = (hid_thr && r < nr) ? p.eemb[((size_t)t * B + b0 + r) * Hd + tid] : 0.f;
  };
  float x_next = T > 0 ? load_x(0) : 0.f;
  if (T > 0) load_em(0);

  int cur = 0;
  for (int t = 0; t < T; ++t) {
    const size_t row0 = (size_t)t * B + b0;  // flattened (t, b0)
    const float x_in = x_next;
    float em[R];
#pragma unroll
    for (int r = 0; r < R; ++r) em[r] = em_next[r];
    if (t + 1 < T) {
      x_next = load_x(t + 1);
      load_em(t + 1);
    }
    // ---- x = [state * nonterm, action]
    if (tid < R * X) {
      float v = 0.f;
      if (x_thr) {
        v = xk < S ? st[xr][xk] * x_in : x_in;
        p.xsa[(row0 + xr) * X + xk] = v;
      }
      xs[xr][xk] = v;
    }
    __syncthreads();
    // ---- e = elu(W_sa x + b)
    if (j < D) {
      float acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.f;
      int k0, k1;
      krange(X, k0, k1);
      const float4* W4 = reinterpret_cast<const float4*>(p.WsaT);
#pragma unroll 2
      for (int g = k0; g < k1; ++g) {
        const float4 w = W4[g * D + j];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fma4(w, &xs[r][4 * g], acc[r]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) part[kq][0][r][j] = acc[r];
    }
    __syncthreads();
    if (tid < D) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = b_sa;
#pragma unroll
        for (int q = 0; q < KQ; ++q) acc += part[q][0][r][tid];
        const float v = elu(acc);
        es[r][tid] = v;
        if (r < nr) p.e[(row0 + r) * D + tid] = v;
      }
    }
    __syncthreads();
    // ---- GRU cell (gate order r,z,n): partial sums over this thread's k range
    if (j < D) {
      float gi[3][R], gh[3][R];
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < R; ++r) gi[g][r] = gh[g][r] = 0.f;
      const float* hc = &hs[cur][0][0];
      int k0, k1;
      krange(D, k0, k1);
      const float4* Wi4 = reinterpret_cast<const float4*>(p.WihT);
      const float4* Wh4 = reinterpret_cast<const float4*>(p.WhhT);
#pragma unroll 2
      for (int g = k0; g < k1; ++g) {
        const float4* wi = Wi4 + (size_t)g * 3 * D + j;
        const float4* wh = Wh4 + (size_t)g * 3 * D + j;
        const float4 wi0 = wi[0], wi1 = wi[D], wi2 = wi[2 * D];
        const float4 wh0 = wh[0], wh1 = wh[D], wh2 = wh[2 * D];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float* ev = &es[r][4 * g];
          const float* hv = hc + r * kMaxW + 4 * g;
          gi[0][r] = fma4(wi0, ev, gi[0][r]);
          gi[1][r] = fma4(wi1, ev, gi[1][r]);
          gi[2][r] = fma4(wi2, ev, gi[2][r]);
          gh[0][r] = fma4(wh0, hv, gh[0][r]);
          gh[1][r] = fma4(wh1, hv, gh[1][r]);
          gh[2][r] = fma4(wh2, hv, gh[2][r]);
        }
      }
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          part[kq][g][r][j] = gi[g][r];
          part[kq][3 + g][r][j] = gh[g][r];
        }
    }
    __syncthreads();
    if (tid < D) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float g6[6];
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          float sacc = b_g[g];
#pragma unroll
          for (int q = 0; q < KQ; ++q) sacc += part[q][g][r][tid];
          g6[g] = sacc;
        }
        const float rg = sigmoidf(g6[0] + g6[3]);
        const float zg = sigmoidf(g6[1] + g6[4]);
        const float ng = tanh_fast(g6[2] + rg * g6[5]);
        const float hprev = hs[cur][r][tid];
        const float hn = (1.f - zg) * ng + zg * hprev;
        hs[cur ^ 1][r][tid] = hn;
        if (r < nr) {
          float* g = p.gates + (row0 + r) * 4 * D;
          g[tid] = rg;
          g[D + tid] = zg;
          g[2 * D + tid] = ng;
          g[3 * D + tid] = g6[5];
          p.featx[((size_t)(t + 1) * B + b0 + r) * F + tid] = hn;
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    // ---- hidden layers of the prior and the posterior heads
    if (j < Hd) {
      float ap[R], aq[R];
#pragma unroll
      for (int r = 0; r < R; ++r) ap[r] = aq[r] = 0.f;
      const float* hc = &hs[cur][0][0];
      int k0, k1;
      krange(D, k0, k1);
      const float4* Wp4 = reinterpret_cast<const float4*>(p.WbpT);
      const float4* Wq4 = reinterpret_cast<const float4*>(p.WbqT);
      if (p.skip_prior) {  // the prior head depends on belief_t only: off the recurrence, done for all steps afterwards
#pragma unroll 2
        for (int g = k0; g < k1; ++g) {
          const float4 wq = Wq4[(size_t)g * Hd + j];
#pragma unroll
          for (int r = 0; r < R; ++r) aq[r] = fma4(wq, hc + r * kMaxW + 4 * g, aq[r]);
        }
      } else {
#pragma unroll 2
        for (int g = k0; g < k1; ++g) {
          const float4 wp = Wp4[(size_t)g * Hd + j];
          const float4 wq = Wq4[(size_t)g * Hd + j];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float* hv = hc + r * kMaxW + 4 * g;
            ap[r] = fma4(wp, hv, ap[r]);
            aq[r] = fma4(wq, hv, aq[r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        part[kq][0][r][j] = ap[r];
        part[kq][1][r][j] = aq[r];
      }
    }
    __syncthreads();
    if (tid < Hd) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float ap = b_bp, aq = b_bq + em[r];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          ap += part[q][0][r][tid];
          aq += part[q][1][r][tid];
        }
        const float vp = elu(ap), vq = elu(aq);
        hps[r][tid] = vp;
        hqs[r][tid] = vq;
        if (r < nr) {
          if (!p.skip_prior) p.hp[(row0 + r) * Hd + tid] = vp;
          p.hq[(row0 + r) * Hd + tid] = vq;
        }
      }
    }
    __syncthreads();
    // ---- output layers: columns [0,2S) prior, [2S,4S) posterior; k split as above
    if (j < 4 * S && (j >= 2 * S || !p.skip_prior)) {
      const bool post = j >= 2 * S;
      const int o = post ? j - 2 * S : j;
      const float* Wt = post ? p.WsqT : p.WspT;
      const float* hsrc = post ? &hqs[0][0] : &hps[0][0];
      float acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.f;
      int k0, k1;
      krange(Hd, k0, k1);
      const float4* W4 = reinterpret_cast<const float4*>(Wt);
#pragma unroll 2
      for (int g = k0; g < k1; ++g) {
        const float4 w = W4[(size_t)g * 2 * S + o];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fma4(w, hsrc + r * kMaxW + 4 * g, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) part[kq][0][r][j] = acc[r];
    }
    __syncthreads();
    if (tid < 4 * S) {
      const bool post = tid >= 2 * S;
      const int o = post ? tid - 2 * S : tid;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = b_out;
#pragma unroll
        for (int q = 0; q < KQ; ++q) acc += part[q][0][r][tid];
        outs[r][tid] = acc;
      }
    }
    __syncthreads();
    // ---- softplus + reparameterised samples
    for (int i = tid; i < R * 2 * S; i += blockDim.x) {
      const int r = i / (2 * S), q = i % (2 * S);
      const bool post = q >= S;
      const int s = post ? q - S : q;
      const int base = post ? 2 * S : 0;
      const float mean = outs[r][base + s];
      const float sd = softplus(outs[r][base + S + s]) + p.min_std;
      if (r < nr && (post || !p.skip_prior)) {
        const size_t o = (row0 + r) * S + s;
        const float eps = post ? p.eps_post.at(o) : p.eps_prior.at(o);
        const float smp = fmaf(sd, eps, mean);
        if (post) {
          p.post_mean[o] = mean;
          p.post_std[o] = sd;
          if (!p.prior_only) {
            p.featx[((size_t)(t + 1) * B + b0 + r) * F + D + s] = smp;
            st[r][s] = smp;
          }
        } else {
          p.prior_mean[o] = mean;
          p.prior_std[o] = sd;
          p.prior_state[o] = smp;
          if (p.prior_only) {
            p.featx[((size_t)(t + 1) * B + b0 + r) * F + D + s] = smp;
            st[r][s] = smp;
          }
        }
      }
    }
    __syncthreads();
  }
}

struct ObsBwdArgs {
  ObsDims d;
  // k4-interleaved weights (pack_k4_kernel) with the reduction over the layer's OUTPUT index:
  // W(n = input index, k = output index) = native[k*ld + n]
  const float *Wsa, *Wih, *Whh, *Wbp, *Wsp, *Wbq, *Wsq;
  // saved by forward
  const float *featx, *nonterms, *e, *gates, *hp, *hq, *prior_std, *post_std;
  NoiseSrc eps_prior, eps_post;
  // upstream gradients, each nullable: d featx[1:] (T,B,D+S); prior_state, and the four (mean,std) (T,B,S)
  const float *dfeat, *dprior_state, *dpm, *dps, *dqm, *dqs;
  // per-step deltas (outputs)
  float *doutp, *doutq;  // (T,B,2S)
  float *dhp, *dhq;      // (T,B,Hd)   d pre-activation of the hidden layers
  float *dgi, *dgh;      // (T,B,3D)
  float* de;             // (T,B,D)    d pre-activation of fc_embed_state_action
  float *dprev_belief, *dprev_state;  // (B,D) (B,S), nullable
  float min_std;
};

template <int R, int KQ>
__global__ __launch_bounds__(256 * KQ) void observe_bwd_kernel(ObsBwdArgs p) {
  const int T = p.d.T, B = p.d.B, A = p.d.A, D = p.d.D, Hd = p.d.Hd, S = p.d.S;
  const int X = S + A, F = D + S;
  __shared__ float dh[R][kMaxW];      // carried d belief
  __shared__ float dst[R][kMaxS2];    // carried d posterior state
  __shared__ float dbel[R][kMaxW];
  __shared__ __attribute__((aligned(16))) float douts[R][2 * kMaxS2];
  __shared__ __attribute__((aligned(16))) float dhps[R][kMaxW];
  __shared__ __attribute__((aligned(16))) float dhqs[R][kMaxW];
  __shared__ __attribute__((aligned(16))) float dgis[R][3 * kMaxW];
  __shared__ __attribute__((aligned(16))) float dghs[R][3 * kMaxW];
  __shared__ __attribute__((aligned(16))) float des[R][kMaxW];
  __shared__ float part[KQ][2][R][kMaxW];  // k-split partial sums

  const int tid = threadIdx.x;
  const int j = tid & 255;
  const int kq = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int b0 = blockIdx.x * R;
  int nr = B - b0;
  if (nr > R) nr = R;
  // this thread group's range of k GROUPS (4 consecutive k each) of a K-long reduction
  auto krange = [&](int K, int& g0, int& g1) {
    const int KG = (K + 3) >> 2, per = (KG + KQ - 1) / KQ;
    g0 = kq * per;
    g1 = min(KG, g0 + per);
  };
  for (int i = tid; i < R * kMaxW; i += blockDim.x) {
    (&dhps[0][0])[i] = 0.f;
    (&dhqs[0][0])[i] = 0.f;
    (&des[0][0])[i] = 0.f;
  }
  for (int i = tid; i < R * 3 * kMaxW; i += blockDim.x) {
    (&dgis[0][0])[i] = 0.f;
    (&dghs[0][0])[i] = 0.f;
  }
  for (int i = tid; i < R * 2 * kMaxS2; i += blockDim.x) (&douts[0][0])[i] = 0.f;
  for (int i = tid; i < R * kMaxW; i += blockDim.x) (&dh[0][0])[i] = 0.f;
  for (int i = tid; i < R * kMaxS2; i += blockDim.x) (&dst[0][0])[i] = 0.f;
  __syncthreads();

  // The saved activations and upstream gradients a step reads (each a dependent global load in front of one of
  // its eight stages) are fetched one step AHEAD into registers: in the update this scan shares the chip with the
  // decoder's backward, and a load issued where its value is needed then costs 1-2 us of the chain.
  struct StepIn {
    float dfb[R], g_r[R], g_z[R], g_n[R], g_hn[R], hprev[R], ev[R];  // feature threads (tid < D)
    float hp[R], hq[R];                                              // hidden threads (tid < Hd)
    float o_dsmp, o_dm, o_dsd, o_sd;                                 // output-delta role (tid < R * 2S)
    float nt[R];                                                     // tid < S
  };
  const int orr = tid / (2 * S), oq = tid % (2 * S);
  const bool o_thr = tid < R * 2 * S && orr < nr, o_post = oq >= S;
  const int o_s = o_post ? oq - S : oq;
  // one register set: each group of fields is re-loaded for step t-1 right after step t's last use of it
  StepIn in;
  auto load_top = [&](int t) __attribute__((always_inline)) {  // consumed by the first stage
    const size_t row0 = (size_t)t * B + b0;
#pragma unroll
    for (int r = 0; r < R; ++r) in.dfb[r] = (tid < D && r < nr && p.dfeat) ? p.dfeat[(row0 + r) * F + tid] : 0.f;
    in.o_dsmp = in.o_dm = in.o_dsd = in.o_sd = 0.f;
    if (o_thr) {
      const size_t o = (row0 + orr) * S + o_s;
      if (o_post) {
        in.o_dsmp = p.dfeat ? p.dfeat[(row0 + orr) * F + D + o_s] : 0.f;
        in.o_dm = p.dqm ? p.dqm[o] : 0.f;
        in.o_dsd = p.dqs ? p.dqs[o] : 0.f;
        in.o_sd = p.post_std[o];
      } else {
        in.o_dsmp = p.dprior_state ? p.dprior_state[o] : 0.f;
        in.o_dm = p.dpm ? p.dpm[o] : 0.f;
        in.o_dsd = p.dps ? p.dps[o] : 0.f;
        in.o_sd = p.prior_std[o];
      }
    }
  };
  auto load_hid = [&](int t) __attribute__((always_inline)) {
    const size_t row0 = (size_t)t * B + b0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ha = tid < Hd && r < nr;
      in.hp[r] = ha ? p.hp[(row0 + r) * Hd + tid] : 0.f;
      in.hq[r] = ha ? p.hq[(row0 + r) * Hd + tid] : 0.f;
    }
  };
  auto load_gru = [&](int t) __attribute__((always_inline)) {
    const size_t row0 = (size_t)t * B + b0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool fa = tid < D && r < nr;
      const float* g = p.gates + (row0 + r) * 4 * D;
      in.g_r[r] = fa ? g[tid] : 0.f;
      in.g_z[r] = fa ? g[D + tid] : 0.f;
      in.g_n[r] = fa ? g[2 * D + tid] : 0.f;
      in.g_hn[r] = fa ? g[3 * D + tid] : 0.f;
      in.hprev[r] = fa ? p.featx[((size_t)t * B + b0 + r) * F + tid] : 0.f;
    }
  };
  auto load_e = [&](int t) __attribute__((always_inline)) {
    const size_t row0 = (size_t)t * B + b0;
#pragma unroll
    for (int r = 0; r < R; ++r) in.ev[r] = (tid < D && r < nr) ? p.e[(row0 + r) * D + tid] : 0.f;
  };
  auto load_nt = [&](int t) __attribute__((always_inline)) {
    const size_t row0 = (size_t)t * B + b0;
#pragma unroll
    for (int r = 0; r < R; ++r) in.nt[r] = (tid < S && r < nr) ? p.nonterms[row0 + r] : 0.f;
  };
  if (T > 0) {
    load_top(T - 1);
    load_hid(T - 1);
    load_gru(T - 1);
    load_e(T - 1);
    load_nt(T - 1);
  }

  for (int t = T - 1; t >= 0; --t) {
    const size_t row0 = (size_t)t * B + b0;
    const int tn = t > 0 ? t - 1 : 0;  // the step whose operands are fetched behind each stage (t = 0: a harmless re-read)
    // ---- total gradient on belief_t; heads' output-layer deltas
    if (tid < D) {
#pragma unroll
      for (int r = 0; r < R; ++r) dbel[r][tid] = dh[r][tid] + in.dfb[r];
    }
    if (tid < R * 2 * S) {
      const int r = orr, s = o_s;
      const bool post = o_post;
      float dm = 0.f, draw = 0.f;
      if (r < nr) {
        const size_t o = (row0 + r) * S + s;
        float dsmp = in.o_dsmp + (post ? dst[r][s] : 0.f), dsd = in.o_dsd;
        const float sd = in.o_sd;
        const float eps = post ? p.eps_post.at(o) : p.eps_prior.at(o);
        dm = in.o_dm;
        dm += dsmp;
        dsd = fmaf(dsmp, eps, dsd);
        // d softplus(raw)/d raw = sigmoid(raw) = 1 - exp(-softplus(raw))
        draw = dsd * (-expm1f(-(sd - p.min_std)));
        float* dst_out = post ? p.doutq : p.doutp;
        dst_out[(row0 + r) * 2 * S + s] = dm;
        dst_out[(row0 + r) * 2 * S + S + s] = draw;
      }
      const int base = post ? 2 * S : 0;
      douts[r][base + s] = dm;
      douts[r][base + S + s] = draw;
    }
    load_top(tn);
    __syncthreads();
    // ---- back through the output layers to the hidden pre-activations
    if (j < Hd) {
      float ap[R], aq[R];
#pragma unroll
      for (int r = 0; r < R; ++r) ap[r] = aq[r] = 0.f;
      int k0, k1;
      krange(2 * S, k0, k1);
      const float4* Wp4 = reinterpret_cast<const float4*>(p.Wsp);
      const float4* Wq4 = reinterpret_cast<const float4*>(p.Wsq);
#pragma unroll 2
      for (int g = k0; g < k1; ++g) {
        const float4 wp = Wp4[(size_t)g * Hd + j];
        const float4 wq = Wq4[(size_t)g * Hd + j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          ap[r] = fma4(wp, &douts[r][4 * g], ap[r]);
          aq[r] = fma4(wq, &douts[r][2 * S + 4 * g], aq[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        part[kq][0][r][j] = ap[r];
        part[kq][1][r][j] = aq[r];
      }
    }
    __syncthreads();
    if (tid < Hd) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float ap = 0.f, aq = 0.f;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          ap += part[q][0][r][tid];
          aq += part[q][1][r][tid];
        }
        float vp = 0.f, vq = 0.f;
        if (r < nr) {
          vp = ap * elu_grad_from_out(in.hp[r]);
          vq = aq * elu_grad_from_out(in.hq[r]);
          p.dhp[(row0 + r) * Hd + tid] = vp;
          p.dhq[(row0 + r) * Hd + tid] = vq;
        }
        dhps[r][tid] = vp;
        dhqs[r][tid] = vq;
      }
    }
    load_hid(tn);
    __syncthreads();
    // ---- into belief_t (k-split partial sums over the hidden index)
    if (j < D) {
      float acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.f;
      int k0, k1;
      krange(Hd, k0, k1);
      const float4* Wp4 = reinterpret_cast<const float4*>(p.Wbp);
      const float4* Wq4 = reinterpret_cast<const float4*>(p.Wbq);
#pragma unroll 2
      for (int g = k0; g < k1; ++g) {
        const float4 wp = Wp4[(size_t)g * D + j];
        const float4 wq = Wq4[(size_t)g * D + j];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fma4(wp, &dhps[r][4 * g], fma4(wq, &dhqs[r][4 * g], acc[r]));
      }
#pragma unroll
      for (int r = 0; r < R; ++r) part[kq][0][r][j] = acc[r];
    }
    __syncthreads();
    // ---- through the GRU gates (pointwise in the feature index)
    if (tid < D) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float db_ = dbel[r][tid];
#pragma unroll
        for (int q = 0; q < KQ; ++q) db_ += part[q][0][r][tid];
        float g_r = 0.f, g_z = 0.f, g_n = 0.f, g_hn = 0.f, dhprev = 0.f;
        if (r < nr) {
          const float rg = in.g_r[r], zg = in.g_z[r], ng = in.g_n[r], ghn = in.g_hn[r];
          const float hprev = in.hprev[r];
          const float dn = db_ * (1.f - zg);
          const float dz = db_ * (hprev - ng);
          dhprev = db_ * zg;
          g_n = dn * (1.f - ng * ng);
          g_hn = g_n * rg;
          g_r = g_n * ghn * rg * (1.f - rg);
          g_z = dz * zg * (1.f - zg);
          float* gi = p.dgi + (row0 + r) * 3 * D;
          float* gh = p.dgh + (row0 + r) * 3 * D;
          gi[tid] = g_r;
          gi[D + tid] = g_z;
          gi[2 * D + tid] = g_n;
          gh[tid] = g_r;
          gh[D + tid] = g_z;
          gh[2 * D + tid] = g_hn;
        }
        dgis[r][tid] = g_r;
        dgis[r][D + tid] = g_z;
        dgis[r][2 * D + tid] = g_n;
        dghs[r][tid] = g_r;
        dghs[r][D + tid] = g_z;
        dghs[r][2 * D + tid] = g_hn;
        dh[r][tid] = dhprev;
      }
    }
    load_gru(tn);
    __syncthreads();
    // ---- through W_hh into belief_{t-1}, through W_ih into e (k-split over the 3D gate index)
    if (j < D) {
      float ah[R], ae[R];
#pragma unroll
      for (int r = 0; r < R; ++r) ah[r] = ae[r] = 0.f;
      int k0, k1;
      krange(3 * D, k0, k1);
      const float4* Wh4 = reinterpret_cast<const float4*>(p.Whh);
      const float4* Wi4 = reinterpret_cast<const float4*>(p.Wih);
#pragma unroll 2
      for (int g = k0; g < k1; ++g) {
        const float4 wh = Wh4[(size_t)g * D + j];
        const float4 wi = Wi4[(size_t)g * D + j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          ah[r] = fma4(wh, &dghs[r][4 * g], ah[r]);
          ae[r] = fma4(wi, &dgis[r][4 * g], ae[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        part[kq][0][r][j] = ah[r];
        part[kq][1][r][j] = ae[r];
      }
    }
    __syncthreads();
    if (tid < D) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float ah = dh[r][tid], ae = 0.f;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          ah += part[q][0][r][tid];
          ae += part[q][1][r][tid];
        }
        dh[r][tid] = ah;
        float v = 0.f;
        if (r < nr) {
          v = ae * elu_grad_from_out(in.ev[r]);
          p.de[(row0 + r) * D + tid] = v;
        }
        des[r][tid] = v;
      }
    }
    load_e(tn);
    __syncthreads();
    // ---- through W_sa into the previous posterior state (masked by nonterminal)
    if (j < S) {
      float acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.f;
      int k0, k1;
      krange(D, k0, k1);
      const float4* W4 = reinterpret_cast<const float4*>(p.Wsa);
#pragma unroll 2
      for (int g = k0; g < k1; ++g) {
        const float4 w = W4[(size_t)g * S + j];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fma4(w, &des[r][4 * g], acc[r]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) part[kq][0][r][j] = acc[r];
    }
    __syncthreads();
    if (tid < S) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < KQ; ++q) acc += part[q][0][r][tid];
        dst[r][tid] = r < nr ? acc * in.nt[r] : 0.f;
      }
    }
    load_nt(tn);
    __syncthreads();
  }
  if (p.dprev_belief)
    for (int i = tid; i < nr * D; i += blockDim.x) p.dprev_belief[(size_t)(b0 + i / D) * D + i % D] = dh[i / D][i % D];
  if (p.dprev_state)
    for (int i = tid; i < nr * S; i += blockDim.x) p.dprev_state[(size_t)(b0 + i / S) * S + i % S] = dst[i / S][i % S];
}

static bool dims_ok(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S) {
  return T >= 0 && B > 0 && A >= 0 && D > 0 && Hd > 0 && S > 0 && D <= kMaxW && Hd <= kMaxW && 2 * S <= kMaxS2 &&
         4 * S <= 256 && S + A <= kMaxX && T * B * 4 * D < kMaxIdx;
}

static size_t bwd_pack_floats(int64_t A, int64_t D, int64_t Hd, int64_t S) {
  (void)A;
  return 2 * packed_k4_floats(Hd, 2 * S) + 2 * packed_k4_floats(D, Hd) + 2 * packed_k4_floats(D, 3 * D) +
         packed_k4_floats(S, D);
}

static size_t fwd_ws_floats(int64_t A, int64_t D, int64_t Hd, int64_t S) {
  return packed_k4_floats(D, S + A) + 2 * packed_k4_floats(3 * D, D) + 2 * packed_k4_floats(Hd, D) +
         2 * packed_k4_floats(2 * S, Hd);
}

}  // namespace repo

using namespace repo;

extern "C" size_t repo_rssm_observe_fwd_workspace_bytes(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd,
                                                        int64_t S, int64_t E) {
  (void)E;
  size_t f = fwd_ws_floats(A, D, Hd, S);
  if (scan_cs_ok(T, B, A, D, Hd, S)) f = std::max(f, scan_cs_fwd_ws_floats(B, A, D, Hd, S));  // prior_only = 3
  return f * sizeof(float);
}

extern "C" int repo_rssm_observe_fwd(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S, int64_t E,
                                     const float* const* params, const float* prev_belief, const float* prev_state,
                                     const float* actions, const float* nonterms, const float* embeds,
                                     const float* eps_prior, const float* eps_post, uint64_t noise_seed,
                                     uint64_t noise_offset, float min_std, float* featx,
                                     float* prior_state, float* prior_mean, float* prior_std, float* post_mean,
                                     float* post_std, float* xsa, float* e, float* gates, float* hp, float* hq,
                                     float* eemb, int prior_only, unsigned* status, void* ws, size_t ws_bytes,
                                     hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(dims_ok(T, B, A, D, Hd, S) && E > 0, REPO_E_SHAPE);
  REPO_REQUIRE(params && prev_belief && prev_state && actions && nonterms && embeds && !eps_prior == !eps_post,
               REPO_E_BADARG);
  REPO_REQUIRE(featx && prior_state && prior_mean && prior_std && post_mean && post_std && xsa && e && gates && hp &&
                   hq && eemb,
               REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= fwd_ws_floats(A, D, Hd, S) * sizeof(float), REPO_E_WS_TOO_SMALL);
  const float* const* P = params;
  if (prior_only == 3) {
    // the column-split, weight-stationary engine (scan_cs.hip): prior head left out as with prior_only = 2
    REPO_REQUIRE(scan_cs_ok(T, B, A, D, Hd, S), REPO_E_SHAPE);
    if (T == 0) return REPO_OK;
    int rc3 = repo_gemm(0, 1, T * B, Hd, E, embeds, E, P[10] + D, D + E, nullptr, 1, eemb, Hd, REPO_EPI_NONE, nullptr,
                        0, 0, stream);
    if (rc3) return rc3;
    ScanCsFwd q{T, B, A, D, Hd, S, E, params, prev_belief, prev_state, actions, nonterms, eemb,
                NoiseSrc{eps_post, noise_seed, noise_offset + (uint64_t)(T * B * S)}, min_std,
                featx, post_mean, post_std, xsa, e, gates, hq, status};
    return scan_cs_fwd(q, ws, ws_bytes, stream);
  }
  float* w = (float*)ws;
  // W(n, k) = P[n*ld + k] (native (out, in) layout) -> k4-interleaved [k/4][n][4]
  const int X = (int)(S + A), d = (int)D, h = (int)Hd, s2 = (int)(2 * S);
  float* WsaT = w;  w += packed_k4_floats(d, X);
  float* WihT = w;  w += packed_k4_floats(3 * d, d);
  float* WhhT = w;  w += packed_k4_floats(3 * d, d);
  float* WbpT = w;  w += packed_k4_floats(h, d);
  float* WbqT = w;  w += packed_k4_floats(h, d);
  float* WspT = w;  w += packed_k4_floats(s2, h);
  float* WsqT = w;
  int rc;
  K4Jobs packs;
  packs.n = 0;
  pack_k4_add(packs, P[0], d, X, X, 1, WsaT);
  pack_k4_add(packs, P[2], 3 * d, d, d, 1, WihT);
  pack_k4_add(packs, P[3], 3 * d, d, d, 1, WhhT);
  pack_k4_add(packs, P[6], h, d, d, 1, WbpT);
  pack_k4_add(packs, P[10], h, d, (int)(D + E), 1, WbqT);
  pack_k4_add(packs, P[8], s2, h, h, 1, WspT);
  pack_k4_add(packs, P[12], s2, h, h, 1, WsqT);
  if ((rc = pack_k4_launch(packs, stream))) return rc;
  if (T == 0) return REPO_OK;
  // hoisted: eemb = embeds @ W_bq[:, D:]^T
  if ((rc = repo_gemm(0, 1, T * B, Hd, E, embeds, E, P[10] + D, D + E, nullptr, 1, eemb, Hd, REPO_EPI_NONE, nullptr, 0,
                      0, stream)))
    return rc;
  ObsFwdArgs a;
  a.d = ObsDims{(int)T, (int)B, (int)A, (int)D, (int)Hd, (int)S};
  a.WsaT = WsaT; a.bsa = P[1]; a.WihT = WihT; a.WhhT = WhhT; a.bih = P[4]; a.bhh = P[5];
  a.WbpT = WbpT; a.bbp = P[7]; a.WspT = WspT; a.bsp = P[9]; a.WbqT = WbqT; a.bbq = P[11]; a.WsqT = WsqT; a.bsq = P[13];
  a.prev_belief = prev_belief; a.prev_state = prev_state; a.actions = actions; a.nonterms = nonterms;
  a.eemb = eemb;
  a.prior_only = prior_only == 1;
  a.skip_prior = prior_only == 2;
  a.eps_prior = NoiseSrc{eps_prior, noise_seed, noise_offset};
  a.eps_post = NoiseSrc{eps_post, noise_seed, noise_offset + (uint64_t)(T * B * S)};
  a.featx = featx; a.prior_state = prior_state; a.prior_mean = prior_mean; a.prior_std = prior_std;
  a.post_mean = post_mean; a.post_std = post_std; a.xsa = xsa; a.e = e; a.gates = gates; a.hp = hp; a.hq = hq;
  a.min_std = min_std;
  // rows per workgroup: spread B over as many CUs as possible (the scan is latency-bound)
  if (B >= 512) {
    hipLaunchKernelGGL((observe_fwd_kernel<4, 1>), dim3(cdiv(B, 4)), dim3(256), 0, stream, a);
  } else if (B >= 128) {
    hipLaunchKernelGGL((observe_fwd_kernel<2, 2>), dim3(cdiv(B, 2)), dim3(512), 0, stream, a);
  } else if (B >= 32) {
    // two rows per workgroup: every streamed weight feeds two FMAs, which halves the scan's L2 traffic.
    // Alone the scan is no faster, but beside the convolution kernels it (and they) lose less to L2
    // contention: 11.1 -> 10.7 ms per pipelined update at B=50 (3 or 4 rows per workgroup: slower)
    hipLaunchKernelGGL((observe_fwd_kernel<2, 4>), dim3(cdiv(B, 2)), dim3(1024), 0, stream, a);
  } else {
    hipLaunchKernelGGL((observe_fwd_kernel<1, 4>), dim3((unsigned)B), dim3(1024), 0, stream, a);
  }
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

// ---- the prior head of all T steps at once (repo_rssm_observe_fwd with prior_only = 2 leaves it out of the scan)
__global__ void prior_sample_kernel(int n, int S, const float* __restrict__ outp, NoiseSrc eps, float min_std,
                                    float* __restrict__ mean_o, float* __restrict__ std_o, float* __restrict__ state_o) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int row = i / S, s = i % S;
    const float mean = outp[(size_t)row * 2 * S + s];
    const float sd = softplus(outp[(size_t)row * 2 * S + S + s]) + min_std;
    mean_o[i] = mean;
    std_o[i] = sd;
    state_o[i] = fmaf(sd, eps.at(i), mean);
  }
}

extern "C" size_t repo_rssm_prior_head_workspace_bytes(int64_t T, int64_t B, int64_t S) {
  return (size_t)(T * B * 2 * S) * sizeof(float);
}

extern "C" int repo_rssm_prior_head(int64_t T, int64_t B, int64_t D, int64_t Hd, int64_t S, const float* const* params,
                                    const float* featx, const float* eps_prior, uint64_t noise_seed,
                                    uint64_t noise_offset, float min_std, float* hp, float* prior_state,
                                    float* prior_mean, float* prior_std, void* ws, size_t ws_bytes,
                                    hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(T >= 0 && B > 0 && D > 0 && Hd > 0 && S > 0 && T * B * (int64_t)(D + S) < kMaxBufElems, REPO_E_SHAPE);
  REPO_REQUIRE(params && featx && hp && prior_state && prior_mean && prior_std, REPO_E_BADARG);
  if (T == 0) return REPO_OK;
  REPO_REQUIRE(ws && ws_bytes >= repo_rssm_prior_head_workspace_bytes(T, B, S), REPO_E_WS_TOO_SMALL);
  const int64_t rows = T * B, F = D + S;
  const float* bel = featx + (size_t)B * F;  // belief_t = featx[t + 1][:, :D]
  float* outp = (float*)ws;
  int rc;
  // hp = elu(belief @ W_bp^T + b);  out = hp @ W_sp^T + b   (fc_embed_belief_prior, fc_state_prior: rssm.py:42-50)
  if ((rc = repo_gemm(0, 1, rows, Hd, D, bel, F, params[6], D, params[7], 1, hp, Hd, REPO_EPI_ELU, nullptr, 0, 0, stream)))
    return rc;
  if ((rc = repo_gemm(0, 1, rows, 2 * S, Hd, hp, Hd, params[8], Hd, params[9], 1, outp, 2 * S, REPO_EPI_NONE, nullptr,
                      0, 0, stream)))
    return rc;
  const int n = (int)(rows * S);
  hipLaunchKernelGGL(prior_sample_kernel, dim3(cdiv(n, 256) < 1024 ? cdiv(n, 256) : 1024), dim3(256), 0, stream, n, (int)S,
                     (const float*)outp, NoiseSrc{eps_prior, noise_seed, noise_offset}, min_std, prior_mean, prior_std,
                     prior_state);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

// The deferred weight / bias gradients of the scan as (T*B)-row GEMM jobs (G = dparams in plist order).
static void obs_wgrad_jobs(WgradDesc* j, int64_t R_, int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S, int64_t E,
                           const float* featx, const float* xsa, const float* e, const float* hp, const float* hq,
                           const float* doutp, const float* doutq, const float* dhp, const float* dhq,
                           const float* dgi, const float* dgh, const float* de, float* const* G) {
  const int64_t F = D + S, X = S + A;
  auto g = [&](int i) -> float* { return G ? G[i] : nullptr; };
  const float* bel = featx ? featx + (size_t)B * F : nullptr;  // belief_t = featx[t+1][:, :D]
  // fc_state_prior / fc_state_posterior
  j[0] = WgradDesc{R_, 2 * S, Hd, doutp, 2 * S, hp, Hd, g(8), Hd, g(9)};
  j[1] = WgradDesc{R_, 2 * S, Hd, doutq, 2 * S, hq, Hd, g(12), Hd, g(13)};
  // fc_embed_belief_prior; fc_embed_belief_posterior's belief columns [0, D) of its (Hd, D + E) weight
  j[2] = WgradDesc{R_, Hd, D, dhp, Hd, bel, F, g(6), D, g(7)};
  j[3] = WgradDesc{R_, Hd, D, dhq, Hd, bel, F, g(10), D + E, g(11)};
  // GRU: weight_ih sees e, weight_hh sees belief_{t-1} = featx[t][:, :D]
  j[4] = WgradDesc{R_, 3 * D, D, dgi, 3 * D, e, D, g(2), D, g(4)};
  j[5] = WgradDesc{R_, 3 * D, D, dgh, 3 * D, featx, F, g(3), D, g(5)};
  // fc_embed_state_action
  j[6] = WgradDesc{R_, D, X, de, D, xsa, X, g(0), X, g(1)};
}

// output deltas of the PRIOR head for all steps (they depend on upstream gradients only, not on the recurrence)
__global__ void prior_delta_kernel(int n, int S, const float* __restrict__ dstate, const float* __restrict__ dpm,
                                   const float* __restrict__ dps, const float* __restrict__ prior_std, NoiseSrc eps,
                                   float min_std, float* __restrict__ doutp) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int row = i / S, s = i % S;
    const float dsmp = dstate ? dstate[i] : 0.f;
    const float dm = (dpm ? dpm[i] : 0.f) + dsmp;
    const float dsd = fmaf(dsmp, eps.at(i), dps ? dps[i] : 0.f);
    doutp[(size_t)row * 2 * S + s] = dm;
    doutp[(size_t)row * 2 * S + S + s] = dsd * (-expm1f(-(prior_std[i] - min_std)));
  }
}

extern "C" size_t repo_rssm_observe_bwd_workspace_bytes(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd,
                                                        int64_t S, int64_t E) {
  // deltas + the largest wgrad slab
  const size_t rows = (size_t)T * B;
  size_t deltas = rows * (size_t)(4 * S + 2 * Hd + 6 * D + D);
  // the seven recurrent-path weight gradients share one launch pair (one slab each); the (Hd x E) one runs alone
  WgradDesc jobs[7];
  obs_wgrad_jobs(jobs, T * B, B, A, D, Hd, S, E, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                 nullptr, nullptr, nullptr, nullptr, nullptr);
  size_t slab = gemm_wgrad_group_ws_bytes(jobs, 7);
  const size_t big = repo_gemm_wgrad_workspace_bytes(T * B, Hd, E);
  if (big > slab) slab = big;
  const size_t packs = bwd_pack_floats(A, D, Hd, S) * sizeof(float);  // live only during the scan kernel
  if (packs > slab) slab = packs;
  if (scan_cs_ok(T, B, A, D, Hd, S)) {  // column-split engine: its packs and exchange buffers + the prior head's d belief
    const size_t cs = (scan_cs_bwd_ws_floats(B, A, D, Hd, S) + rows * (size_t)D) * sizeof(float) + 256;
    if (cs > slab) slab = cs;
  }
  return deltas * sizeof(float) + slab + 256;
}

extern "C" int repo_rssm_observe_bwd(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S, int64_t E,
                                     const float* const* params, const float* nonterms, const float* embeds,
                                     const float* eps_prior, const float* eps_post, uint64_t noise_seed,
                                     uint64_t noise_offset, float min_std, const float* featx,
                                     const float* prior_std, const float* post_std, const float* xsa, const float* e,
                                     const float* gates, const float* hp, const float* hq, const float* dfeat,
                                     const float* dprior_state, const float* dpm, const float* dps, const float* dqm,
                                     const float* dqs, float* const* dparams, float* dembeds, float* dprev_belief,
                                     float* dprev_state, int accumulate, unsigned* status, void* ws,
                                     size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(dims_ok(T, B, A, D, Hd, S) && E > 0 && T > 0, REPO_E_SHAPE);
  REPO_REQUIRE(params && nonterms && embeds && !eps_prior == !eps_post && featx && prior_std && post_std && xsa && e &&
                   gates && hp && hq && dparams,
               REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_rssm_observe_bwd_workspace_bytes(T, B, A, D, Hd, S, E), REPO_E_WS_TOO_SMALL);
  const size_t rows = (size_t)T * B;
  float* w = (float*)ws;
  float* doutp = w;  w += rows * 2 * S;
  float* doutq = w;  w += rows * 2 * S;
  float* dhp = w;    w += rows * Hd;
  float* dhq = w;    w += rows * Hd;
  float* dgi = w;    w += rows * 3 * D;
  float* dgh = w;    w += rows * 3 * D;
  float* de = w;     w += rows * D;
  // align the slab region to 256 B
  uintptr_t sl = ((uintptr_t)w + 255) & ~(uintptr_t)255;
  void* slab = (void*)sl;
  const size_t slab_bytes = ws_bytes - (sl - (uintptr_t)ws);

  const float* const* P = params;
  const bool cs_engine = (accumulate & 2) != 0;
  accumulate &= 1;
  if (cs_engine) {
    // column-split, weight-stationary reverse scan (scan_cs.hip).  The prior head is off the recurrence: its output
    // deltas, its hidden delta and its share of d belief_t are three launches over all T*B rows, in front of the scan
    REPO_REQUIRE(scan_cs_ok(T, B, A, D, Hd, S), REPO_E_SHAPE);
    const int n = (int)(rows * S);
    hipLaunchKernelGGL(prior_delta_kernel, dim3(cdiv(n, 256) > 1024 ? 1024 : cdiv(n, 256)), dim3(256), 0, stream, n, (int)S,
                       dprior_state, dpm, dps, prior_std, NoiseSrc{eps_prior, noise_seed, noise_offset}, min_std, doutp);
    REPO_CHECK_LAUNCH();
    int rc1 = repo_gemm(0, 0, (int64_t)rows, Hd, 2 * S, doutp, 2 * S, P[8], Hd, nullptr, 1, dhp, Hd, REPO_EPI_MUL_DELU, hp,
                        Hd, 0, stream);
    if (rc1) return rc1;
    float* dbx = (float*)slab;
    if ((rc1 = repo_gemm(0, 0, (int64_t)rows, D, Hd, dhp, Hd, P[6], D, nullptr, 1, dbx, D, REPO_EPI_NONE, nullptr, 0, 0,
                         stream)))
      return rc1;
    void* cws = (void*)(((uintptr_t)(dbx + rows * D) + 255) & ~(uintptr_t)255);
    ScanCsBwd q{T, B, A, D, Hd, S, E, params, nonterms,
                NoiseSrc{eps_post, noise_seed, noise_offset + (uint64_t)(T * B * S)}, min_std,
                featx, post_std, e, gates, hq, dfeat, dqm, dqs, dbx, doutq, dhq, dgi, dgh, de, dprev_belief, dprev_state, status};
    if ((rc1 = scan_cs_bwd(q, cws, slab_bytes - ((uintptr_t)cws - (uintptr_t)slab), stream))) return rc1;
  } else {
  ObsBwdArgs a;
  a.d = ObsDims{(int)T, (int)B, (int)A, (int)D, (int)Hd, (int)S};
  {  // packed weights at the head of the slab region: dead before the first weight-gradient GEMM uses it
    float* pw = (float*)slab;
    const int d = (int)D, h = (int)Hd, s2 = (int)(2 * S), X_ = (int)(S + A);
    K4Jobs packs;
    packs.n = 0;
    auto put = [&](const float* src, int N, int K, int ld, const float** dstp) {
      *dstp = pw;
      pw = pack_k4_add(packs, src, N, K, 1, ld, pw);
    };
    put(P[8], h, s2, h, &a.Wsp);
    put(P[12], h, s2, h, &a.Wsq);
    put(P[6], d, h, d, &a.Wbp);
    put(P[10], d, h, (int)(D + E), &a.Wbq);
    put(P[3], d, 3 * d, d, &a.Whh);
    put(P[2], d, 3 * d, d, &a.Wih);
    put(P[0], (int)S, d, X_, &a.Wsa);
    const int rc0 = pack_k4_launch(packs, stream);
    if (rc0) return rc0;
  }
  a.featx = featx; a.nonterms = nonterms; a.e = e; a.gates = gates; a.hp = hp; a.hq = hq;
  a.prior_std = prior_std; a.post_std = post_std;
  a.eps_prior = NoiseSrc{eps_prior, noise_seed, noise_offset};
  a.eps_post = NoiseSrc{eps_post, noise_seed, noise_offset + (uint64_t)(T * B * S)};
  a.dfeat = dfeat; a.dprior_state = dprior_state; a.dpm = dpm; a.dps = dps; a.dqm = dqm; a.dqs = dqs;
  a.doutp = doutp; a.doutq = doutq; a.dhp = dhp; a.dhq = dhq; a.dgi = dgi; a.dgh = dgh; a.de = de;
  a.dprev_belief = dprev_belief; a.dprev_state = dprev_state; a.min_std = min_std;
  if (B >= 512) {
    hipLaunchKernelGGL((observe_bwd_kernel<4, 1>), dim3(cdiv(B, 4)), dim3(256), 0, stream, a);
  } else if (B >= 128) {
    hipLaunchKernelGGL((observe_bwd_kernel<2, 2>), dim3(cdiv(B, 2)), dim3(512), 0, stream, a);
  } else if (B >= 32) {
    hipLaunchKernelGGL((observe_bwd_kernel<2, 4>), dim3(cdiv(B, 2)), dim3(1024), 0, stream, a);
  } else {
    hipLaunchKernelGGL((observe_bwd_kernel<1, 4>), dim3((unsigned)B), dim3(1024), 0, stream, a);
  }
  REPO_CHECK_LAUNCH();

  }
  // deferred weight/bias gradients: (T*B)-row MFMA GEMMs
  float* const* G = dparams;
  const int64_t R_ = (int64_t)rows;
  int rc;
  WgradDesc jobs[7];
  obs_wgrad_jobs(jobs, R_, B, A, D, Hd, S, E, featx, xsa, e, hp, hq, doutp, doutq, dhp, dhq, dgi, dgh, de, G);
  if ((rc = gemm_wgrad_group(jobs, 7, accumulate, slab, slab_bytes, stream))) return rc;
  // fc_embed_belief_posterior, columns [D, D+E): the embedding's share
  if ((rc = repo_gemm_wgrad(R_, Hd, E, dhq, Hd, embeds, E, G[10] + D, D + E, nullptr, accumulate, slab, slab_bytes, stream))) return rc;
  // gradient into the encoder embedding: d embeds = dhq @ W_bq[:, D:]
  if (dembeds)
    if ((rc = repo_gemm(0, 0, R_, E, Hd, dhq, Hd, P[10] + D, D + E, nullptr, 1, dembeds, E, REPO_EPI_NONE, nullptr, 0, 0, stream))) return rc;
  return REPO_OK;
}
