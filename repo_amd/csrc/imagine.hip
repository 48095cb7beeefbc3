// ELU-MLP heads (reward / value / actor) and the latent imagination rollout.
//
// Reference: RewardModel / ValueModel / ActorModel forward (models/decoder.py:189-195,
// models/actor_critic.py:20-26,76-102), TransitionModel.imagine (models/rssm.py:148-184)
// and autograd's backward through them in actor_loss.backward() (dreamer.py:357).
//
// Imagination rows (N = T*B start states) are independent, so every layer of a step is a
// (N x K) x (K x n) GEMM on the fp32 matrix cores; the element-wise pieces between the
// GEMMs (tanh-Normal action sample, GRU gates, softplus + reparameterised prior sample) are
// small fused kernels.  World-model weights are frozen here: the reverse pass computes input
// gradients only, and the actor's own weight gradients are deferred to one pass over all
// (H-1)*N rows (the caller runs repo_mlp_bwd on the d_araw this file emits).
#include "common.h"

namespace repo {

// ------------------------------------------------------------------ element-wise kernels
// raw (rows,2A) -> mean = ms*tanh(raw_m/ms), std = softplus(raw_s + init) + min_std
// if eps: action = tanh(mean + std*eps), xsa[row] = [state(row), action]
__global__ void actor_head_fwd_kernel(int rows, int A, int S, const float* __restrict__ raw,
                                      const float* __restrict__ eps, const float* __restrict__ state, int ldstate,
                                      float min_std, float init_std, float mean_scale, float* __restrict__ mean,
                                      float* __restrict__ stdv, float* __restrict__ xsa, int ldx) {
  const int X = S + A;   // ldx >= X: the conditioned rollout keeps C more columns per xsa row ([state | action | cond])
  const int total = rows * (eps ? X : A);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (eps) {
      const int row = i / X, k = i % X;
      if (k < S) {
        xsa[(size_t)row * ldx + k] = state[(size_t)row * ldstate + k];
      } else {
        const int a = k - S;
        const float mu = mean_scale * tanh_fast(raw[(size_t)row * 2 * A + a] / mean_scale);
        const float sd = softplus(raw[(size_t)row * 2 * A + A + a] + init_std) + min_std;
        mean[(size_t)row * A + a] = mu;
        stdv[(size_t)row * A + a] = sd;
        xsa[(size_t)row * ldx + k] = tanh_fast(fmaf(sd, eps[(size_t)row * A + a], mu));
      }
    } else {
      const int row = i / A, a = i % A;
      mean[i] = mean_scale * tanh_fast(raw[(size_t)row * 2 * A + a] / mean_scale);
      stdv[i] = softplus(raw[(size_t)row * 2 * A + A + a] + init_std) + min_std;
    }
  }
}

// draw (rows,2A) from either (dmean,dstd) [distribution path] or d action [sample path]
__global__ void actor_head_bwd_kernel(int rows, int A, const float* __restrict__ dmean, const float* __restrict__ dstd,
                                      const float* __restrict__ daction, int ldda, const float* __restrict__ action,
                                      int ldact, const float* __restrict__ eps, const float* __restrict__ mean,
                                      const float* __restrict__ stdv, float min_std, float mean_scale,
                                      float* __restrict__ draw, int accumulate) {
  const int total = rows * A;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / A, a = i % A;
    float gm = dmean ? dmean[i] : 0.f, gs = dstd ? dstd[i] : 0.f;
    if (daction) {
      const float act = action[(size_t)row * ldact + a];
      const float du = daction[(size_t)row * ldda + a] * (1.f - act * act);
      gm += du;
      gs = fmaf(du, eps[i], gs);
    }
    const float tm = mean[i] / mean_scale;  // tanh(raw_m / ms)
    const float v0 = gm * (1.f - tm * tm), v1 = gs * (-expm1f(-(stdv[i] - min_std)));
    float* d0 = draw + (size_t)row * 2 * A + a;
    d0[0] = accumulate ? d0[0] + v0 : v0;
    d0[A] = accumulate ? d0[A] + v1 : v1;
  }
}

__global__ void gru_fwd_kernel(int rows, int D, const float* __restrict__ gi, const float* __restrict__ gh,
                               const float* __restrict__ hprev, int ldh, float* __restrict__ hnew, int ldn,
                               float* __restrict__ gates) {
  const int total = rows * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / D, j = i % D;
    const float* a = gi + (size_t)row * 3 * D;
    const float* b = gh + (size_t)row * 3 * D;
    const float rg = sigmoidf(a[j] + b[j]);
    const float zg = sigmoidf(a[D + j] + b[D + j]);
    const float ghn = b[2 * D + j];
    const float ng = tanh_fast(a[2 * D + j] + rg * ghn);
    const float hp = hprev[(size_t)row * ldh + j];
    hnew[(size_t)row * ldn + j] = (1.f - zg) * ng + zg * hp;
    float* g = gates + (size_t)row * 4 * D;
    g[j] = rg;
    g[D + j] = zg;
    g[2 * D + j] = ng;
    g[3 * D + j] = ghn;
  }
}

__global__ void gru_bwd_kernel(int rows, int D, const float* __restrict__ dh, int lddh,
                               const float* __restrict__ gates, const float* __restrict__ hprev, int ldh,
                               float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dhprev, int lddp) {
  const int total = rows * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / D, j = i % D;
    const float* g = gates + (size_t)row * 4 * D;
    const float rg = g[j], zg = g[D + j], ng = g[2 * D + j], ghn = g[3 * D + j];
    const float d = dh[(size_t)row * lddh + j];
    const float hp = hprev[(size_t)row * ldh + j];
    const float g_n = d * (1.f - zg) * (1.f - ng * ng);
    const float g_z = d * (hp - ng) * zg * (1.f - zg);
    const float g_r = g_n * ghn * rg * (1.f - rg);
    float* a = dgi + (size_t)row * 3 * D;
    float* b = dgh + (size_t)row * 3 * D;
    a[j] = g_r;
    a[D + j] = g_z;
    a[2 * D + j] = g_n;
    b[j] = g_r;
    b[D + j] = g_z;
    b[2 * D + j] = g_n * rg;
    dhprev[(size_t)row * lddp + j] = d * zg;
  }
}

__global__ void gauss_fwd_kernel(int rows, int S, const float* __restrict__ out, const float* __restrict__ eps,
                                 float min_std, float* __restrict__ mean, float* __restrict__ stdv,
                                 float* __restrict__ sample, int lds_) {
  const int total = rows * S;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / S, s = i % S;
    const float mu = out[(size_t)row * 2 * S + s];
    const float sd = softplus(out[(size_t)row * 2 * S + S + s]) + min_std;
    mean[i] = mu;
    stdv[i] = sd;
    sample[(size_t)row * lds_ + s] = fmaf(sd, eps[i], mu);
  }
}

__global__ void gauss_bwd_kernel(int rows, int S, const float* __restrict__ dsample, int ldds,
                                 const float* __restrict__ dmean, const float* __restrict__ dstd,
                                 const float* __restrict__ stdv, const float* __restrict__ eps, float min_std,
                                 float* __restrict__ dout) {
  const int total = rows * S;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / S, s = i % S;
    const float ds = dsample ? dsample[(size_t)row * ldds + s] : 0.f;
    const float gm = ds + (dmean ? dmean[i] : 0.f);
    const float gs = fmaf(ds, eps[i], dstd ? dstd[i] : 0.f);
    dout[(size_t)row * 2 * S + s] = gm;
    dout[(size_t)row * 2 * S + S + s] = gs * (-expm1f(-(stdv[i] - min_std)));
  }
}

// dst[row][0..w) = a[row][0..w) (+ b[row][0..w))
__global__ void add_cols_kernel(int rows, int w, const float* __restrict__ a, int lda, const float* __restrict__ b,
                                int ldb, float* __restrict__ dst, int ldd) {
  const int total = rows * w;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / w, j = i % w;
    float v = a[(size_t)row * lda + j];
    if (b) v += b[(size_t)row * ldb + j];
    dst[(size_t)row * ldd + j] = v;
  }
}

static inline int ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

#define REPO_RC(x)        \
  do {                    \
    int _rc = (x);        \
    if (_rc) return _rc;  \
  } while (0)

static inline int lin(int64_t rows, int64_t n, int64_t k, const float* x, int64_t ldx, const float* w,
                      const float* b, float* y, int64_t ldy, int epi, hipStream_t s) {
  return repo_gemm(0, 1, rows, n, k, x, ldx, w, k, b, 1, y, ldy, epi, nullptr, 0, 0, s);
}
// dx = (dy @ W) [* elu'(aux)]
static inline int lin_bwd_data(int64_t rows, int64_t n, int64_t k, const float* dy, int64_t lddy, const float* w,
                               float* dx, int64_t lddx, const float* aux, int64_t ldaux, int accumulate,
                               hipStream_t s) {
  return repo_gemm(0, 0, rows, k, n, dy, lddy, w, k, nullptr, 1, dx, lddx, aux ? REPO_EPI_MUL_DELU : REPO_EPI_NONE,
                   aux, ldaux, accumulate, s);
}

// Materialise normals [offset, offset+n) of stream `seed` (the per-step engine takes tensors).
__global__ __launch_bounds__(256) void philox_fill_kernel(float* __restrict__ out, long n, uint64_t seed, uint64_t offset) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = philox_normal(seed, offset + i);
}
int philox_fill(float* out, long n, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (n <= 0) return REPO_OK;
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, out, n, seed, offset);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// persistent row-tiled rollout (imagine16.hip)
bool imagine_fused_ok(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, int n_actor_layers, int64_t C);
size_t imagine_fused_fwd_ws_floats(int64_t A, int64_t D, int64_t Hd, int64_t S);
size_t imagine_fused_bwd_ws_floats(int64_t A, int64_t D, int64_t Hd, int64_t S);
int imagine_fused_fwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, const float* const* rp,
                      const float* const* ap, const float* belief0, const float* state0, const float* cond, int64_t C,
                      NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std, float a_init_std, float a_mean_scale,
                      float* featx, float* prior_mean, float* prior_std, float* a_hidden, int64_t a_layer_rows,
                      float* a_raw, float* a_mean, float* a_std, float* xsa, float* e, float* gates, float* hp, void* ws,
                      hipStream_t stream);
int imagine_fused_bwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, const float* const* rp,
                      int64_t C, NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std,
                      float a_mean_scale, const float* featx, const float* prior_std, const float* a_mean,
                      const float* a_std, const float* xsa, const float* e, const float* gates, const float* hp,
                      const float* dfeat, const float* dprior_mean, const float* dprior_std, float* d_araw,
                      float* dfeat0, void* ws, hipStream_t stream);

// the same rollout on 32-row tiles and the bf16 matrix pipe (imagine32.hip)
bool imagine32_ok(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, int n_actor_layers, int64_t C);
size_t imagine32_fwd_ws_bytes(int64_t A, int64_t D, int64_t Hd, int64_t S);
int imagine32_fwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, const float* const* rp,
                  const float* const* ap, const float* belief0, const float* state0, const float* cond, int64_t C,
                  NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std, float a_init_std,
                  float a_mean_scale, float* featx, float* prior_mean, float* prior_std, float* a_hidden,
                  int64_t a_layer_rows, float* a_raw, float* a_mean, float* a_std, float* xsa, float* e, float* gates,
                  float* hp, void* ws, hipStream_t stream);

size_t imagine32_bwd_ws_bytes(int64_t A, int64_t D, int64_t Hd, int64_t S);
int imagine32_bwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, const float* const* rp, int64_t C,
                  NoiseSrc eps_act, NoiseSrc eps_prior, float min_std, float a_min_std, float a_mean_scale,
                  const float* featx, const float* prior_std, const float* a_mean, const float* a_std, const float* xsa,
                  const float* e, const float* gates, const float* hp, const float* dfeat, const float* dprior_mean,
                  const float* dprior_std, float* d_araw, float* dfeat0, void* ws, hipStream_t stream);

// fused dense heads (mlp16.hip)
bool mlp_fused_ok(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers);
size_t mlp_fused_ws_floats(int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers);
int mlp_fused_fwd(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int L, const float* x, int64_t ldx,
                  const float* const* params, float* const* hidden_out, float* out, int64_t ldo, void* ws,
                  hipStream_t stream);
int mlp_fused_bwd(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int L, const float* const* params,
                  const float* const* hidden_acts, const float* dout, int64_t lddout, float* const* dsave, float* dx,
                  int64_t lddx, int accumulate_dx, void* ws, hipStream_t stream, const float* dout_w = nullptr,
                  int64_t rows_w = 0);

}  // namespace repo

using namespace repo;

// =================================================================== MLP heads
// the fused kernels move biases and hidden activations as 16-byte quads
static bool mlp_quads_aligned(int n_layers, const float* const* params, const float* const* hid) {
  uintptr_t bits = 0;
  for (int l = 0; l < n_layers - 1; ++l) bits |= (uintptr_t)params[2 * l + 1] | (uintptr_t)hid[l];
  return (bits & 15) == 0;
}

static int mlp_fwd_layers(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers, const float* x,
                          int64_t ldx, const float* const* params, float* const* hidden_out, float* out, int64_t ldo,
                          hipStream_t stream) {
  const float* cur = x;
  int64_t ldc = ldx, kc = in_dim;
  for (int l = 0; l < n_layers; ++l) {
    const bool last = l == n_layers - 1;
    float* y = last ? out : hidden_out[l];
    const int64_t n = last ? out_dim : hidden, ldy = last ? ldo : hidden;
    REPO_RC(lin(rows, n, kc, cur, ldc, params[2 * l], params[2 * l + 1], y, ldy, last ? REPO_EPI_NONE : REPO_EPI_ELU,
                stream));
    cur = y;
    ldc = ldy;
    kc = n;
  }
  return REPO_OK;
}

extern "C" size_t repo_mlp_fwd_workspace_bytes(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim,
                                               int n_layers) {
  return mlp_fused_ok(rows, in_dim, hidden, out_dim, n_layers)
             ? mlp_fused_ws_floats(in_dim, hidden, out_dim, n_layers) * sizeof(float)
             : 0;
}

extern "C" int repo_mlp_fwd(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers,
                            const float* x, int64_t ldx, const float* const* params, float* const* hidden_out,
                            float* out, int64_t ldo, void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(rows >= 0 && in_dim > 0 && hidden > 0 && out_dim > 0 && n_layers >= 1, REPO_E_SHAPE);
  if (rows == 0) return REPO_OK;
  REPO_REQUIRE(x && params && out && (n_layers == 1 || hidden_out), REPO_E_BADARG);
  // the fused kernel reads x through a 32-bit buffer descriptor
  if (mlp_fused_ok(rows, in_dim, hidden, out_dim, n_layers) && (rows - 1) * ldx + in_dim < (int64_t(1) << 30) &&
      mlp_quads_aligned(n_layers, params, hidden_out)) {
    REPO_REQUIRE(ws && ws_bytes >= repo_mlp_fwd_workspace_bytes(rows, in_dim, hidden, out_dim, n_layers),
                 REPO_E_WS_TOO_SMALL);
    return mlp_fused_fwd(rows, in_dim, hidden, out_dim, n_layers, x, ldx, params, hidden_out, out, ldo, ws, stream);
  }
  return mlp_fwd_layers(rows, in_dim, hidden, out_dim, n_layers, x, ldx, params, hidden_out, out, ldo, stream);
}

// workspace: [pre-activation gradients: (n_layers - 1) x rows x hidden | weight-gradient slab | packs]
static int mlp_wgrad_jobs(WgradDesc* d, int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers,
                          const float* x, int64_t ldx, const float* const* hidden_acts, const float* dout,
                          int64_t lddout, float* const* dsave, float* const* dparams) {
  for (int l = 0; l < n_layers; ++l) {
    const bool last = l == n_layers - 1;
    const int64_t n = last ? out_dim : hidden, k = (l == 0) ? in_dim : hidden;
    d[l] = WgradDesc{rows, n, k, last ? dout : (dsave ? dsave[l] : nullptr), last ? lddout : hidden,
                     (l == 0) ? x : (hidden_acts ? hidden_acts[l - 1] : nullptr), (l == 0) ? ldx : hidden,
                     dparams ? dparams[2 * l] : nullptr, k, dparams ? dparams[2 * l + 1] : nullptr};
  }
  return n_layers;
}
static size_t mlp_bwd_slab_bytes(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers) {
  if (n_layers > kMaxWgradGroup - 1) {  // layer by layer: the largest slab
    size_t slab = repo_gemm_wgrad_workspace_bytes(rows, hidden, in_dim);
    const size_t s2 = repo_gemm_wgrad_workspace_bytes(rows, hidden, hidden);
    const size_t s3 = repo_gemm_wgrad_workspace_bytes(rows, out_dim, n_layers > 1 ? hidden : in_dim);
    if (s2 > slab) slab = s2;
    if (s3 > slab) slab = s3;
    return (slab + 255) & ~(size_t)255;
  }
  WgradDesc d[kMaxWgradGroup];
  const int n = mlp_wgrad_jobs(d, rows, in_dim, hidden, out_dim, n_layers, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr);
  return gemm_wgrad_group_ws_bytes(d, n);
}
extern "C" size_t repo_mlp_bwd_workspace_bytes(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim,
                                               int n_layers) {
  const bool fused = mlp_fused_ok(rows, in_dim, hidden, out_dim, n_layers);
  const size_t nd = fused ? (size_t)(n_layers - 1) : 2;
  return nd * (size_t)rows * hidden * sizeof(float) + mlp_bwd_slab_bytes(rows, in_dim, hidden, out_dim, n_layers) +
         (fused ? mlp_fused_ws_floats(in_dim, hidden, out_dim, n_layers) * sizeof(float) : 0) + 512;
}

extern "C" int repo_mlp_bwd(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers,
                            const float* x, int64_t ldx, const float* const* params,
                            const float* const* hidden_acts, const float* dout, int64_t lddout,
                            float* const* dparams, int accumulate_w, float* dx, int64_t lddx, int accumulate_dx,
                            const float* dout_w, int64_t rows_w, void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(rows > 0 && in_dim > 0 && hidden > 0 && out_dim > 0 && n_layers >= 1, REPO_E_SHAPE);
  REPO_REQUIRE(x && params && dout && (n_layers == 1 || hidden_acts), REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_mlp_bwd_workspace_bytes(rows, in_dim, hidden, out_dim, n_layers),
               REPO_E_WS_TOO_SMALL);
  if (dout_w) {   // two upstream gradients: dx from dout (all rows), the weight gradients from dout_w (the first rows_w rows)
    REPO_REQUIRE(out_dim == 1 && lddout == 1 && dparams && dx && rows_w > 0 && rows_w <= rows, REPO_E_BADARG);
    if (!(mlp_fused_ok(rows, in_dim, hidden, out_dim, n_layers) && mlp_quads_aligned(n_layers, params, hidden_acts) &&
          ((uintptr_t)ws & 15) == 0)) {
      // no fused chain for this shape: the two passes the one-chain form replaces
      REPO_RC(repo_mlp_bwd(rows, in_dim, hidden, out_dim, n_layers, x, ldx, params, hidden_acts, dout, lddout, nullptr, 0,
                           dx, lddx, accumulate_dx, nullptr, 0, ws, ws_bytes, stream));
      return repo_mlp_bwd(rows_w, in_dim, hidden, out_dim, n_layers, x, ldx, params, hidden_acts, dout_w, 1, dparams,
                          accumulate_w, nullptr, 0, 0, nullptr, 0, ws, ws_bytes, stream);
    }
  }
  const bool shape_ok = mlp_fused_ok(rows, in_dim, hidden, out_dim, n_layers);
  const bool fused = shape_ok && mlp_quads_aligned(n_layers, params, hidden_acts) && ((uintptr_t)ws & 15) == 0;
  const size_t nd = shape_ok ? (size_t)(n_layers - 1) : 2;
  float* d0 = (float*)ws;
  uintptr_t sl = ((uintptr_t)(d0 + nd * (size_t)rows * hidden) + 255) & ~(uintptr_t)255;
  void* slab = (void*)sl;
  const size_t slab_bytes = mlp_bwd_slab_bytes(rows, in_dim, hidden, out_dim, n_layers);
  if (fused) {
    // one kernel for the whole reverse chain; the weight gradients are GEMMs over the saved pre-activation gradients
    float* dsave[8];
    for (int l = 0; l < n_layers - 1; ++l) dsave[l] = d0 + (size_t)l * rows * hidden;
    REPO_RC(mlp_fused_bwd(rows, in_dim, hidden, out_dim, n_layers, params, hidden_acts, dout, lddout,
                          dparams ? dsave : nullptr, dx, lddx, accumulate_dx, (void*)(sl + slab_bytes), stream, dout_w,
                          rows_w));
    if (!dparams) return REPO_OK;
    WgradDesc jobs[kMaxWgradGroup];
    const int nj = mlp_wgrad_jobs(jobs, dout_w ? rows_w : rows, in_dim, hidden, out_dim, n_layers, x, ldx, hidden_acts,
                                  dout_w ? dout_w : dout, lddout, dsave, dparams);
    return gemm_wgrad_group(jobs, nj, accumulate_w, slab, slab_bytes, stream);
  }
  float* d1 = d0 + (size_t)rows * hidden;
  const float* dcur = dout;
  int64_t lddc = lddout;
  for (int l = n_layers - 1; l >= 0; --l) {
    const int64_t n = (l == n_layers - 1) ? out_dim : hidden;
    const int64_t k = (l == 0) ? in_dim : hidden;
    const float* inp = (l == 0) ? x : hidden_acts[l - 1];
    const int64_t ldin = (l == 0) ? ldx : hidden;
    if (dparams)
      REPO_RC(repo_gemm_wgrad(rows, n, k, dcur, lddc, inp, ldin, dparams[2 * l], k, dparams[2 * l + 1], accumulate_w,
                              slab, slab_bytes, stream));
    if (l > 0) {
      float* dn = (dcur == d0) ? d1 : d0;
      REPO_RC(lin_bwd_data(rows, n, k, dcur, lddc, params[2 * l], dn, hidden, hidden_acts[l - 1], hidden, 0, stream));
      dcur = dn;
      lddc = hidden;
    } else if (dx) {
      REPO_RC(lin_bwd_data(rows, n, k, dcur, lddc, params[0], dx, lddx, nullptr, 0, accumulate_dx, stream));
    }
  }
  return REPO_OK;
}

// =================================================================== actor distribution head
static int actor_head_fwd_ld(int64_t rows, int64_t A, int64_t S, const float* raw, const float* eps,
                             const float* state, int64_t ldstate, float min_std, float init_std, float mean_scale,
                             float* mean, float* std, float* xsa, int64_t ldx, hipStream_t stream) {
  REPO_REQUIRE(rows > 0 && A > 0 && ldx >= S + A && rows * ldx < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(raw && mean && std && (!eps || (state && xsa)), REPO_E_BADARG);
  hipLaunchKernelGGL(actor_head_fwd_kernel, dim3(ew_blocks(rows * (S + A))), dim3(256), 0, stream, (int)rows, (int)A,
                     (int)S, raw, eps, state, (int)ldstate, min_std, init_std, mean_scale, mean, std, xsa, (int)ldx);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

extern "C" int repo_actor_head_fwd(int64_t rows, int64_t A, int64_t S, const float* raw, const float* eps,
                                   const float* state, int64_t ldstate, float min_std, float init_std,
                                   float mean_scale, float* mean, float* std, float* xsa, hipStream_t stream) {
  REPO_ARCH_GUARD();
  return actor_head_fwd_ld(rows, A, S, raw, eps, state, ldstate, min_std, init_std, mean_scale, mean, std, xsa, S + A,
                           stream);
}

extern "C" int repo_actor_head_bwd(int64_t rows, int64_t A, const float* dmean, const float* dstd,
                                   const float* daction, int64_t ldda, const float* action, int64_t ldact,
                                   const float* eps, const float* mean, const float* std, float min_std,
                                   float mean_scale, float* draw, int accumulate, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(rows > 0 && A > 0 && rows * 2 * A < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(mean && std && draw && (!daction || (action && eps)), REPO_E_BADARG);
  hipLaunchKernelGGL(actor_head_bwd_kernel, dim3(ew_blocks(rows * A)), dim3(256), 0, stream, (int)rows, (int)A, dmean,
                     dstd, daction, (int)ldda, action, (int)ldact, eps, mean, std, min_std, mean_scale, draw, accumulate);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

// =================================================================== imagination rollout
// most condition columns the per-step engine takes (the multitask kernels' own limit is 13: multitask.hip)
constexpr int64_t kImgMaxCond = 16;
static bool img_dims_ok(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S) {
  return Hm >= 1 && N > 0 && A > 0 && D > 0 && Hd > 0 && S > 0 && (Hm + 1) * N * 4 * D < kMaxIdx;
}

extern "C" size_t repo_rssm_imagine_fwd_workspace_bytes(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd,
                                                        int64_t S) {
  // per-step engine: gate / head scratch + room to materialise the two noise tensors when they are drawn in-library
  // (+ the [belief | state | cond] rows of one step when the rollout is conditioned and outside the persistent
  // engines' shapes: C <= kImgMaxCond)
  const size_t unfused = ((size_t)N * (6 * D + 2 * S + D + S + kImgMaxCond) + (size_t)Hm * N * (A + S)) * sizeof(float);
  size_t fused = imagine_fused_fwd_ws_floats(A, D, Hd, S) * sizeof(float);
  const size_t fused32 = imagine32_fwd_ws_bytes(A, D, Hd, S);
  if (fused32 > fused) fused = fused32;
  return unfused > fused ? unfused : fused;
}

extern "C" int repo_rssm_imagine_fwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S,
                                     int n_actor_layers, const float* const* rssm_params,
                                     const float* const* actor_params, const float* belief0, const float* state0,
                                     const float* cond, int64_t C,
                                     const float* eps_act, const float* eps_prior, uint64_t noise_seed,
                                     uint64_t noise_offset, float min_std, float a_min_std,
                                     float a_init_std, float a_mean_scale, float* featx, float* prior_mean,
                                     float* prior_std, float* a_hidden, int64_t a_layer_rows, float* a_raw,
                                     float* a_mean, float* a_std, float* xsa, float* e, float* gates, float* hp,
                                     void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(img_dims_ok(Hm, N, A, D, Hd, S) && n_actor_layers >= 2 && n_actor_layers <= 8, REPO_E_SHAPE);
  REPO_REQUIRE(a_layer_rows >= Hm * N, REPO_E_SHAPE);
  REPO_REQUIRE(rssm_params && actor_params && belief0 && state0 && !eps_act == !eps_prior && featx && prior_mean &&
                   prior_std && a_hidden && a_raw && a_mean && a_std && xsa && e && gates && hp,
               REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_rssm_imagine_fwd_workspace_bytes(Hm, N, A, D, Hd, S), REPO_E_WS_TOO_SMALL);
  REPO_REQUIRE(C >= 0 && (C == 0 || cond), REPO_E_BADARG);
  // a conditioned rollout rides in the K padding of the persistent engines' tiles where it fits (230 + C <= 240 and
  // S + A + C within S + A's 16-column block: A = 3..8 at S = 30); other shapes -- the reference's A = 2 multitask suites,
  // tabletop/pointmass.py:114 and the dmc-mixed tasks -- take the per-step engine below with widened rows
  REPO_REQUIRE(C <= kImgMaxCond, REPO_E_SHAPE);
  if (imagine32_ok(Hm, N, A, D, Hd, S, n_actor_layers, C))
    return imagine32_fwd(Hm, N, A, D, Hd, S, rssm_params, actor_params, belief0, state0, cond, C,
                         NoiseSrc{eps_act, noise_seed, noise_offset},
                         NoiseSrc{eps_prior, noise_seed, noise_offset + (uint64_t)(Hm * N * A)}, min_std, a_min_std,
                         a_init_std, a_mean_scale, featx, prior_mean, prior_std, a_hidden, a_layer_rows, a_raw, a_mean,
                         a_std, xsa, e, gates, hp, ws, stream);
  if (imagine_fused_ok(Hm, N, A, D, Hd, S, n_actor_layers, C))
    return imagine_fused_fwd(Hm, N, A, D, Hd, S, rssm_params, actor_params, belief0, state0, cond, C,
                             NoiseSrc{eps_act, noise_seed, noise_offset},
                             NoiseSrc{eps_prior, noise_seed, noise_offset + (uint64_t)(Hm * N * A)}, min_std, a_min_std, a_init_std, a_mean_scale, featx, prior_mean, prior_std, a_hidden,
                             a_layer_rows, a_raw, a_mean, a_std, xsa, e, gates, hp, ws, stream);
  const int64_t F = D + S, X = S + A + C, Fw = F + C, rowsAll = Hm * N;   // X, Fw: the conditioned rows' widths
  const float* const* P = rssm_params;
  float* gi = (float*)ws;
  float* gh = gi + (size_t)N * 3 * D;
  float* pout = gh + (size_t)N * 3 * D;
  float* wide = pout + (size_t)N * 2 * S;   // [belief | state | cond] rows of the current step (C > 0)
  if (!eps_act) {  // draw the tensors the fused engine would have drawn element by element
    float* na = wide + (size_t)N * (F + kImgMaxCond);
    float* np_ = na + (size_t)Hm * N * A;
    REPO_RC(philox_fill(na, Hm * N * A, noise_seed, noise_offset, stream));
    REPO_RC(philox_fill(np_, Hm * N * S, noise_seed, noise_offset + (uint64_t)(Hm * N * A), stream));
    eps_act = na;
    eps_prior = np_;
  }
  // slot 0 = start states
  hipLaunchKernelGGL(add_cols_kernel, dim3(ew_blocks(N * D)), dim3(256), 0, stream, (int)N, (int)D, belief0, (int)D,
                     (const float*)nullptr, 0, featx, (int)F);
  hipLaunchKernelGGL(add_cols_kernel, dim3(ew_blocks(N * S)), dim3(256), 0, stream, (int)N, (int)S, state0, (int)S,
                     (const float*)nullptr, 0, featx + D, (int)F);
  REPO_CHECK_LAUNCH();
  for (int64_t t = 0; t < Hm; ++t) {
    const float* ft = featx + (size_t)t * N * F;
    float* fn = featx + (size_t)(t + 1) * N * F;
    const size_t r0 = (size_t)t * N;
    // actor MLP on (detached) [belief, state]
    float* hid[8];
    for (int l = 0; l < n_actor_layers - 1; ++l) hid[l] = a_hidden + ((size_t)l * a_layer_rows + r0) * Hd;
    const float* ain = ft;
    if (C) {   // the actor's fc1 and W_sa carry C more input columns: [belief | state | cond], [state | action | cond]
      hipLaunchKernelGGL(add_cols_kernel, dim3(ew_blocks(N * F)), dim3(256), 0, stream, (int)N, (int)F, ft, (int)F,
                         (const float*)nullptr, 0, wide, (int)Fw);
      hipLaunchKernelGGL(add_cols_kernel, dim3(ew_blocks(N * C)), dim3(256), 0, stream, (int)N, (int)C, cond, (int)C,
                         (const float*)nullptr, 0, wide + F, (int)Fw);
      hipLaunchKernelGGL(add_cols_kernel, dim3(ew_blocks(N * C)), dim3(256), 0, stream, (int)N, (int)C, cond, (int)C,
                         (const float*)nullptr, 0, xsa + r0 * X + S + A, (int)X);
      REPO_CHECK_LAUNCH();
      ain = wide;
    }
    REPO_RC(mlp_fwd_layers(N, Fw, Hd, 2 * A, n_actor_layers, ain, Fw, actor_params, hid, a_raw + r0 * 2 * A, 2 * A, stream));
    REPO_RC(actor_head_fwd_ld(N, A, S, a_raw + r0 * 2 * A, eps_act + r0 * A, ft + D, F, a_min_std, a_init_std,
                              a_mean_scale, a_mean + r0 * A, a_std + r0 * A, xsa + r0 * X, X, stream));
    // belief update
    REPO_RC(lin(N, D, X, xsa + r0 * X, X, P[0], P[1], e + r0 * D, D, REPO_EPI_ELU, stream));
    REPO_RC(lin(N, 3 * D, D, e + r0 * D, D, P[2], P[4], gi, 3 * D, REPO_EPI_NONE, stream));
    REPO_RC(lin(N, 3 * D, D, ft, F, P[3], P[5], gh, 3 * D, REPO_EPI_NONE, stream));
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(ew_blocks(N * D)), dim3(256), 0, stream, (int)N, (int)D, gi, gh, ft,
                       (int)F, fn, (int)F, gates + r0 * 4 * D);
    REPO_CHECK_LAUNCH();
    // prior head
    REPO_RC(lin(N, Hd, D, fn, F, P[6], P[7], hp + r0 * Hd, Hd, REPO_EPI_ELU, stream));
    REPO_RC(lin(N, 2 * S, Hd, hp + r0 * Hd, Hd, P[8], P[9], pout, 2 * S, REPO_EPI_NONE, stream));
    hipLaunchKernelGGL(gauss_fwd_kernel, dim3(ew_blocks(N * S)), dim3(256), 0, stream, (int)N, (int)S, pout,
                       eps_prior + r0 * S, min_std, prior_mean + r0 * S, prior_std + r0 * S, fn + D, (int)F);
    REPO_CHECK_LAUNCH();
  }
  return REPO_OK;
}

extern "C" size_t repo_rssm_imagine_bwd_workspace_bytes(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd,
                                                        int64_t S) {
  // g(F) carry(F) dpout(2S) dhp(Hd) dbel(D) dgi(3D) dgh(3D) de(D) dxsa(S+A) + the two noise tensors
  const size_t unfused = ((size_t)N * (2 * (D + S) + 2 * S + Hd + D + 6 * D + D + (S + A + kImgMaxCond)) + (size_t)Hm * N * (A + S)) * sizeof(float);
  size_t fused = imagine_fused_bwd_ws_floats(A, D, Hd, S) * sizeof(float);
  if (fused < imagine32_bwd_ws_bytes(A, D, Hd, S)) fused = imagine32_bwd_ws_bytes(A, D, Hd, S);
  return unfused > fused ? unfused : fused;
}

extern "C" int repo_rssm_imagine_bwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, int64_t C,
                                     const float* const* rssm_params, const float* eps_act, const float* eps_prior,
                                     uint64_t noise_seed, uint64_t noise_offset, float min_std, float a_min_std,
                                     float a_mean_scale, const float* featx,
                                     const float* prior_std, const float* a_mean, const float* a_std,
                                     const float* xsa, const float* e, const float* gates, const float* hp,
                                     const float* dfeat, const float* dprior_mean, const float* dprior_std,
                                     float* d_araw, float* dfeat0, void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(img_dims_ok(Hm, N, A, D, Hd, S), REPO_E_SHAPE);
  REPO_REQUIRE(rssm_params && !eps_act == !eps_prior && featx && prior_std && a_mean && a_std && xsa && e && gates &&
                   hp && dfeat && d_araw,
               REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_rssm_imagine_bwd_workspace_bytes(Hm, N, A, D, Hd, S), REPO_E_WS_TOO_SMALL);
  REPO_REQUIRE(C >= 0 && C <= kImgMaxCond, REPO_E_SHAPE);
  if (imagine32_ok(Hm, N, A, D, Hd, S, 5, C))
    return imagine32_bwd(Hm, N, A, D, Hd, S, rssm_params, C, NoiseSrc{eps_act, noise_seed, noise_offset},
                         NoiseSrc{eps_prior, noise_seed, noise_offset + (uint64_t)(Hm * N * A)}, min_std, a_min_std,
                         a_mean_scale, featx, prior_std, a_mean, a_std, xsa, e, gates, hp, dfeat, dprior_mean,
                         dprior_std, d_araw, dfeat0, ws, stream);
  if (imagine_fused_ok(Hm, N, A, D, Hd, S, 5, C))
    return imagine_fused_bwd(Hm, N, A, D, Hd, S, rssm_params, C, NoiseSrc{eps_act, noise_seed, noise_offset},
                             NoiseSrc{eps_prior, noise_seed, noise_offset + (uint64_t)(Hm * N * A)}, min_std, a_min_std, a_mean_scale,
                             featx, prior_std, a_mean, a_std, xsa, e, gates, hp, dfeat, dprior_mean, dprior_std,
                             d_araw, dfeat0, ws, stream);
  const int64_t F = D + S, X = S + A + C;   // xsa rows = [state | action | cond]; the cond columns' gradient is unused
  const float* const* P = rssm_params;
  float* w = (float*)ws;
  float* g = w;      w += (size_t)N * F;   // total grad on featx[t+1]
  float* carry = w;  w += (size_t)N * F;   // grad flowing from step t+1 into featx[t+1]
  float* dpout = w;  w += (size_t)N * 2 * S;
  float* dhp = w;    w += (size_t)N * Hd;
  float* dbel = w;   w += (size_t)N * D;
  float* dgi = w;    w += (size_t)N * 3 * D;
  float* dgh = w;    w += (size_t)N * 3 * D;
  float* de = w;     w += (size_t)N * D;
  float* dxsa = w;   w += (size_t)N * X;
  if (!eps_act) {
    float* na = w;
    float* np_ = na + (size_t)Hm * N * A;
    REPO_RC(philox_fill(na, Hm * N * A, noise_seed, noise_offset, stream));
    REPO_RC(philox_fill(np_, Hm * N * S, noise_seed, noise_offset + (uint64_t)(Hm * N * A), stream));
    eps_act = na;
    eps_prior = np_;
  }
  bool have_carry = false;
  for (int64_t t = Hm - 1; t >= 0; --t) {
    const size_t r0 = (size_t)t * N;
    const float* ft = featx + (size_t)t * N * F;
    // g = dfeat[t] (+ carry)
    hipLaunchKernelGGL(add_cols_kernel, dim3(ew_blocks(N * F)), dim3(256), 0, stream, (int)N, (int)F,
                       dfeat + r0 * F, (int)F, have_carry ? (const float*)carry : (const float*)nullptr, (int)F, g,
                       (int)F);
    // prior head backward
    hipLaunchKernelGGL(gauss_bwd_kernel, dim3(ew_blocks(N * S)), dim3(256), 0, stream, (int)N, (int)S, g + D, (int)F,
                       dprior_mean ? dprior_mean + r0 * S : (const float*)nullptr,
                       dprior_std ? dprior_std + r0 * S : (const float*)nullptr, prior_std + r0 * S,
                       eps_prior + r0 * S, min_std, dpout);
    REPO_CHECK_LAUNCH();
    REPO_RC(lin_bwd_data(N, 2 * S, Hd, dpout, 2 * S, P[8], dhp, Hd, hp + r0 * Hd, Hd, 0, stream));
    // dbel = g[:, :D] + dhp @ W_bp   (accumulate into g's belief columns in place)
    REPO_RC(lin_bwd_data(N, Hd, D, dhp, Hd, P[6], g, F, nullptr, 0, 1, stream));
    // GRU backward: writes carry[:, :D] = dbel * z
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(ew_blocks(N * D)), dim3(256), 0, stream, (int)N, (int)D, g, (int)F,
                       gates + r0 * 4 * D, ft, (int)F, dgi, dgh, carry, (int)F);
    REPO_CHECK_LAUNCH();
    REPO_RC(lin_bwd_data(N, 3 * D, D, dgh, 3 * D, P[3], carry, F, nullptr, 0, 1, stream));
    REPO_RC(lin_bwd_data(N, 3 * D, D, dgi, 3 * D, P[2], de, D, e + r0 * D, D, 0, stream));
    REPO_RC(lin_bwd_data(N, D, X, de, D, P[0], dxsa, X, nullptr, 0, 0, stream));
    // carry[:, D:] = d state_t ; d action -> actor head
    hipLaunchKernelGGL(add_cols_kernel, dim3(ew_blocks(N * S)), dim3(256), 0, stream, (int)N, (int)S, dxsa, (int)X,
                       (const float*)nullptr, 0, carry + D, (int)F);
    REPO_CHECK_LAUNCH();
    REPO_RC(repo_actor_head_bwd(N, A, nullptr, nullptr, dxsa + S, X, xsa + r0 * X + S, X, eps_act + r0 * A,
                                a_mean + r0 * A, a_std + r0 * A, a_min_std, a_mean_scale, d_araw + r0 * 2 * A, 0, stream));
    have_carry = true;
    (void)dbel;
  }
  if (dfeat0) {
    hipLaunchKernelGGL(add_cols_kernel, dim3(ew_blocks(N * F)), dim3(256), 0, stream, (int)N, (int)F, carry, (int)F,
                       (const float*)nullptr, 0, dfeat0, (int)F);
    REPO_CHECK_LAUNCH();
  }
  return REPO_OK;
}
