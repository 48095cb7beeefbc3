// Kernels of the multitask (task-conditioned) agents, SURVEY.md section 8 row f4:
//   * FiLM modulation of a conv layer's output, h = relu((1 + gamma[n][c]) * y[n][c][p] + beta[n][c])
//     (ConditionalVisualEncoder.mod / ConditionalVisualObservationModel.mod,
//     /root/reference/algorithms/repo/models/encoder.py:75-88, models/decoder.py:108-123) and its backward;
//   * the KL balance with a PER-ROW Lagrange multiplier beta_row = exp(tasks_row . log_beta) and the dual step on the
//     per-task log_beta vector (MultitaskRePo, /root/reference/algorithms/repo/repo_mt.py:24-32,75-112).
// All are HBM-bound streaming kernels (bytes read once, wave64 shuffle reductions, fixed-order partial sums: bitwise
// reproducible, no float atomics).  The one-hot concatenations of the conditioned dense layers need no kernel: the
// condition is K columns of the caller's feature rows (heads), of the pseudo-actions (observe scan) or of the K
// padding of the rollout's tiles (repo_rssm_imagine_fwd, cond).
#include "common.h"

namespace repo {

constexpr int kMtRedBlocks = 512;
constexpr int kMtMaxTasks = 13;  // 3 + C partial-sum columns <= 16

// ------------------------------------------------------------------ FiLM forward
// One workgroup walks whole (image, channel) planes: the plane's two modulation scalars are wave-uniform loads.
// film[n * ld + goff + c] = gamma, film[n * ld + boff + c] = beta (the reference's film(condition).chunk(2).split(...)).
__global__ __launch_bounds__(256) void film_fwd_kernel(long planes, int C, int P, const float* __restrict__ y,
                                                       const float* __restrict__ film, int ld, int goff, int boff,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  if (P >= 64) {
    for (long pl = wave; pl < planes; pl += nwaves) {
      const long n = pl / C;
      const int c = (int)(pl % C);
      const float g = 1.f + film[n * ld + goff + c], b = film[n * ld + boff + c];
      const float* src = y + pl * P;
      float* dst = out + pl * P;
      for (int p = lane; p < P; p += 64) dst[p] = fmaxf(fmaf(g, src[p], b), 0.f);
    }
  } else {
    // small planes (2x2, 5x5, 6x6): a lane per element over a run of consecutive planes
    const long total = planes * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const long pl = i / P, n = pl / C;
      const int c = (int)(pl % C);
      out[i] = fmaxf(fmaf(1.f + film[n * ld + goff + c], y[i], film[n * ld + boff + c]), 0.f);
    }
  }
}

// ------------------------------------------------------------------ FiLM backward
// dh is the gradient at the ReLU's INPUT (the producer -- a data-gradient kernel with the MUL_DRELU epilogue, or
// repo_relu_mask -- has applied the mask): dy = dh * (1 + gamma); d gamma[n][c] = sum_p dh * y; d beta[n][c] = sum_p dh.
// One wave per plane, fixed summation order.  dfilm rows are written (each (n, c) slot is owned by one plane).
// FROM_H: `y` is the layer's OUTPUT h (the conv ran with REPO_EPI_FILM_RELU): y = (h - beta) / (1 + gamma) where dh != 0.
// h = fl((1 + gamma) y + beta) resolves y to eps |h| / |1 + gamma|, so the recovered y carries a relative error of about
// eps |beta| / |(1 + gamma) y| (measured: d gamma within 5e-6 at |1 + gamma| = 2e-3), and a plane whose 1 + gamma is 0
// has lost y altogether (the reference keeps the exact y: models/encoder.py:84-87): the (nearly) GATED-OFF planes,
// |1 + gamma| < kFilmExactBelow, get their gamma gradient from film_exact_kernel below instead.
constexpr float kFilmExactBelow = 1e-3f;
template <bool FROM_H>
__global__ __launch_bounds__(256) void film_bwd_kernel(long planes, int C, int P, const float* __restrict__ dh,
                                                       const float* __restrict__ y, const float* __restrict__ film,
                                                       int ld, int goff, int boff, float* __restrict__ dy,
                                                       float* __restrict__ dfilm, unsigned* __restrict__ gated_epoch,
                                                       unsigned epoch) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long pl = wave; pl < planes; pl += nwaves) {
    const long n = pl / C;
    const int c = (int)(pl % C);
    const float g = 1.f + film[n * ld + goff + c];
    const float bt = FROM_H ? film[n * ld + boff + c] : 0.f;
    const float rg = (FROM_H && g != 0.f) ? 1.f / g : 0.f;
    // a gated-off plane: tell film_exact_kernel (the launch behind this one) that it has work -- the word holds the epoch of
    // the last call that met one (a monotonic host counter: nothing to reset)
    if (FROM_H && gated_epoch && lane == 0 && fabsf(g) < kFilmExactBelow)
      __hip_atomic_store(gated_epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* d = dh + pl * P;
    const float* src = y + pl * P;
    float* dst = dy + pl * P;
    float sg = 0.f, sb = 0.f;
    for (int p = lane; p < P; p += 64) {
      const float v = d[p];
      const float yy = FROM_H ? (v != 0.f ? (src[p] - bt) * rg : 0.f) : src[p];
      sg = fmaf(v, yy, sg);
      sb += v;
      dst[p] = v * g;
    }
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if (lane == 0) {
      dfilm[n * ld + goff + c] = sg;
      dfilm[n * ld + boff + c] = sb;
    }
  }
}

// The gamma gradient of the GATED-OFF planes, exactly: y recomputed from the layer's own input, weights and bias (a plain
// fp32 dot product per output element), d gamma = sum_p dh * y written over what film_bwd_kernel<true> left there.  One
// launch per modulated layer behind film_bwd_kernel; a workgroup walks its share of the planes, skips every plane with
// |1 + gamma| >= kFilmExactBelow after ONE load (a launch without gated-off planes is a few microseconds), and puts all
// 256 threads on a gated-off one: thread = (pixel, k-slice), the slices of a pixel meet in LDS in a fixed order.  This is
// a slow path by design -- a gated-off (task, channel) pair costs its planes a scalar convolution -- kept off the
// streaming kernel above so that kernel's registers and time are what they were (an in-line version made every launch
// of it slower and a gated-off pair of decoder conv3 cost 4-5 ms: round 6).
struct FilmExact {
  int kind;   // 1: stride-2 convolution big (CB,HB,HB) -> small (CS,HS,HS), the plane is a SMALL channel (encoder layers),
              // 2: its transpose small -> big, the plane is a BIG channel (decoder conv2 / conv3),
              // 3: dense, y[n][c * P + p] = bias[c] + sum_k x[n][k] w[k][c * P + p] (decoder conv1: 1 x 1 -> 5 x 5)
  int CB, CS, HB, HS, KS, K;
  const void* x;       // the layer's input activation (kind 1: big, uint8 frames if x_u8; kind 2: small; kind 3: (nimg, K))
  int x_u8;
  const float* w;      // (CS, CB, KS, KS) as repo_conv_down / repo_conv_up take it; kind 3: (K, C * P)
  const float* bias;   // per output channel (nullable)
};
// the share of one (pixel, slice) of one output element: the outer reduction index (input channel / k) strided by S.  The
// pixel's taps are listed once (offsets into an input plane and into a weight slab), then every outer step issues all of
// them into four independent sums -- the loads of a step are in flight together (nested tap loops with one dependent load
// pair per step ran at ~450 ns per tap: 4 ms for one gated-off pair of decoder conv3).
__device__ float film_exact_part(const FilmExact& e, long n, int c, int p, int C, int P, int sl, int S) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e.kind == 3) {
    const float* xs = (const float*)e.x + (size_t)n * e.K;
    const size_t N = (size_t)C * P, j = (size_t)c * P + p;
    int k = sl;
    for (; k + 3 * S < e.K; k += 4 * S) {
      a0 = fmaf(xs[k], e.w[(size_t)k * N + j], a0);
      a1 = fmaf(xs[k + S], e.w[(size_t)(k + S) * N + j], a1);
      a2 = fmaf(xs[k + 2 * S], e.w[(size_t)(k + 2 * S) * N + j], a2);
      a3 = fmaf(xs[k + 3 * S], e.w[(size_t)(k + 3 * S) * N + j], a3);
    }
    for (; k < e.K; k += S) a0 = fmaf(xs[k], e.w[(size_t)k * N + j], a0);
    return (a0 + a1) + (a2 + a3);
  }
  // the pixel's taps as a rectangle (a, b) in [a0, a1) x [b0, b1): input element (xr0 + xrs a, xc0 + xcs b) of a plane of
  // width XW, weight element (wr0 + wrs a, wc0 + wcs b) of a KS x KS slab -- no per-tap branches, no indexed arrays (a
  // dynamically indexed private array lives in scratch memory: the version with tap lists ran at 6.8 ms per gated-off pair)
  int a0_, a1_, b0_, b1_, xr0, xrs, xc0, xcs, wr0, wrs, wc0, wcs, XW, outer;
  size_t xplane, wslab, xbase, wbase;
  if (e.kind == 1) {
    const int oy = p / e.HS, ox = p % e.HS;
    a0_ = 0, a1_ = e.KS, b0_ = 0, b1_ = e.KS;
    xr0 = 2 * oy, xrs = 1, xc0 = 2 * ox, xcs = 1, wr0 = 0, wrs = 1, wc0 = 0, wcs = 1, XW = e.HB;
    outer = e.CB, xplane = (size_t)e.HB * e.HB, wslab = (size_t)e.KS * e.KS;
    xbase = (size_t)n * e.CB * xplane, wbase = (size_t)c * e.CB * wslab;
  } else {
    // ky = py + 2 a reads input row iyb - a (iyb = (Y - py) / 2): valid for 0 <= iyb - a < HS and ky < KS
    const int Y = p / e.HB, X = p % e.HB, py = Y & 1, px = X & 1, iyb = (Y - py) >> 1, ixb = (X - px) >> 1;
    a0_ = max(0, iyb - e.HS + 1), a1_ = min((e.KS - py + 1) >> 1, iyb + 1);
    b0_ = max(0, ixb - e.HS + 1), b1_ = min((e.KS - px + 1) >> 1, ixb + 1);
    xr0 = iyb, xrs = -1, xc0 = ixb, xcs = -1, wr0 = py, wrs = 2, wc0 = px, wcs = 2, XW = e.HS;
    outer = e.CS, xplane = (size_t)e.HS * e.HS, wslab = (size_t)e.CB * e.KS * e.KS;   // w[cs][cb][ky][kx]
    xbase = (size_t)n * e.CS * xplane, wbase = (size_t)c * e.KS * e.KS;
  }
  auto xat = [&](size_t i) __attribute__((always_inline)) {
    return e.x_u8 ? pix_norm(((const uint8_t*)e.x)[i]) : ((const float*)e.x)[i];
  };
  int o = sl;
  for (; o + 3 * S < outer; o += 4 * S) {   // four channels of the slice at once: eight loads in flight per tap
    const size_t xb = xbase + (size_t)o * xplane, wb = wbase + (size_t)o * wslab;
    const size_t dx = (size_t)S * xplane, dw = (size_t)S * wslab;
    for (int a = a0_; a < a1_; ++a)
      for (int b = b0_; b < b1_; ++b) {
        const size_t xi = xb + (size_t)((xr0 + xrs * a) * XW + xc0 + xcs * b), wi = wb + (size_t)((wr0 + wrs * a) * e.KS + wc0 + wcs * b);
        const float x0 = xat(xi), x1 = xat(xi + dx), x2 = xat(xi + 2 * dx), x3 = xat(xi + 3 * dx);
        const float w0 = e.w[wi], w1 = e.w[wi + dw], w2 = e.w[wi + 2 * dw], w3 = e.w[wi + 3 * dw];
        a0 = fmaf(w0, x0, a0), a1 = fmaf(w1, x1, a1), a2 = fmaf(w2, x2, a2), a3 = fmaf(w3, x3, a3);
      }
  }
  for (; o < outer; o += S) {
    const size_t xb = xbase + (size_t)o * xplane, wb = wbase + (size_t)o * wslab;
    for (int a = a0_; a < a1_; ++a)
      for (int b = b0_; b < b1_; ++b)
        a0 = fmaf(e.w[wb + (size_t)((wr0 + wrs * a) * e.KS + wc0 + wcs * b)], xat(xb + (size_t)((xr0 + xrs * a) * XW + xc0 + xcs * b)), a0);
  }
  return (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void film_exact_kernel(long planes, int C, int P, const float* __restrict__ dh,
                                                         const float* __restrict__ film, int ld, int goff,
                                                         float* __restrict__ dfilm, FilmExact ex,
                                                         const unsigned* __restrict__ gated_epoch, unsigned epoch) {
  // no gated-off plane in this layer (what film_bwd_kernel<true> just found): nothing to do
  if (gated_epoch && __hip_atomic_load(gated_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) return;
  __shared__ float part[256];
  __shared__ float red[16];
  // k-slices per pixel: a power of two, as many as 256 threads give (1 for planes of 256 pixels or more)
  int S = 1;
  while (2 * S * P <= 256) S *= 2;
  const int outer = ex.kind == 1 ? ex.CB : ex.kind == 2 ? ex.CS : ex.K;
  if (S > outer) S = outer;
  for (long pl = blockIdx.x; pl < planes; pl += gridDim.x) {
    const long n = pl / C;
    const int c = (int)(pl % C);
    const float g = 1.f + film[n * ld + goff + c];
    if (!(fabsf(g) < kFilmExactBelow)) continue;   // (uniform over the workgroup)
    float sg = 0.f;
    for (int p0 = 0; p0 < P; p0 += 256 / S) {       // a batch of 256 / S pixels, S slices each
      const int pi = threadIdx.x % (256 / S), sl = threadIdx.x / (256 / S), p = p0 + pi;
      const float v = (p < P && sl < S) ? dh[pl * P + p] : 0.f;
      // (a masked-out element needs no y: dh == 0 there)
      part[threadIdx.x] = (v != 0.f) ? film_exact_part(ex, n, c, p, C, P, sl, S) : 0.f;
      __syncthreads();
      if (sl == 0 && p < P && v != 0.f) {
        float y = ex.bias ? ex.bias[(ex.kind == 3 && ex.KS) ? c * P + p : c] : 0.f;
        for (int q = 0; q < S; ++q) y += part[q * (256 / S) + pi];
        sg = fmaf(v, y, sg);
      }
      __syncthreads();
    }
    sg = block_sum(sg, red);
    if (threadIdx.x == 0) dfilm[n * ld + goff + c] = sg;
  }
}

// ------------------------------------------------------------------ FiLM tables (REPO_EPI_FILM_RELU's aux)
struct FilmTabArgs {
  int nl, ch[4], off[4], total;   // off[l] = sum ch[:l]
};
__global__ __launch_bounds__(256) void film_tables_kernel(long nimg, FilmTabArgs a, const float* __restrict__ film, int ld,
                                                          float* __restrict__ tab) {
  const long per = 2L * a.total;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nimg * per; i += (long)gridDim.x * 256) {
    // flat order of `tab`: [layer][n][2][C_l]
    long r = i;
    int l = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (l + 1 < a.nl && r >= nimg * 2L * a.ch[l]) {
        r -= nimg * 2L * a.ch[l];
        ++l;
      }
    const int C = a.ch[l];
    const long n = r / (2 * C);
    const int w = (int)(r % (2 * C)), half = w / C, c = w % C;
    tab[i] = half ? film[n * ld + a.total + a.off[l] + c] : 1.f + film[n * ld + a.off[l] + c];
  }
}

// ------------------------------------------------------------------ KL balance with per-row beta
// parts is [3 + C][gridDim.x]: sum KL, sum beta_row * viol_row, sum lb_row * viol_row, sum tasks[row][i] * viol_row.
__global__ __launch_bounds__(256) void kl_tasks_kernel(int rows, int S, int C, const float* __restrict__ pm,
                                                       const float* __restrict__ ps, const float* __restrict__ qm,
                                                       const float* __restrict__ qs, float alpha,
                                                       const float* __restrict__ log_beta,
                                                       const float* __restrict__ tasks, float target_kl, float scale,
                                                       float* __restrict__ dpm, float* __restrict__ dps,
                                                       float* __restrict__ dqm, float* __restrict__ dqs,
                                                       float* __restrict__ parts) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float acc[3 + kMtMaxTasks];
#pragma unroll
  for (int i = 0; i < 3 + kMtMaxTasks; ++i) acc[i] = 0.f;
  for (int row = blockIdx.x * nw + wid; row < rows; row += gridDim.x * nw) {
    float kl = 0.f, gpm = 0.f, gps = 0.f, gqm = 0.f, gqs = 0.f;
    const size_t o = (size_t)row * S + lane;
    if (lane < S) {
      const float mp = pm[o], sp = ps[o], mq = qm[o], sq = qs[o];
      const float ratio = sq / sp, vr = ratio * ratio;
      const float dm = (mq - mp) / sp, t1 = dm * dm;
      kl = 0.5f * (vr + t1 - 1.f - logf(vr));
      const float isp2 = 1.f / (sp * sp);
      gqm = (mq - mp) * isp2;
      gpm = -gqm;
      gqs = -1.f / sq + sq * isp2;
      gps = 1.f / sp - (sq * sq + (mq - mp) * (mq - mp)) * isp2 / sp;
    }
    const float klrow = wave_sum(kl);
    float lb = 0.f;  // log_beta of this row = tasks[row] @ log_beta (repo_mt.py:89)
    for (int i = 0; i < C; ++i) lb = fmaf(tasks[(size_t)row * C + i], log_beta[i], lb);
    const float beta = expf(lb), viol = klrow - target_kl;
    const float wp = beta * alpha * scale, wq = beta * (1.f - alpha) * scale;
    if (lane == 0) {
      acc[0] += klrow;
      acc[1] = fmaf(beta, viol, acc[1]);
      acc[2] = fmaf(lb, viol, acc[2]);
#pragma unroll
      for (int i = 0; i < kMtMaxTasks; ++i)
        if (i < C) acc[3 + i] = fmaf(tasks[(size_t)row * C + i], viol, acc[3 + i]);
    }
    if (lane < S) {
      if (dpm) dpm[o] = gpm * wp;
      if (dps) dps[o] = gps * wp;
      if (dqm) dqm[o] = gqm * wq;
      if (dqs) dqs[o] = gqs * wq;
    }
  }
#pragma unroll
  for (int i = 0; i < 3 + kMtMaxTasks; ++i) {
    if (i < 3 + C) {  // (uniform)
      const float s = block_sum(acc[i], red);
      if (threadIdx.x == 0) parts[i * gridDim.x + blockIdx.x] = s;
      __syncthreads();
    }
  }
}

__global__ void mt_final_sum_kernel(const float* __restrict__ parts, int n, int nvals, float* __restrict__ out) {
  __shared__ float red[16];
  for (int v = 0; v < nvals; ++v) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += parts[v * n + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[v] = s;
    __syncthreads();
  }
}

// ------------------------------------------------------------------ dual step on the per-task log_beta vector
// sums = the (3 + C) sums above over the GLOBAL batch.  beta_loss = -mean(lb_row * viol_row)  =>
// d / d log_beta[i] = -sum_rows tasks[row][i] * viol_row / rows; one torch.optim.Adam step on the C-vector.
// scalars_out: [kl_div, kl_loss, beta_loss, beta_0 .. beta_{C-1} AFTER the step] (repo_mt.py:100-112).
__global__ void dual_step_tasks_kernel(int C, float* __restrict__ log_beta, float* __restrict__ m,
                                       float* __restrict__ v, const float* __restrict__ sums, float inv_rows, float lr,
                                       float b1, float b2, float eps, float bc1, float bc2_sqrt, int apply,
                                       float* __restrict__ scalars_out, const unsigned* __restrict__ skip) {
  const int i = threadIdx.x;
  if (blockIdx.x != 0) return;
  if (skip && *skip) apply = 0;   // a faulted update: see repo_clip_adam
  if (i == 0) {
    scalars_out[0] = sums[0] * inv_rows;
    scalars_out[1] = sums[1] * inv_rows;
    scalars_out[2] = -sums[2] * inv_rows;
  }
  if (i < C) {
    const float g = -sums[3 + i] * inv_rows;
    float nlb = log_beta[i];
    if (apply) {
      const float mm = b1 * m[i] + (1.f - b1) * g;
      const float vv = b2 * v[i] + (1.f - b2) * g * g;
      m[i] = mm;
      v[i] = vv;
      nlb -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
      log_beta[i] = nlb;
    }
    scalars_out[3 + i] = expf(nlb);
  }
}

}  // namespace repo

using namespace repo;

static bool film_args_ok(int64_t nimg, int64_t C, int64_t P, int64_t ld, int64_t goff, int64_t boff) {
  return nimg > 0 && C > 0 && P > 0 && nimg * C * P < kMaxIdx * 4L && goff >= 0 && boff >= 0 && goff + C <= ld &&
         boff + C <= ld && (goff + C <= boff || boff + C <= goff);
}

extern "C" int repo_film_fwd(int64_t nimg, int64_t C, int64_t P, const float* y, const float* film, int64_t ldfilm,
                             int64_t gamma_off, int64_t beta_off, float* out, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(film_args_ok(nimg, C, P, ldfilm, gamma_off, beta_off), REPO_E_SHAPE);
  REPO_REQUIRE(y && film && out, REPO_E_BADARG);
  const long planes = nimg * C;
  long blocks = P >= 64 ? (planes + 3) / 4 : (planes * P + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(film_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, planes, (int)C, (int)P, y, film,
                     (int)ldfilm, (int)gamma_off, (int)beta_off, out);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

extern "C" int repo_film_bwd(int64_t nimg, int64_t C, int64_t P, const float* dh, const float* y, const float* film,
                             int64_t ldfilm, int64_t gamma_off, int64_t beta_off, float* dy, float* dfilm,
                             hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(film_args_ok(nimg, C, P, ldfilm, gamma_off, beta_off), REPO_E_SHAPE);
  REPO_REQUIRE(dh && y && film && dy && dfilm, REPO_E_BADARG);
  const long planes = nimg * C;
  long blocks = (planes + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(film_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, planes, (int)C, (int)P, dh, y,
                     film, (int)ldfilm, (int)gamma_off, (int)beta_off, dy, dfilm, (unsigned*)nullptr, 0u);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

extern "C" int repo_film_bwd_h(int64_t nimg, int64_t C, int64_t P, const float* dh, const float* h, const float* film,
                               int64_t ldfilm, int64_t gamma_off, int64_t beta_off, float* dy, float* dfilm,
                               int conv_kind, const int64_t* geo, const void* x, int x_is_u8, const float* w,
                               const float* bias, unsigned* gated_epoch, unsigned epoch, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(film_args_ok(nimg, C, P, ldfilm, gamma_off, beta_off), REPO_E_SHAPE);
  REPO_REQUIRE(dh && h && film && dy && dfilm, REPO_E_BADARG);
  REPO_REQUIRE(conv_kind >= 0 && conv_kind <= 3 && (!gated_epoch || epoch != 0), REPO_E_BADARG);
  FilmExact ex{};
  ex.kind = conv_kind;
  if (conv_kind) {
    REPO_REQUIRE(geo && x && w, REPO_E_BADARG);
    ex.x = x; ex.x_u8 = x_is_u8 ? 1 : 0; ex.w = w; ex.bias = bias;
    if (conv_kind == 3) {
      REPO_REQUIRE(geo[0] > 0 && geo[0] < (1 << 20) && !x_is_u8, REPO_E_SHAPE);
      ex.K = (int)geo[0];
      ex.KS = geo[1] ? 1 : 0;   // (kind 3 only: the bias is per output element, not per channel)
    } else {
      const int64_t CB = geo[0], CS = geo[1], HB = geo[2], KS = geo[3];
      REPO_REQUIRE(CB > 0 && CS > 0 && KS > 0 && HB >= KS && CB < 4096 && CS < 4096 && HB < 4096, REPO_E_SHAPE);
      const int64_t HS = (HB - KS) / 2 + 1;
      ex.CB = (int)CB; ex.CS = (int)CS; ex.HB = (int)HB; ex.HS = (int)HS; ex.KS = (int)KS;
      // the plane the FiLM acts on is the layer's OUTPUT: a small channel for the convolution, a big one for its transpose
      REPO_REQUIRE(conv_kind == 1 ? (C == CS && P == HS * HS) : (C == CB && P == HB * HB && !x_is_u8), REPO_E_SHAPE);
    }
  }
  const long planes = nimg * C;
  long blocks = (planes + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(film_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, planes, (int)C, (int)P, dh, h,
                     film, (int)ldfilm, (int)gamma_off, (int)beta_off, dy, dfilm, conv_kind ? gated_epoch : nullptr, epoch);
  REPO_CHECK_LAUNCH();
  if (conv_kind) {   // the gated-off planes' gamma gradients, exactly (every block returns at once when there are none)
    long eb = planes < 4096 ? planes : 4096;
    hipLaunchKernelGGL(film_exact_kernel, dim3((unsigned)eb), dim3(256), 0, stream, planes, (int)C, (int)P, dh, film,
                       (int)ldfilm, (int)gamma_off, dfilm, ex, gated_epoch, epoch);
    REPO_CHECK_LAUNCH();
  }
  return REPO_OK;
}

extern "C" int repo_film_tables(int64_t nimg, int nlayers, const int* channels, const float* film, int64_t ldfilm,
                                float* tables, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(nimg > 0 && nlayers >= 1 && nlayers <= 4 && channels, REPO_E_SHAPE);
  REPO_REQUIRE(film && tables, REPO_E_BADARG);
  FilmTabArgs a{};
  a.nl = nlayers;
  int tot = 0;
  for (int l = 0; l < nlayers; ++l) {
    REPO_REQUIRE(channels[l] > 0, REPO_E_SHAPE);
    a.ch[l] = channels[l];
    a.off[l] = tot;
    tot += channels[l];
  }
  a.total = tot;
  REPO_REQUIRE(ldfilm >= 2 * tot && nimg * 2L * tot < kMaxIdx, REPO_E_SHAPE);
  long blocks = (nimg * 2L * tot + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(film_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (long)nimg, a, film, (int)ldfilm, tables);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

extern "C" size_t repo_kl_balance_tasks_workspace_bytes(void) { return (3 + kMtMaxTasks) * kMtRedBlocks * sizeof(float); }

extern "C" int repo_kl_balance_tasks(int64_t rows, int64_t S, int64_t C, const float* pm, const float* ps,
                                     const float* qm, const float* qs, float alpha, const float* log_beta,
                                     const float* tasks, float target_kl, float scale, float* dpm, float* dps,
                                     float* dqm, float* dqs, float* sums, void* ws, size_t ws_bytes,
                                     hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(rows > 0 && S > 0 && S <= 64 && rows * S < kMaxIdx && C >= 1 && C <= kMtMaxTasks, REPO_E_SHAPE);
  REPO_REQUIRE(pm && ps && qm && qs && log_beta && tasks && sums, REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_kl_balance_tasks_workspace_bytes(), REPO_E_WS_TOO_SMALL);
  long blocks = (rows + 3) / 4;
  if (blocks > kMtRedBlocks) blocks = kMtRedBlocks;
  hipLaunchKernelGGL(kl_tasks_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (int)rows, (int)S, (int)C, pm, ps, qm,
                     qs, alpha, log_beta, tasks, target_kl, scale, dpm, dps, dqm, dqs, (float*)ws);
  REPO_CHECK_LAUNCH();
  hipLaunchKernelGGL(mt_final_sum_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, (int)blocks, (int)(3 + C),
                     sums);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

extern "C" int repo_dual_step_tasks(int64_t C, float* log_beta, float* exp_avg, float* exp_avg_sq, const float* sums,
                                    int64_t rows, float lr, float beta1, float beta2, float eps, int64_t step,
                                    int apply, float* scalars_out, const unsigned* skip_if_nonzero,
                                    hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(C >= 1 && C <= kMtMaxTasks && rows > 0 && step >= 1, REPO_E_SHAPE);
  REPO_REQUIRE(log_beta && exp_avg && exp_avg_sq && sums && scalars_out, REPO_E_BADARG);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(dual_step_tasks_kernel, dim3(1), dim3(64), 0, stream, (int)C, log_beta, exp_avg, exp_avg_sq, sums,
                     (float)(1.0 / (double)rows), lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), apply, scalars_out,
                     skip_if_nonzero);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}
