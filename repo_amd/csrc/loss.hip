// Loss / regulariser reductions of the update: KL balance + Lagrangian dual, masked
// Gaussian NLL of the scalar heads, Monte-Carlo tanh-Normal entropy, Normal entropy,
// lambda-returns.  All are HBM-bound streaming kernels: coalesced loads, wave64 shuffle
// reductions, one partial per workgroup and a fixed-order final sum (bitwise reproducible;
// no float atomics) taken by the launch's LAST block (common.h, last_block_finishes: one launch
// per reduction since round 6, same bits as the follow-up launch it replaces).
#include "common.h"

namespace repo {

constexpr int kRedBlocks = 1024;  // max partials per reduction

__global__ void final_sum_kernel(const float* __restrict__ parts, int n, int nvals, float* __restrict__ out) {
  // parts is [nvals][n]; out[v] = sum_i parts[v][i]
  __shared__ float red[16];
  for (int v = 0; v < nvals; ++v) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += parts[v * n + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[v] = s;
  }
}

// ------------------------------------------------------------------ KL(q || p) of diagonal Gaussians
// mode 0 (RePo, repo.py:64-83): out = sum_rows KL_row; gradients of
//   beta * (alpha * KL(sg q || p) + (1-alpha) * KL(q || sg p)) * scale
// mode 1 (Dreamer, dreamer.py:278-282): out = sum_rows max(KL_row, free_nats); gradients of
//   max(KL_row, free_nats) * scale on both sides.
// One wave per row (S <= 64 lanes active).
__global__ __launch_bounds__(256) void kl_kernel(int rows, int S, const float* __restrict__ pm,
                                                 const float* __restrict__ ps, const float* __restrict__ qm,
                                                 const float* __restrict__ qs, int mode, float alpha,
                                                 const float* __restrict__ log_beta, float free_nats, float scale,
                                                 float* __restrict__ dpm, float* __restrict__ dps,
                                                 float* __restrict__ dqm, float* __restrict__ dqs,
                                                 float* __restrict__ parts, float* __restrict__ red_out,
    unsigned* __restrict__ ticket) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float beta = (mode == 0 && log_beta) ? expf(*log_beta) : 1.f;
  float acc = 0.f;
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
    float wp, wq;
    if (mode == 0) {
      wp = beta * alpha * scale;
      wq = beta * (1.f - alpha) * scale;
      if (lane == 0) acc += klrow;
    } else {
      const bool active = klrow > free_nats;
      wp = wq = active ? scale : 0.f;
      if (lane == 0) acc += active ? klrow : free_nats;
    }
    if (lane < S) {
      if (dpm) dpm[o] = gpm * wp;
      if (dps) dps[o] = gps * wp;
      if (dqm) dqm[o] = gqm * wq;
      if (dqs) dqs[o] = gqs * wq;
    }
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = s;
  last_block_finishes(parts, 1, red_out, ticket, red);
}

// ------------------------------------------------------------------ dual variable (repo.py:83,93-105)
// scalars_out: [kl_div, kl_loss, beta_loss, beta_after]
__global__ void dual_step_kernel(float* __restrict__ log_beta, float* __restrict__ m, float* __restrict__ v,
                                 const float* __restrict__ kl_sum, float inv_rows, float target_kl, float lr, float b1,
                                 float b2, float eps, float bc1, float bc2_sqrt, int apply,
                                 float* __restrict__ scalars_out, const unsigned* __restrict__ skip) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (skip && *skip) apply = 0;   // a faulted update: log the (NaN) scalars, leave the dual variable alone
  const float lb = *log_beta;
  const float kl_div = *kl_sum * inv_rows;
  const float viol = kl_div - target_kl;
  const float g = -viol;  // d(-log_beta * viol)/d log_beta
  scalars_out[0] = kl_div;
  scalars_out[1] = expf(lb) * viol;
  scalars_out[2] = -lb * viol;
  float nlb = lb;
  if (apply) {
    const float mm = b1 * (*m) + (1.f - b1) * g;
    const float vv = b2 * (*v) + (1.f - b2) * g * g;
    *m = mm;
    *v = vv;
    nlb = lb - (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
    *log_beta = nlb;
  }
  scalars_out[3] = expf(nlb);
}

// ------------------------------------------------------------------ masked unit-variance NLL of a scalar head
// parts[0][blk] = sum 0.5*(p-t)^2*mask ; parts[1][blk] = sum mask ; dpred = (p-t)*mask*scale
__global__ __launch_bounds__(256) void scalar_nll_kernel(int n, const float* __restrict__ pred,
                                                         const float* __restrict__ target,
                                                         const float* __restrict__ mask, float scale,
                                                         float* __restrict__ dpred, float* __restrict__ parts, float* __restrict__ red_out,
    unsigned* __restrict__ ticket) {
  __shared__ float red[16];
  float a = 0.f, msum = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float mk = mask ? mask[i] : 1.f;
    const float d = pred[i] - target[i];
    a += 0.5f * d * d * mk;
    msum += mk;
    if (dpred) dpred[i] = d * mk * scale;
  }
  const float s0 = block_sum(a, red);
  const float s1 = block_sum(msum, red);
  if (threadIdx.x == 0) {
    parts[blockIdx.x] = s0;
    parts[gridDim.x + blockIdx.x] = s1;
  }
  last_block_finishes(parts, 2, red_out, ticket, red);
}

// ------------------------------------------------------------------ Monte-Carlo entropy of tanh(Normal)
// SampleDist.entropy (models/utils.py:160-163) over TanhBijector (models/utils.py:126-134):
//   u = mean + std*eps ; y = tanh(u) ; x = atanh(clamp(y, +-0.99999994)) (inverse recomputed)
//   logp = -0.5((x-mean)/std)^2 - log std - 0.5 log 2pi - 2(log2 - x - softplus(-2x))
// element e = row*A + a; eps is (NS, n) with n = rows*A, or null: then sample s of element e is normal number
// offset + e*NS + s of the Philox stream (SAMPLE-fastest, so a thread's NS draws are NS/4 counter blocks).  Writes
//   parts[blk]   = sum_e ( -(1/NS) sum_s logp_{s,e} )         (sum over rows of the row entropy)
//   dmean/dstd[e] = gscale * d(entropy_sum)/d(mean,std)[e]
__global__ __launch_bounds__(256) void tanh_normal_entropy_kernel(int n, int NS, const float* __restrict__ mean,
                                                                  const float* __restrict__ stdv,
                                                                  const float* __restrict__ eps, uint64_t nseed,
                                                                  uint64_t noffset, float gscale,
                                                                  float* __restrict__ dmean, float* __restrict__ dstd,
                                                                  float* __restrict__ parts, float* __restrict__ red_out,
    unsigned* __restrict__ ticket) {
  __shared__ float red[16];
  const float kClamp = 0.99999994f;
  const float kLog2 = 0.69314718055994531f;
  float acc = 0.f;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const float mu = mean[e], sd = stdv[e];
    const float isd = 1.f / sd, isd2 = isd * isd, lsd = __logf(sd);
    float slog = 0.f, gmu = 0.f, gsd = 0.f;
    float z[4];
    const uint64_t first = noffset + (uint64_t)e * NS;  // this element's first normal
    for (int s = 0; s < NS; ++s) {
      float ep;
      if (eps) {
        ep = eps[(size_t)s * n + e];
      } else {
        const uint64_t i = first + s;
        if (s == 0 || (i & 3) == 0) philox_normal4(nseed, i >> 2, z);
        const int l = (int)(i & 3);
        ep = l == 0 ? z[0] : l == 1 ? z[1] : l == 2 ? z[2] : z[3];
      }
      const float u = fmaf(sd, ep, mu);
      // On the hardware transcendentals (v_exp / v_rcp / v_log, ~1 ulp each) instead of libm's tanhf / atanhf call
      // sequences, which were 3/4 of this kernel's instructions (162 -> 60 us at 100 x 205800 samples), keeping the
      // reference's rounding point: y = tanh(u) is formed as an fp32 value, clamped, and the inverse is recomputed
      // FROM it.  With e = exp(-2|u|): |y| = 1 - 2e / (1 + e); with r = (1 - |yc|) / (1 + |yc|) = exp(-2|x|):
      //   |x| = atanh(|yc|) = -log(r) / 2      and      log 2 - x - softplus(-2x) = log 2 - |x| - log(1 + r)
      // (both signs of x: softplus(-2x) = log(1 + r) + 2 max(-x, 0)).
      // (|y| as 1 - 2e/(1 + e): the small term carries its full relative accuracy, so the SUBTRACTION rounds |y| like a
      // correctly rounded tanh does where it matters -- within a few ulps of 1, where atanh amplifies one ulp of y to
      // 0.2-0.35 in x)
      const float e2 = __expf(-2.f * fabsf(u));
      const float ya = 1.f - 2.f * e2 * rcp_fast(1.f + e2);
      const bool pass = ya <= kClamp;  // clamp passes gradient inside (inclusive) the bounds
      const float ya_c = fminf(ya, kClamp);
      const float r = (1.f - ya_c) * rcp_fast(1.f + ya_c);
      const float ax = -0.5f * __logf(r);
      const float x = copysignf(ax, u), yc = copysignf(ya_c, u);
      const float dlt = x - mu;
      slog += -0.5f * dlt * dlt * isd2 - lsd - 0.5f * kLog2Pi - 2.f * (kLog2 - ax - __logf(1.f + r));
      // dlogp/dx (through the recomputed inverse) times dx/du (1 inside the clamp, else 0)
      // d/dx of the log-det term is 2 tanh(x), and tanh(x) = tanh(atanh(yc)) = yc
      const float via_x = pass ? (-dlt * isd2 + 2.f * yc) : 0.f;
      gmu += dlt * isd2 + via_x;
      gsd += dlt * dlt * isd2 * isd - isd + via_x * ep;
    }
    const float inv = 1.f / (float)NS;
    acc += -slog * inv;
    if (dmean) dmean[e] = -gmu * inv * gscale;
    if (dstd) dstd[e] = -gsd * inv * gscale;
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = s;
  last_block_finishes(parts, 1, red_out, ticket, red);
}

// sum over elements of (0.5 + 0.5 log 2pi + log std); dstd = gscale / std   (dreamer.py:327-328)
__global__ __launch_bounds__(256) void normal_entropy_kernel(int n, const float* __restrict__ stdv, float gscale,
                                                             float* __restrict__ dstd, float* __restrict__ parts, float* __restrict__ red_out,
    unsigned* __restrict__ ticket) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float sd = stdv[i];
    acc += 0.5f + 0.5f * kLog2Pi + logf(sd);
    if (dstd) dstd[i] = gscale / sd;
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = s;
  last_block_finishes(parts, 1, red_out, ticket, red);
}

// ------------------------------------------------------------------ lambda-returns (common/utils.py:61-71)
// r, v are (Hm, N) (reward / value predictions on the imagined steps); uses r[:-1], v[:-1],
// bootstrap v[-1], constant discount gamma.  One thread per column, serial over the Hm-1 steps.
//   returns (Hm-1, N);  parts[blk] = sum returns
//   dr, dv (Hm, N): gradient of gret * sum(returns)      (gret = -1/((Hm-1)N) for -mean(returns))
__global__ __launch_bounds__(256) void lambda_return_kernel(int Hm, int N, const float* __restrict__ r,
                                                            const float* __restrict__ v, float gamma, float lam,
                                                            float gret, float* __restrict__ returns,
                                                            float* __restrict__ dr, float* __restrict__ dv,
                                                            float* __restrict__ parts, float* __restrict__ red_out,
    unsigned* __restrict__ ticket) {
  __shared__ float red[16];
  float acc = 0.f;
  const int L = Hm - 1;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float boot = v[(size_t)L * N + n];
    float last = boot;
    for (int t = L - 1; t >= 0; --t) {
      const float nextv = (t == L - 1) ? boot : v[(size_t)(t + 1) * N + n];
      const float inp = r[(size_t)t * N + n] + gamma * nextv * (1.f - lam);
      last = inp + gamma * lam * last;
      returns[(size_t)t * N + n] = last;
      acc += last;
    }
    if (dr && dv) {
      // G_t = gret + gamma*lam*G_{t-1}: total gradient reaching inputs_t
      float G = 0.f;
      dv[n] = 0.f;  // values[0] is never read
      for (int t = 0; t < L; ++t) {
        G = gret + gamma * lam * G;
        dr[(size_t)t * N + n] = G;
        const float gnext = gamma * (1.f - lam) * G;
        if (t < L - 1) dv[(size_t)(t + 1) * N + n] = gnext;
        else dv[(size_t)L * N + n] = gnext + gamma * lam * G;  // bootstrap: next_value and the scan seed
      }
      dr[(size_t)L * N + n] = 0.f;
    }
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = s;
  last_block_finishes(parts, 1, red_out, ticket, red);
}


// ------------------------------------------------------------------ SampleDist.mode (models/utils.py:149-158)
// action[row] = the sample with the highest log-probability among NS draws; eps (NS, rows, A).
// One wave per row: lanes stride over the NS samples (this runs on ONE row per environment step; a thread
// per row spent 316 us walking 100 samples serially).  argmax with torch's tie rule (first maximum).
__global__ __launch_bounds__(256) void tanh_normal_mode_kernel(int rows, int A, int NS, const float* __restrict__ mean,
                                                               const float* __restrict__ stdv,
                                                               const float* __restrict__ eps,
                                                               float* __restrict__ action) {
  const float kClamp = 0.99999994f;
  const float kLog2 = 0.69314718055994531f;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int s = lane; s < NS; s += 64) {
      float lp = 0.f;
      for (int a = 0; a < A; ++a) {
        const float mu = mean[(size_t)row * A + a], sd = stdv[(size_t)row * A + a];
        const float y = tanhf(fmaf(sd, eps[((size_t)s * rows + row) * A + a], mu));
        const float x = atanhf(fminf(fmaxf(y, -kClamp), kClamp));
        const float d = (x - mu) / sd;
        lp += -0.5f * d * d - logf(sd) - 0.5f * kLog2Pi - 2.f * (kLog2 - x - softplus(-2.f * x));
      }
      if (lp > best) {  // ascending s within a lane: the first maximum of this lane's samples
        best = lp;
        bi = s;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (bi == 0x7fffffff) bi = 0;  // every log-probability was NaN / -inf: torch.argmax returns 0 for all-equal -inf
    for (int a = lane; a < A; a += 64) {
      const float mu = mean[(size_t)row * A + a], sd = stdv[(size_t)row * A + a];
      action[(size_t)row * A + a] = tanhf(fmaf(sd, eps[((size_t)bi * rows + row) * A + a], mu));
    }
  }
}

// ------------------------------------------------------------------ TIA: mask head + blend + pixel NLL
// tia.py:123-133 (reference): the task and distractor decoders each emit 6 channels (3 "recon" + 3 "mask");
//   m = sigmoid(b + sum_c w[c] t_mask[c] + w[3+c] d_mask[c])          (mask_head = Conv2d(6, 1, 1) + Sigmoid)
//   recon[c] = t_recon[c] m + d_recon[c] (1 - m);  loss = sum 0.5 (recon - target)^2
// One streaming pass: reads both 6-channel outputs and the target, writes the gradients with respect to both outputs
// (may alias the inputs: every thread reads its 4 pixels before it writes them) and 8 partials per workgroup:
// the loss and d loss / d (w[0..5], b).  Thread = 4 consecutive pixels of one frame (16-byte accesses).
template <class TgtT>
__global__ __launch_bounds__(256) void tia_blend_nll_kernel(int n4, int pix4, const float* __restrict__ wb,
                                                            const float* t_out, const float* d_out,
                                                            const TgtT* __restrict__ target, float gscale, float* dt_out,
                                                            float* dd_out, float* __restrict__ recon_out,
                                                            float* __restrict__ parts, float* __restrict__ red_out,
    unsigned* __restrict__ ticket) {
  __shared__ float red[16];
  float w[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) w[c] = wb[c];
  const float b = wb[6];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const int img = i / pix4, q = i % pix4;
    const size_t base6 = ((size_t)img * 6 * pix4 + q) * 4;  // channel c at + c * 4 * pix4
    const size_t base3 = ((size_t)img * 3 * pix4 + q) * 4;
    float4 t[6], d[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      t[c] = *reinterpret_cast<const float4*>(t_out + base6 + (size_t)c * 4 * pix4);
      d[c] = *reinterpret_cast<const float4*>(d_out + base6 + (size_t)c * 4 * pix4);
    }
    float tg[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) tg[c][e] = load_as_float(target, (unsigned)(base3 + (size_t)c * 4 * pix4 + e));
    float4 gt[6], gd[6], rc[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      auto el = [e](float4& v) -> float& { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
      float z = b;
#pragma unroll
      for (int c = 0; c < 3; ++c) z += w[c] * el(t[3 + c]) + w[3 + c] * el(d[3 + c]);
      const float m = 1.f / (1.f + expf(-z));
      float dm = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float tr = el(t[c]), dr = el(d[c]);
        const float r = tr * m + dr * (1.f - m);
        const float df = r - tg[c][e];
        acc[0] += 0.5f * df * df;
        const float g = df * gscale;
        el(gt[c]) = g * m;
        el(gd[c]) = g * (1.f - m);
        el(rc[c]) = r;
        dm += g * (tr - dr);
      }
      const float dz = dm * m * (1.f - m);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        el(gt[3 + c]) = dz * w[c];
        el(gd[3 + c]) = dz * w[3 + c];
        acc[1 + c] += dz * el(t[3 + c]);
        acc[4 + c] += dz * el(d[3 + c]);
      }
      acc[7] += dz;
    }
    if (dt_out) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        *reinterpret_cast<float4*>(dt_out + base6 + (size_t)c * 4 * pix4) = gt[c];
        *reinterpret_cast<float4*>(dd_out + base6 + (size_t)c * 4 * pix4) = gd[c];
      }
    }
    if (recon_out) {
#pragma unroll
      for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(recon_out + base3 + (size_t)c * 4 * pix4) = rc[c];
    }
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const float sv = block_sum(acc[v], red);
    if (threadIdx.x == 0) parts[v * gridDim.x + blockIdx.x] = sv;
  }
  last_block_finishes(parts, 8, red_out, ticket, red);
}

static inline int red_blocks(long n, int per_block) {
  long b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > kRedBlocks) b = kRedBlocks;
  return (int)b;
}

}  // namespace repo

using namespace repo;

// reduction workspace: [zeroed header: the ticket word of last_block_finishes | partials].  A grid of at most
// kLastBlockMaxGrid blocks finishes its own sum (ticket given); a larger one gets the follow-up launch (ticket null).
static inline float* red_parts(void* ws) { return (float*)((char*)ws + kRedHeaderBytes); }
static inline unsigned* red_ticket(void* ws, int blocks) { return blocks <= kLastBlockMaxGrid ? (unsigned*)ws : nullptr; }
static int red_finish(void* ws, int blocks, int nvals, float* out, hipStream_t s) {
  if (blocks <= kLastBlockMaxGrid) return REPO_OK;   // the launch's last block has written `out`
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, red_parts(ws), blocks, nvals, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}
extern "C" size_t repo_reduce_workspace_bytes(void) { return kRedHeaderBytes + 2 * kRedBlocks * sizeof(float); }

extern "C" int repo_kl_balance(int64_t rows, int64_t S, const float* pm, const float* ps, const float* qm,
                               const float* qs, int mode, float alpha, const float* log_beta, float free_nats,
                               float scale, float* dpm, float* dps, float* dqm, float* dqs, float* kl_sum, void* ws,
                               size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(rows > 0 && S > 0 && S <= 64 && rows * S < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(pm && ps && qm && qs && kl_sum && (mode == 0 || mode == 1), REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_reduce_workspace_bytes(), REPO_E_WS_TOO_SMALL);
  const int blocks = red_blocks(rows, 40);   // <= 64 blocks up to 2560 rows: the launch finishes its own sum (common.h)
  hipLaunchKernelGGL(kl_kernel, dim3(blocks), dim3(256), 0, stream, (int)rows, (int)S, pm, ps, qm, qs, mode, alpha,
                     log_beta, free_nats, scale, dpm, dps, dqm, dqs, red_parts(ws), kl_sum, red_ticket(ws, blocks));
  REPO_CHECK_LAUNCH();
  return red_finish(ws, blocks, 1, kl_sum, stream);
}

extern "C" int repo_dual_step(float* log_beta, float* exp_avg, float* exp_avg_sq, const float* kl_sum, int64_t rows,
                              float target_kl, float lr, float beta1, float beta2, float eps, int64_t step, int apply,
                              float* scalars_out, const unsigned* skip_if_nonzero, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(log_beta && exp_avg && exp_avg_sq && kl_sum && scalars_out && rows > 0 && step >= 1, REPO_E_BADARG);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(dual_step_kernel, dim3(1), dim3(64), 0, stream, log_beta, exp_avg, exp_avg_sq, kl_sum,
                     (float)(1.0 / (double)rows), target_kl, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), apply,
                     scalars_out, skip_if_nonzero);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

extern "C" int repo_scalar_nll(int64_t n, const float* pred, const float* target, const float* mask, float scale,
                               float* dpred, float* sums2, void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(n > 0 && n < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(pred && target && sums2, REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_reduce_workspace_bytes(), REPO_E_WS_TOO_SMALL);
  const int blocks = red_blocks(n, 1024);
  hipLaunchKernelGGL(scalar_nll_kernel, dim3(blocks), dim3(256), 0, stream, (int)n, pred, target, mask, scale, dpred,
                     red_parts(ws), sums2, red_ticket(ws, blocks));
  REPO_CHECK_LAUNCH();
  return red_finish(ws, blocks, 2, sums2, stream);
}

extern "C" size_t repo_tia_blend_nll_workspace_bytes(void) { return kRedHeaderBytes + 8 * kRedBlocks * sizeof(float); }

extern "C" int repo_tia_blend_nll(int64_t nimg, int64_t pixels, const float* t_out, const float* d_out,
                                  const float* mask_wb, const void* target, int target_is_u8, float grad_scale,
                                  float* dt_out, float* dd_out, float* recon, float* sums8, void* ws, size_t ws_bytes,
                                  hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(nimg > 0 && pixels > 0 && pixels % 4 == 0 && nimg * pixels * 6 < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(t_out && d_out && mask_wb && target && sums8 && ((dt_out == nullptr) == (dd_out == nullptr)),
               REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_tia_blend_nll_workspace_bytes(), REPO_E_WS_TOO_SMALL);
  const long n4 = nimg * pixels / 4;
  const int blocks = red_blocks(n4, 512);
  if (target_is_u8)
    hipLaunchKernelGGL(tia_blend_nll_kernel<uint8_t>, dim3(blocks), dim3(256), 0, stream, (int)n4, (int)(pixels / 4),
                       mask_wb, t_out, d_out, (const uint8_t*)target, grad_scale, dt_out, dd_out, recon, red_parts(ws), sums8,
                       red_ticket(ws, blocks));
  else
    hipLaunchKernelGGL(tia_blend_nll_kernel<float>, dim3(blocks), dim3(256), 0, stream, (int)n4, (int)(pixels / 4),
                       mask_wb, t_out, d_out, (const float*)target, grad_scale, dt_out, dd_out, recon, red_parts(ws), sums8, red_ticket(ws, blocks));
  REPO_CHECK_LAUNCH();
  return red_finish(ws, blocks, 8, sums8, stream);
}

extern "C" int repo_tanh_normal_entropy(int64_t rows, int64_t A, int64_t samples, const float* mean, const float* std,
                                        const float* eps, uint64_t noise_seed, uint64_t noise_offset, float gscale,
                                        float* dmean, float* dstd, float* ent_sum, void* ws, size_t ws_bytes,
                                        hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(rows > 0 && A > 0 && samples > 0 && rows * A < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(mean && std && ent_sum, REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_reduce_workspace_bytes(), REPO_E_WS_TOO_SMALL);
  const long n = rows * A;
  const int blocks = red_blocks(n, 256);
  hipLaunchKernelGGL(tanh_normal_entropy_kernel, dim3(blocks), dim3(256), 0, stream, (int)n, (int)samples, mean, std,
                     eps, noise_seed, noise_offset, gscale, dmean, dstd, red_parts(ws), ent_sum, red_ticket(ws, blocks));
  REPO_CHECK_LAUNCH();
  return red_finish(ws, blocks, 1, ent_sum, stream);
}

extern "C" int repo_normal_entropy(int64_t n, const float* std, float gscale, float* dstd, float* ent_sum, void* ws,
                                   size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(n > 0 && n < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(std && ent_sum, REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_reduce_workspace_bytes(), REPO_E_WS_TOO_SMALL);
  const int blocks = red_blocks(n, 1024);
  hipLaunchKernelGGL(normal_entropy_kernel, dim3(blocks), dim3(256), 0, stream, (int)n, std, gscale, dstd, red_parts(ws), ent_sum, red_ticket(ws, blocks));
  REPO_CHECK_LAUNCH();
  return red_finish(ws, blocks, 1, ent_sum, stream);
}

extern "C" int repo_lambda_return(int64_t Hm, int64_t N, const float* rewards, const float* values, float gamma,
                                  float lambda_, float gret, float* returns, float* drewards, float* dvalues,
                                  float* ret_sum, void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(Hm >= 2 && N > 0 && Hm * N < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(rewards && values && returns && ret_sum && ((drewards == nullptr) == (dvalues == nullptr)),
               REPO_E_BADARG);
  REPO_REQUIRE(ws && ws_bytes >= repo_reduce_workspace_bytes(), REPO_E_WS_TOO_SMALL);
  const int blocks = red_blocks(N, 256);
  hipLaunchKernelGGL(lambda_return_kernel, dim3(blocks), dim3(256), 0, stream, (int)Hm, (int)N, rewards, values, gamma,
                     lambda_, gret, returns, drewards, dvalues, red_parts(ws), ret_sum, red_ticket(ws, blocks));
  REPO_CHECK_LAUNCH();
  return red_finish(ws, blocks, 1, ret_sum, stream);
}

extern "C" int repo_tanh_normal_mode(int64_t rows, int64_t A, int64_t samples, const float* mean, const float* std,
                                     const float* eps, float* action, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(rows > 0 && A > 0 && samples > 0 && rows * A * samples < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(mean && std && eps && action, REPO_E_BADARG);
  const int blocks = (int)((rows + 3) / 4 > 4096 ? 4096 : (rows + 3) / 4);
  hipLaunchKernelGGL(tanh_normal_mode_kernel, dim3(blocks), dim3(256), 0, stream, (int)rows, (int)A, (int)samples, mean,
                     std, eps, action);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}
