// Flat-buffer optimiser: global gradient norm + fused clip_grad_norm_ and Adam.
// Replaces nn.utils.clip_grad_norm_ + torch.optim.Adam.step over 40 / 10 / 8 tensors
// (algorithms/repo/repo.py:87-90, dreamer.py:356-359,370-373) with two HBM-bound passes
// over one contiguous parameter / gradient / moment buffer.
#include "common.h"

namespace repo {

__global__ __launch_bounds__(256) void sqnorm_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ parts) {
  __shared__ float red[16];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (int64_t j = i; j < n; ++j) acc += g[j] * g[j];
    }
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = s;
}

__global__ void sqnorm_final_kernel(const float* __restrict__ parts, int n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += parts[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out = s;
}

// clip_coef = min(1, max_norm / (sqrt(sqnorm) + 1e-6)); g *= clip_coef; Adam (no weight decay, no amsgrad)
__global__ __launch_bounds__(256) void clip_adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ sqnorm, float max_norm, float lr_bc1,
                                                        float b1, float b2, float eps, float inv_bc2_sqrt,
                                                        const unsigned* __restrict__ skip) {
  if (skip && *skip) return;   // a faulted update (scan timeout: NaN gradients) must not touch the model
  float coef = 1.f;
  if (sqnorm) {
    coef = max_norm / (sqrtf(*sqnorm) + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gg = g[i] * coef;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] = p[i] - lr_bc1 * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
  }
}

}  // namespace repo

using namespace repo;

// a reduction workspace (include/repo_hip.h, "losses and regularisers"): its 256-byte header belongs to the single-launch
// reductions' ticket and stays untouched (zero) here -- this grid (up to 1024 blocks) keeps its follow-up launch
extern "C" size_t repo_grad_sqnorm_workspace_bytes(void) { return kRedHeaderBytes + 1024 * sizeof(float); }

extern "C" int repo_grad_sqnorm(int64_t n, const float* g, float* sqnorm, void* ws, size_t ws_bytes,
                                hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(n > 0, REPO_E_SHAPE);
  REPO_REQUIRE(g && sqnorm, REPO_E_BADARG);
  REPO_REQUIRE(((uintptr_t)g & 15) == 0, REPO_E_ALIGN);
  REPO_REQUIRE(ws && ws_bytes >= repo_grad_sqnorm_workspace_bytes(), REPO_E_WS_TOO_SMALL);
  long blocks = (n + 4095) / 4096;
  if (blocks > 1024) blocks = 1024;
  float* parts = (float*)((char*)ws + kRedHeaderBytes);
  hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, g, parts);
  REPO_CHECK_LAUNCH();
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)parts, (int)blocks, sqnorm);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

extern "C" int repo_clip_adam(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              const float* sqnorm, float max_norm, float lr, float beta1, float beta2, float eps,
                              int64_t step, const unsigned* skip_if_nonzero, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(n > 0 && step >= 1, REPO_E_SHAPE);
  REPO_REQUIRE(params && grads && exp_avg && exp_avg_sq, REPO_E_BADARG);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, params, grads, exp_avg,
                     exp_avg_sq, sqnorm, max_norm, (float)((double)lr / bc1), beta1, beta2, eps,
                     (float)(1.0 / sqrt(bc2)), skip_if_nonzero);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}
