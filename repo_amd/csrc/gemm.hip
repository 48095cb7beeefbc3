// Dense-layer GEMMs on the igemm tile engine: forward / backward-data (repo_gemm) and the
// split-K weight gradient (repo_gemm_wgrad).
#include <stdlib.h>

#include <atomic>

#include "bgemm.h"
#include "igemm.h"
#include "vgemm.h"

namespace repo {

// Test aid (repo_debug_bgemm): 0 keeps every product on the fp32-MFMA tile engines, for A/B runs in one process.
static thread_local int t_bgemm_enabled = 1;   // thread-local: see api.hip

template <bool TA, bool TB>
struct GemmOp {
  static constexpr bool A_KMAJOR = !TA;  // A[m][k]: k contiguous
  static constexpr bool B_KMAJOR = TB;   // B[n][k]: k contiguous
  const float* A;
  const float* B;
  const float* bias;
  const float* aux;
  float* C;
  int lda, ldb, ldc, ldaux, bias_div;
  int M_, N_, K_;
  int epi, accumulate;

  typedef int AM;
  typedef int AK;
  typedef int BN;
  typedef int BK;
  __device__ void init(int) {}
  __device__ int M() const { return M_; }
  __device__ int N() const { return N_; }
  __device__ int kbeg() const { return 0; }
  __device__ int kend() const { return K_; }
  __device__ AM a_m(int m) const { return TA ? m : m * lda; }
  __device__ AK a_k(int k) const { return TA ? k * lda : k; }
  __device__ float a(const AM& m, const AK& k) const { return A[(unsigned)(m + k)]; }
  __device__ BN b_n(int n) const { return TB ? n * ldb : n; }
  __device__ BK b_k(int k) const { return TB ? k : k * ldb; }
  __device__ float b(const BK& k, const BN& n) const { return B[(unsigned)(k + n)]; }
  __device__ void store_col(int mb, int n, const f32x16& acc, int M) {
    const float bv = bias ? bias[bias_div > 1 ? n / bias_div : n] : 0.f;   // bias_div < 0: one bias per output column (FiLM: header)
    float* c = C + mb * ldc + n;
    const float* ax = aux ? aux + mb * ldaux + n : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dm = (r & 3) + 8 * (r >> 2);
      if (mb + dm < M) {
        float v = acc[r] + bv;
        if (epi == REPO_EPI_ELU) v = elu(v);
        else if (epi == REPO_EPI_RELU) v = fmaxf(v, 0.f);
        else if (epi == REPO_EPI_MUL_DELU) v *= elu_grad_from_out(ax[dm * ldaux]);
        else if (epi == REPO_EPI_MUL_DRELU) v = ax[dm * ldaux] > 0.f ? v : 0.f;
        else if (epi == REPO_EPI_FILM_RELU) {   // row's FiLM table [scale (C) | shift (C)], C = ldaux / 2 (bgemm.h)
          const int ch = bias_div == 1 ? n : n / (bias_div < 0 ? -bias_div : bias_div);
          const float* tb = aux + (size_t)(mb + dm) * ldaux;
          v = fmaxf(fmaf(tb[ch], v, tb[(ldaux >> 1) + ch]), 0.f);
        }
        if (accumulate) v += c[dm * ldc];
        c[dm * ldc] = v;
      }
    }
  }
  __device__ void finish() {}
};

// The same product on the vector-load engine (vgemm.h): operands as raw-buffer 2-D arrays.
template <bool TA, bool TB, int VW_>
struct VGemmOp {
  static constexpr bool A_VK = !TA;  // A[m][k]: k contiguous
  static constexpr bool B_VK = TB;   // B[n][k]: k contiguous
  static constexpr int VW = VW_;
  Dense2D A, B;
  const float* bias;
  const float* aux;
  float* C;
  int ldc, ldaux, bias_div;
  int M_, N_, K_;
  int epi, accumulate;

  __device__ void init(int) {}
  __device__ int M() const { return M_; }
  __device__ int N() const { return N_; }
  __device__ int kbeg() const { return 0; }
  __device__ int kend() const { return K_; }
  template <class V>
  __device__ void fix_b(V&, int) const {}
  __device__ void store_col(int mb, int n, const f32x16& acc, int M) {
    const float bv = bias ? bias[bias_div > 1 ? n / bias_div : n] : 0.f;   // bias_div < 0: one bias per output column (FiLM: header)
    float* c = C + mb * ldc + n;
    const float* ax = aux ? aux + mb * ldaux + n : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dm = (r & 3) + 8 * (r >> 2);
      if (mb + dm < M) {
        float v = acc[r] + bv;
        if (epi == REPO_EPI_ELU) v = elu(v);
        else if (epi == REPO_EPI_RELU) v = fmaxf(v, 0.f);
        else if (epi == REPO_EPI_MUL_DELU) v *= elu_grad_from_out(ax[dm * ldaux]);
        else if (epi == REPO_EPI_MUL_DRELU) v = ax[dm * ldaux] > 0.f ? v : 0.f;
        else if (epi == REPO_EPI_FILM_RELU) {   // row's FiLM table [scale (C) | shift (C)], C = ldaux / 2 (bgemm.h)
          const int ch = bias_div == 1 ? n : n / (bias_div < 0 ? -bias_div : bias_div);
          const float* tb = aux + (size_t)(mb + dm) * ldaux;
          v = fmaxf(fmaf(tb[ch], v, tb[(ldaux >> 1) + ch]), 0.f);
        }
        if (accumulate) v += c[dm * ldc];
        c[dm * ldc] = v;
      }
    }
  }
  __device__ void finish() {}
};

// M <= 8 rows (the acting path: one frame per environment step): a tile engine spends its time in the
// 13-77 dependent K slices (14 us per 200x230 layer); here one wave owns an output column and its lanes
// split K (k-contiguous weights) or one thread owns a column and walks K with coalesced rows.
template <bool TB>
__global__ __launch_bounds__(256) void gemv_small_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                         const float* __restrict__ B, int ldb,
                                                         const float* __restrict__ bias, int bias_div,
                                                         float* __restrict__ C, int ldc, int epi,
                                                         const float* __restrict__ aux, int ldaux, int accumulate) {
  constexpr int MR = 8;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  int n;
  bool writer;
  if (TB) {  // B[n][k]: one wave per column, lanes over k
    const int lane = threadIdx.x & 63;
    n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    for (int k = lane; k < K; k += 64) {
      const float b = B[(size_t)n * ldb + k];
#pragma unroll
      for (int m = 0; m < MR; ++m)
        if (m < M) acc[m] = fmaf(A[(size_t)m * lda + k], b, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = wave_sum(acc[m]);
    writer = lane == 0;
  } else {  // B[k][n]: one thread per column
    n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < K; ++k) {
      const float b = B[(size_t)k * ldb + n];
#pragma unroll
      for (int m = 0; m < MR; ++m)
        if (m < M) acc[m] = fmaf(A[(size_t)m * lda + k], b, acc[m]);
    }
    writer = true;
  }
  if (!writer) return;
  const float bv = bias ? bias[bias_div > 1 ? n / bias_div : n] : 0.f;   // bias_div < 0: one bias per output column (FiLM: header)
#pragma unroll
  for (int m = 0; m < MR; ++m)
    if (m < M) {
      float v = acc[m] + bv;
      if (epi == REPO_EPI_ELU) v = elu(v);
      else if (epi == REPO_EPI_RELU) v = fmaxf(v, 0.f);
      else if (epi == REPO_EPI_MUL_DELU) v *= elu_grad_from_out(aux[(size_t)m * ldaux + n]);
      else if (epi == REPO_EPI_MUL_DRELU) v = aux[(size_t)m * ldaux + n] > 0.f ? v : 0.f;
      float* c = C + (size_t)m * ldc + n;
      if (accumulate) v += *c;
      *c = v;
    }
}

template <class Op>
static int vgemm_dispatch(const Op& op, long M, long N, hipStream_t s) {
  // large products: 128x128 tiles with ONE register staging set (148 VGPRs -> 3 waves per SIMD; the
  // two-set variant needs > 256 and drops to one wave per SIMD: 102 vs 124-131 TFLOP/s at 4096^3)
  const long t128 = ((M + 127) / 128) * ((N + 127) / 128);
  if (M >= 512 && N >= 512 && t128 >= 192) return launch_vgemm<T128x128s1>(op, M, N, 1, s);
  if (M <= 32) return launch_vgemm<T32x128>(op, M, N, 1, s);
  return launch_vgemm<T64x64>(op, M, N, 1, s);
}

template <bool TA, bool TB>
static int gemm_dispatch(const GemmOp<TA, TB>& op, long M, long N, hipStream_t s) {
  // pick the tile by how many workgroups the problem yields (256 CUs to fill)
  const long t128 = ((M + 127) / 128) * ((N + 127) / 128);
  if (M >= 512 && N >= 512 && t128 >= 192) return launch_igemm<T128x128>(op, M, N, 1, s);
  if (M <= 32) return launch_igemm<T32x128>(op, M, N, 1, s);
  return launch_igemm<T64x64>(op, M, N, 1, s);
}

// dW[n][k] = sum_m dY[m][n] X[m][k]; column K of the product is the bias gradient.  Split-K over row groups on the
// vector-load engine: both operands are contiguous along m'/n'.
struct VWgradOp {
  static constexpr bool A_VK = false;  // A(m'=n, k'=row) = dY[row][n]
  static constexpr bool B_VK = false;  // B(k'=row, n'=k) = X[row][k]
  static constexpr int VW = 4;
  Dense2D A, B;
  float* slab;  // [splits][N][K+1]
  int rows, N_, K_, rows_per_split;
  int z, kb, ke;

  __device__ void init(int zz) {
    z = zz;
    kb = zz * rows_per_split;
    ke = min(rows, kb + rows_per_split);
  }
  __device__ int M() const { return N_; }
  __device__ int N() const { return K_ + 1; }
  __device__ int kbeg() const { return kb; }
  __device__ int kend() const { return ke; }
  template <class V>
  __device__ void fix_b(V& v, int n0) const {  // column K_ of B is the ones column (bias gradient)
#pragma unroll
    for (int i = 0; i < VW; ++i) v[i] = (n0 + i == K_) ? 1.f : v[i];
  }
  __device__ void store_col(int mb, int n, const f32x16& acc, int M) {
    float* c = slab + ((size_t)z * N_ + mb) * (K_ + 1) + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dm = (r & 3) + 8 * (r >> 2);
      if (mb + dm < M) c[dm * (K_ + 1)] = acc[r];
    }
  }
  __device__ void finish() {}
};

// out[m][n] (+)= sum_z slab[z][m][n] for n < K ; db[m] (+)= sum_z slab[z][m][K]
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int splits, int Mrows, int Kcols,
                                   float* __restrict__ dW, int lddw, float* __restrict__ db,
                                   int accumulate) {
  const int total = Mrows * (Kcols + 1);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int m = i / (Kcols + 1), n = i % (Kcols + 1);
    // four independent chains (fixed order -> reproducible) so the loads of a thread overlap
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
      s0 += slab[(size_t)z * total + i];
      s1 += slab[(size_t)(z + 1) * total + i];
      s2 += slab[(size_t)(z + 2) * total + i];
      s3 += slab[(size_t)(z + 3) * total + i];
    }
    for (; z < splits; ++z) s0 += slab[(size_t)z * total + i];
    const float s = (s0 + s1) + (s2 + s3);
    if (n < Kcols) {
      float* p = dW + (size_t)m * lddw + n;
      *p = accumulate ? *p + s : s;
    } else if (db) {
      db[m] = accumulate ? db[m] + s : s;
    }
  }
}

}  // namespace repo
#include "wgrad_direct.h"
#include "wgrad_tr.h"
namespace repo {

static int wgrad_splits(long rows, long N, long K) {
  const long tiles = ((N + 63) / 64) * ((K + 1 + 63) / 64);
  long want = (768 + tiles - 1) / tiles;  // ~3 workgroups per CU
  long maxs = (rows + 63) / 64;           // at least 64 rows per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

// ---- several weight gradients in ONE launch pair (the layers of a head, the eight matrices of the scan): a dense 1-D grid
// walks the concatenated (tile x, tile y, split) ranges of the jobs.  A 200 x 200 gradient alone is 16 tiles x 48 splits of a few microseconds each; back
// to back, each launch drains before the next fills the chip, and the 1-row output layers get a launch of their own.
struct WgradJobDev {
  Dense2D A, B;
  float* slab;
  float* dW;
  float* db;
  int rows, N, K, rps, zstart, lddw;
  int bstart, gx, gy;  // first flat tile id of the job, its tile grid
};
struct WgradJobs {
  WgradJobDev job[kMaxWgradGroup];
  int njobs;
};
struct VWgradGroupOp {
  static constexpr bool A_VK = false;
  static constexpr bool B_VK = false;
  static constexpr int VW = 4;
  Dense2D A, B;
  float* slab;
  int N_, K_, z, kb, ke, rows_, rps_;
  WgradJobs g;

  // The job table is a kernel argument: it is only ever indexed with compile-time constants (a run-time index
  // would move the whole operator into scratch memory); the selection is a chain of scalar selects.
  __device__ void decode(int t, int& bx, int& by, int& bz) {
    WgradJobDev w = g.job[0];
#pragma unroll
    for (int i = 1; i < kMaxWgradGroup - 1; ++i)
      if (i < g.njobs && t >= g.job[i].bstart) w = g.job[i];
    const int local = t - w.bstart;
    bx = local % w.gx;
    by = (local / w.gx) % w.gy;
    bz = local / (w.gx * w.gy);  // split index inside the job
    A = w.A, B = w.B, slab = w.slab, N_ = w.N, K_ = w.K;
    rows_ = w.rows, rps_ = w.rps;
  }
  __device__ void init(int zz) {
    z = zz;
    kb = zz * rps_;
    ke = min(rows_, kb + rps_);
  }
  __device__ int M() const { return N_; }
  __device__ int N() const { return K_ + 1; }
  __device__ int kbeg() const { return kb; }
  __device__ int kend() const { return ke; }
  template <class V>
  __device__ void fix_b(V& v, int n0) const {
#pragma unroll
    for (int i = 0; i < VW; ++i) v[i] = (n0 + i == K_) ? 1.f : v[i];
  }
  __device__ void store_col(int mb, int n, const f32x16& acc, int M) {
    float* c = slab + ((size_t)z * N_ + mb) * (K_ + 1) + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dm = (r & 3) + 8 * (r >> 2);
      if (mb + dm < M) c[dm * (K_ + 1)] = acc[r];
    }
  }
  __device__ void finish() {}
};

__global__ void slab_reduce_group_kernel(WgradJobs g, int accumulate) {
  WgradJobDev w = g.job[0];
  int zend = g.job[1].zstart;
#pragma unroll
  for (int i = 1; i < kMaxWgradGroup - 1; ++i)
    if (i == (int)blockIdx.y) w = g.job[i], zend = g.job[i + 1].zstart;  // job[njobs] is the end marker
  const int splits = zend - w.zstart;
  const int total = w.N * (w.K + 1);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int m = i / (w.K + 1), n = i % (w.K + 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // fixed order: reproducible
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
      s0 += w.slab[(size_t)z * total + i];
      s1 += w.slab[(size_t)(z + 1) * total + i];
      s2 += w.slab[(size_t)(z + 2) * total + i];
      s3 += w.slab[(size_t)(z + 3) * total + i];
    }
    for (; z < splits; ++z) s0 += w.slab[(size_t)z * total + i];
    const float s = (s0 + s1) + (s2 + s3);
    if (n < w.K) {
      float* p = w.dW + (size_t)m * w.lddw + n;
      *p = accumulate ? *p + s : s;
    } else if (w.db) {
      w.db[m] = accumulate ? w.db[m] + s : s;
    }
  }
}

size_t gemm_wgrad_group_ws_bytes(const WgradDesc* d, int n) {
  size_t b = 0;
  for (int i = 0; i < n; ++i) b += (repo_gemm_wgrad_workspace_bytes(d[i].M, d[i].N, d[i].K) + 255) & ~(size_t)255;
  return b;
}

int gemm_wgrad_group(const WgradDesc* d, int n, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n <= 0) return REPO_OK;
  bool plain = n == 1 || n > kMaxWgradGroup - 1;
  for (int i = 0; i < n; ++i) plain = plain || d[i].M <= 0 || d[i].N <= 0 || d[i].K <= 0;
  if (plain) {  // degenerate shapes take the single-job path with its own checks
    for (int i = 0; i < n; ++i) {
      const int rc = repo_gemm_wgrad(d[i].M, d[i].N, d[i].K, d[i].dY, d[i].lddy, d[i].X, d[i].ldx, d[i].dW, d[i].lddw,
                                     d[i].db, accumulate, ws, ws_bytes, stream);
      if (rc) return rc;
    }
    return REPO_OK;
  }
  REPO_REQUIRE(ws && ws_bytes >= gemm_wgrad_group_ws_bytes(d, n), REPO_E_WS_TOO_SMALL);
  VWgradGroupOp op{};
  WdJobs dj{};
  int ndirect = 0;
  bool use_tr = t_bgemm_enabled != 0;   // wgrad_tr.h (bf16x6) if every direct job fits it, else wgrad_direct.h (fp32 MFMA)
#ifdef WT_DISABLE   // A/B builds (tools/build_variant.sh)
  use_tr = false;
#endif
  for (int i = 0; i < n; ++i) {
    const bool dk = wgrad_direct_ok(d[i].M, d[i].N, d[i].K, d[i].lddy, d[i].ldx);
    ndirect += dk ? 1 : 0;
    use_tr = use_tr && (!dk || wgrad_tr_ok(d[i].M, d[i].N, d[i].K, d[i].lddy, d[i].ldx));
  }
  if (ndirect > kWdMaxJobs) ndirect = kWdMaxJobs;
  // one workgroup per CU over all direct jobs of the launch (the transposing kernel has two per row range)
  int dsplits = ndirect ? (use_tr ? 128 : 256) / ndirect : 0;
  if (dsplits > kWdMaxSplits) dsplits = kWdMaxSplits;
  char* w = (char*)ws;
  long nblocks = 0;
  int z = 0, rmax = 0;
  for (int i = 0; i < n; ++i) {
    const WgradDesc& q = d[i];
    REPO_REQUIRE(q.dY && q.X && q.dW, REPO_E_BADARG);
    REPO_REQUIRE(q.M < kMaxIdx && q.N < kMaxIdx && q.K < kMaxIdx - 1 && q.M * q.lddy < kMaxBufElems &&
                     q.M * q.ldx < kMaxBufElems,
                 REPO_E_SHAPE);
    // the wide hidden layers of a head at tens of thousands of rows go straight from global memory to the matrix
    // cores (wgrad_direct.h); everything else (1- and 12-row output layers, short row counts) through the tile engine
    const bool direct = dj.njobs < kWdMaxJobs && wgrad_direct_ok(q.M, q.N, q.K, q.lddy, q.ldx);
    const int splits = direct ? dsplits : wgrad_splits(q.M, q.N, q.K);
    if (direct) {
      WdJob& x = dj.job[dj.njobs++];
      x.dY = q.dY, x.X = q.X, x.slab = (float*)w;
      x.rows = (int)q.M, x.N = (int)q.N, x.K = (int)q.K, x.lddy = (int)q.lddy, x.ldx = (int)q.ldx;
      x.rps = (int)(((q.M + splits - 1) / splits + 1) & ~1L);  // even: row pairs never straddle two ranges
    }
    WgradJobDev& j = op.g.job[i];
    j.A = Dense2D{q.dY, 4u * (unsigned)((q.M - 1) * q.lddy + q.N), (int)q.lddy};
    j.B = Dense2D{q.X, 4u * (unsigned)((q.M - 1) * q.ldx + q.K), (int)q.ldx};
    j.slab = (float*)w;
    j.dW = q.dW, j.db = q.db, j.lddw = (int)q.lddw;
    j.rows = (int)q.M, j.N = (int)q.N, j.K = (int)q.K;
    j.rps = (int)((q.M + splits - 1) / splits);
    j.zstart = z;
    j.gx = (int)cdiv(q.K + 1, T64x64::BN), j.gy = (int)cdiv(q.N, T64x64::BM);
    j.bstart = (int)nblocks;
    if (!direct) nblocks += (long)j.gx * j.gy * splits;  // a direct job owns no tiles of the flat grid
    z += splits;
    w += (repo_gemm_wgrad_workspace_bytes(q.M, q.N, q.K) + 255) & ~(size_t)255;
    const int total = (int)(q.N * (q.K + 1));
    if (total > rmax) rmax = total;
  }
  op.g.job[n].zstart = z;  // end marker
  op.g.job[n].bstart = (int)nblocks;
  op.g.njobs = n;
  if (dj.njobs > 0) {
    if (use_tr) {
      const int rc = launch_wgrad_tr(dj, dsplits, stream);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(wgrad_direct_kernel, dim3(dsplits, dj.njobs), dim3(512), 0, stream, dj);
      REPO_CHECK_LAUNCH();
    }
  }
  if (nblocks > 0) {
    const int rc = launch_vgemm_flat<T64x64>(op, nblocks, stream);
    if (rc) return rc;
  }
  const int blocks = cdiv(rmax, 256) < 1024 ? cdiv(rmax, 256) : 1024;
  hipLaunchKernelGGL(slab_reduce_group_kernel, dim3(blocks, n), dim3(256), 0, stream, op.g, accumulate);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

}  // namespace repo

using namespace repo;

extern "C" int repo_gemm(int transa, int transb, int64_t M, int64_t N, int64_t K, const float* A,
                         int64_t lda, const float* B, int64_t ldb, const float* bias, int64_t bias_div,
                         float* C, int64_t ldc, int epi, const float* aux, int64_t ldaux, int accumulate,
                         hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(M >= 0 && N >= 0 && K >= 0, REPO_E_SHAPE);
  if (M == 0 || N == 0) return REPO_OK;
  REPO_REQUIRE(A && B && C, REPO_E_BADARG);
  REPO_REQUIRE((epi >= REPO_EPI_NONE && epi <= REPO_EPI_MUL_DRELU) || epi == REPO_EPI_FILM_RELU, REPO_E_BADARG);
  REPO_REQUIRE((epi != REPO_EPI_MUL_DELU && epi != REPO_EPI_MUL_DRELU && epi != REPO_EPI_FILM_RELU) || aux, REPO_E_BADARG);
  // FiLM: one table row per output row, [scale | shift] over the N / bias_div channels; not on the <= 8-row vector path
  {
    const int64_t fd = bias_div < 0 ? -bias_div : (bias_div > 0 ? bias_div : 1);   // pixels per FiLM channel
    REPO_REQUIRE(epi != REPO_EPI_FILM_RELU || (M > 8 && !accumulate && ldaux % 2 == 0 && ldaux / 2 >= (N + fd - 1) / fd), REPO_E_BADARG);
  }
  REPO_REQUIRE(M < kMaxIdx && N < kMaxIdx && K < kMaxIdx && lda < kMaxIdx && ldb < kMaxIdx && ldc < kMaxIdx,
               REPO_E_SHAPE);
  {  // operand offsets are 32-bit inside the kernel
    const int64_t ea = transa ? (K - 1) * lda + M : (M - 1) * lda + K;
    const int64_t eb = transb ? (N - 1) * ldb + K : (K - 1) * ldb + N;
    REPO_REQUIRE(ea < kMaxBufElems && eb < kMaxBufElems && M * ldc < kMaxIdx && M * ldaux < kMaxIdx, REPO_E_SHAPE);
  }
  if (bias_div == 0 || (bias_div < 0 && epi != REPO_EPI_FILM_RELU)) bias_div = 1;   // (< 0 only means something to FiLM: header)
  // K == 1 (outer products, e.g. the gradient through a scalar output layer): a contiguous M x 1 / N x 1
  // operand is its own transpose, which turns its k-vectors (K % 2 != 0: gather engine) into m/n-vectors
  if (K == 1 && !transa && lda == 1) { transa = 1; lda = M; }
  if (K == 1 && transb && ldb == 1) { transb = 0; ldb = N; }
  if (M <= 8 && !transa) {
    if (transb) {
      hipLaunchKernelGGL((gemv_small_kernel<true>), dim3(cdiv(N, 4)), dim3(256), 0, stream, (int)M, (int)N, (int)K, A,
                         (int)lda, B, (int)ldb, bias, (int)bias_div, C, (int)ldc, epi, aux, (int)ldaux, accumulate);
    } else {
      hipLaunchKernelGGL((gemv_small_kernel<false>), dim3(cdiv(N, 256)), dim3(256), 0, stream, (int)M, (int)N, (int)K,
                         A, (int)lda, B, (int)ldb, bias, (int)bias_div, C, (int)ldc, epi, aux, (int)ldaux, accumulate);
    }
    REPO_CHECK_LAUNCH();
    return REPO_OK;
  }
  // big products: the bf16x6 engine (bgemm.h): fp32-accurate at 6/16 of the fp32 MFMA's time per k
  if (t_bgemm_enabled && bgemm_ok(M, N, K, !transa, lda, transb != 0, ldb, A, B)) {
    BgArgs a{Dense2D{A, 4u * (unsigned)(transa ? (K - 1) * lda + M : (M - 1) * lda + K), (int)lda},
             Dense2D{B, 4u * (unsigned)(transb ? (N - 1) * ldb + K : (K - 1) * ldb + N), (int)ldb},
             bias, aux, C, (int)ldc, (int)ldaux, (int)bias_div, (int)M, (int)N, (int)K, epi, accumulate};
    if (!transa && transb) return bgemm_dispatch<true, true>(a, stream);
    if (!transa && !transb) return bgemm_dispatch<true, false>(a, stream);
    if (transa && !transb) return bgemm_dispatch<false, false>(a, stream);
    return bgemm_dispatch<false, true>(a, stream);
  }
  // vector-load engine whenever the k-contiguous operands (A if !transa, B if transb) have K % VW == 0
  const bool kvec = !transa || transb;
  const int vw = (!kvec || K % 4 == 0) ? 4 : (K % 2 == 0 ? 2 : 0);
  if (vw) {
    const unsigned abytes = 4u * (unsigned)(transa ? (K - 1) * lda + M : (M - 1) * lda + K);
    const unsigned bbytes = 4u * (unsigned)(transb ? (N - 1) * ldb + K : (K - 1) * ldb + N);
#define REPO_VGEMM_CASE(TA, TB, VW)                                                                            \
  {                                                                                                            \
    VGemmOp<TA, TB, VW> op{Dense2D{A, abytes, (int)lda}, Dense2D{B, bbytes, (int)ldb}, bias, aux, C, (int)ldc, \
                           (int)ldaux, (int)bias_div, (int)M, (int)N, (int)K, epi, accumulate};                \
    return vgemm_dispatch(op, M, N, stream);                                                                   \
  }
    if (vw == 4) {
      if (!transa && !transb) REPO_VGEMM_CASE(false, false, 4)
      if (!transa && transb) REPO_VGEMM_CASE(false, true, 4)
      if (transa && !transb) REPO_VGEMM_CASE(true, false, 4)
      REPO_VGEMM_CASE(true, true, 4)
    } else {
      if (!transa && !transb) REPO_VGEMM_CASE(false, false, 2)
      if (!transa && transb) REPO_VGEMM_CASE(false, true, 2)
      if (transa && !transb) REPO_VGEMM_CASE(true, false, 2)
      REPO_VGEMM_CASE(true, true, 2)
    }
#undef REPO_VGEMM_CASE
  }
#define REPO_GEMM_CASE(TA, TB)                                                                   \
  {                                                                                              \
    GemmOp<TA, TB> op{A,        B,        bias,           aux,    C,      (int)lda, (int)ldb, (int)ldc, \
                      (int)ldaux, (int)bias_div, (int)M, (int)N, (int)K, epi,      accumulate};         \
    return gemm_dispatch(op, M, N, stream);                                                      \
  }
  if (!transa && !transb) REPO_GEMM_CASE(false, false)
  if (!transa && transb) REPO_GEMM_CASE(false, true)
  if (transa && !transb) REPO_GEMM_CASE(true, false)
  REPO_GEMM_CASE(true, true)
#undef REPO_GEMM_CASE
}

// dst[c][r] = src[r][c] (rows x cols -> cols x rows, ldd >= rows; columns [rows, ldd) of dst are written as zeros).
// 64 x 64 tiles through LDS, 16-byte accesses on both sides.  Used to bring an operand of a big product into the
// k-contiguous form the bf16x6 engine runs fastest on (bgemm.h: NT 117 vs NN 153 us at 2450 x 3200 x 1024).
namespace repo {
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const float* __restrict__ src, int lds_,
                                                        float* __restrict__ dst, int ldd) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 16 * i, c = c0 + 4 * tx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      if (c + 3 < cols) v = *reinterpret_cast<const f32x4*>(src + (size_t)r * lds_ + c);
      else
        for (int e = 0; e < 4; ++e)
          if (c + e < cols) v[e] = src[(size_t)r * lds_ + c + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[ty + 16 * i][4 * tx + e] = v[e];
  }
  __syncthreads();
  const bool last_rt = r0 + 64 >= rows;   // this tile also owns dst's pad columns [rows, ldd)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 16 * i, r = r0 + 4 * tx;   // dst row c, columns r .. r+3
    if (c >= cols) continue;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = tile[4 * tx + e][ty + 16 * i];   // zero beyond `rows` (loaded as zeros)
    float* d = dst + (size_t)c * ldd + r;
    if (r + 3 < (last_rt ? ldd : rows)) *reinterpret_cast<f32x4*>(d) = v;
    else
      for (int e = 0; e < 4; ++e)
        if (r + e < (last_rt ? ldd : rows)) d[e] = v[e];
  }
}
}  // namespace repo

extern "C" int repo_transpose(int64_t rows, int64_t cols, const float* src, int64_t lds, float* dst, int64_t ldd,
                              hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(rows > 0 && cols > 0 && lds >= cols && ldd >= rows && ldd < rows + 64, REPO_E_SHAPE);
  REPO_REQUIRE(rows * lds < kMaxIdx && cols * ldd < kMaxIdx, REPO_E_SHAPE);
  REPO_REQUIRE(src && dst, REPO_E_BADARG);
  REPO_REQUIRE(lds % 4 == 0 && ldd % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, REPO_E_ALIGN);
  hipLaunchKernelGGL(repo::transpose_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0,
                     stream, (int)rows, (int)cols, src, (int)lds, dst, (int)ldd);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}

extern "C" int repo_gemm_nt_pays(int64_t M, int64_t N, int64_t K) {
  static float probe[4] __attribute__((aligned(16)));   // sizes only: an aligned operand with ld % 4 == 0 stands in
  return t_bgemm_enabled && bgemm_ok(M, N, K, true, (K + 3) / 4 * 4, true, (K + 3) / 4 * 4, probe, probe) ? 1 : 0;
}

extern "C" int repo_debug_bgemm(int enable) {
  const int prev = t_bgemm_enabled;
  t_bgemm_enabled = enable ? 1 : 0;
  return prev;
}

extern "C" size_t repo_gemm_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int splits = wgrad_splits(M, N, K);
  // room for the direct kernel's row ranges (wgrad_direct.h) -- only for jobs that can take that path (its shape
  // conditions; the leading dimensions are not known here): a 128-slab floor for every job made decoder fc1's and
  // W_bq's workspaces 6-10x larger than the tile engine needs
  if (wgrad_direct_ok(M, N, K, 4, 4) && splits < kWdMaxSplits) splits = kWdMaxSplits;
  return (size_t)splits * (size_t)N * (size_t)(K + 1) * sizeof(float);
}

extern "C" int repo_gemm_wgrad(int64_t M, int64_t N, int64_t K, const float* dY, int64_t lddy,
                               const float* X, int64_t ldx, float* dW, int64_t lddw, float* db,
                               int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(M >= 0 && N >= 0 && K >= 0, REPO_E_SHAPE);
  if (N == 0 || K == 0) return REPO_OK;
  REPO_REQUIRE(dY && X && dW, REPO_E_BADARG);
  REPO_REQUIRE(M < kMaxIdx && N < kMaxIdx && K < kMaxIdx - 1, REPO_E_SHAPE);
  REPO_REQUIRE(M * lddy < kMaxBufElems && M * ldx < kMaxBufElems, REPO_E_SHAPE);
  if (M == 0) {
    if (!accumulate) {
      for (int64_t n = 0; n < N; ++n) (void)hipMemsetAsync(dW + n * lddw, 0, K * sizeof(float), stream);
      if (db) (void)hipMemsetAsync(db, 0, N * sizeof(float), stream);
    }
    return REPO_OK;
  }
  // a big weight gradient without a bias column (the decoder's 1024 x 3200 first transposed conv): ONE product on the
  // bf16x6 engine, dW[n][k] = sum_rows dY[row][n] X[row][k] with both operands row-contiguous -- no split-K slabs
  if (!db && t_bgemm_enabled && bgemm_ok(N, K, M, false, lddy, false, ldx, dY, X)) {
    BgArgs a{Dense2D{dY, 4u * (unsigned)((M - 1) * lddy + N), (int)lddy}, Dense2D{X, 4u * (unsigned)((M - 1) * ldx + K), (int)ldx},
             nullptr, nullptr, dW, (int)lddw, 0, 1, (int)N, (int)K, (int)M, REPO_EPI_NONE, accumulate};
    return bgemm_dispatch<false, false>(a, stream);
  }
  const int splits = wgrad_splits(M, N, K);
  REPO_REQUIRE(ws && ws_bytes >= repo_gemm_wgrad_workspace_bytes(M, N, K), REPO_E_WS_TOO_SMALL);
  const int rps = (int)((M + splits - 1) / splits);
  VWgradOp op{Dense2D{dY, 4u * (unsigned)((M - 1) * lddy + N), (int)lddy},
              Dense2D{X, 4u * (unsigned)((M - 1) * ldx + K), (int)ldx},
              (float*)ws, (int)M, (int)N, (int)K, rps, 0, 0, 0};
  const int rc = launch_vgemm<T64x64>(op, N, K + 1, splits, stream);
  if (rc) return rc;
  const int total = (int)(N * (K + 1));
  const int blocks = cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)ws, splits, (int)N,
                     (int)K, dW, (int)lddw, db, accumulate);
  REPO_CHECK_LAUNCH();
  return REPO_OK;
}
