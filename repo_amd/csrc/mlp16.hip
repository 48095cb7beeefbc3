// Fused dense heads (reward / value / actor MLPs): the whole layer chain of a row tile in ONE kernel.
//
// A head is in_dim -> hidden (ELU) x (L-1) -> out_dim.  Layer by layer through the GEMM engine every hidden
// activation is written to HBM and read back by the next launch, and a 200-wide layer is too small a GEMM to fill
// the chip (0.35-0.4 of the fp32 MFMA peak in isolation, 4-5 launches per call).  Here a (persistent) workgroup takes
// 16 or 32 rows at a time through the whole chain: activations stay in LDS (k4-interleaved tiles, rowtile.h), weights stream from L2 as packed
// 16-byte fragments one stage ahead of their use, and the hidden activations leave the CU once, as the values the
// backward pass needs.  The reverse chain (d out -> d pre-activations -> d input) has the same shape with the
// transposed packs; it emits the per-layer pre-activation gradients for the weight-gradient GEMMs.
//
// Reference: the nn.Sequential heads of models/actor_critic.py:10-60 and models/reward.py (Linear + ELU stacks) and
// autograd's backward through them.
#include "rowtile.h"

namespace repo {

constexpr int kMaxL = 5;

// tools/probe/mlp_trace.hip builds this file with MLP_TRACE: per wave and layer cycle stamps
#ifdef MLP_TRACE
#define MLP_STAMP(slot) \
  if (p.trace && lane == 0) p.trace[(((size_t)blockIdx.x * 8 + tile_no) * 8 + l) * 32 + wave * 4 + (slot)] = clock64()
#else
#define MLP_STAMP(slot)
#endif

struct MlpFwdArgs {
  long long* trace;
  int rows, in_dim, hidden, out_dim, ldx, ldo;
  const float* x;
  const float* wpack;
  unsigned wbytes;
  unsigned W[kMaxL];
  const float* b[kMaxL];
  float* hid[kMaxL - 1];
  float* out;
};

struct MlpBwdArgs {
  int rows, in_dim, hidden, out_dim, lddout, lddx, accumulate_dx;
  const float* dout;
  const float* wpack;  // transposed packs: layer l as (N' = k_l, K' = n_l)
  unsigned wbytes;
  unsigned W[kMaxL];
  const float* hid[kMaxL - 1];
  float* dsave[kMaxL - 1];  // d pre-activation of hidden layer l (rows x hidden), or null
  float* dx;                // null: skip the input gradient
  // TWO upstream gradients through ONE chain (out_dim == 1; null: off).  A scalar head's reverse chain is, per row, the
  // chain of a UNIT upstream times that row's scalar, so one pass serves two losses that meet in the head: it runs on
  // upstream 1, the input gradient leaves scaled by dout[row] and the saved pre-activation gradients (the weight
  // gradients' operands) by dout_w[row] (rows >= rows_w: 0).  The actor-critic update used to run the value head's
  // chain twice -- d returns / d v for the actor's loss (input gradient only), the critic's own loss (weights only).
  const float* dout_w;
  int rows_w;
};

// acc[rb][t] += (A_rb[16 x K] * column tile t of the stream)^T, for RB row blocks sharing every weight fragment.
// The weights are the MFMA's A operand and the activations its B operand (the fragments of the two have the same
// lane layout), so the accumulator comes out transposed: lane l holds row l % 16 of the tile and the FOUR
// CONSECUTIVE columns 4 * (l / 16) + r -- one 16-byte LDS store into the k4-interleaved tile of the next layer and one
// 16-byte global store per lane, where the untransposed product needs four scalar stores of each kind.
template <int NBLK, int RB>
__device__ __forceinline__ void mrun(f32x4v (&acc)[RB][2], const float* A, int rb_stride, WWin& w,
                                     __amdgpu_buffer_rsrc_t rw, int lane) {
  constexpr int PD = NBLK < kPD ? NBLK : kPD;
  const int row = lane & 15, kq = lane >> 4;
  const float* ap = A + (kq * kR + row) * 4;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    f32x4v a[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const f32x4v*>(ap + rb * rb_stride + b * 16 * kR);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.b0[b % PD][j], a[rb][j], acc[rb][0], 0, 0, 0);
        if (w.two) acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.b1[b % PD][j], a[rb][j], acc[rb][1], 0, 0, 0);
      }
    }
    if (b + PD < NBLK) {
      w.b0[b % PD] = wld(rw, w.v0, w.W0 + (b + PD) * w.s0);
      w.b1[b % PD] = wld(rw, w.v1, w.W1 + (b + PD) * w.s1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int RB>
__device__ __forceinline__ void zero_acc(f32x4v (&acc)[RB][2]) {
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = f32x4v{0.f, 0.f, 0.f, 0.f};
}

// Which 16-row blocks a workgroup owns.  The grid is at most two workgroups per CU (what LDS allows) and each
// owns a contiguous run of blocks, processed in pairs (32 rows share every weight fragment) plus at most one single:
// with one workgroup per 32 rows, 34300 rows are 1072 workgroups on 512 slots -- 2.09 rounds, the last one 9 % full.
struct Blocks {
  int first, count;
};
__device__ __forceinline__ Blocks my_blocks(int rows) {
  const int nb = (rows + kR - 1) / kR, g = gridDim.x, b = blockIdx.x;
  const int base = nb / g, rem = nb % g;
  return Blocks{b * base + min(b, rem), base + (b < rem ? 1 : 0)};
}
// Column tiles: 13 tiles of a 200-wide layer on 8 waves give five waves two tiles and three waves one, so one SIMD
// (waves w and w + 4 share one) carries 4 tiles against 3 on the others.  Rotating which waves are the heavy ones
// by workgroup and by tile evens the MFMA load of the four SIMDs out over a CU's resident workgroups.
__device__ __forceinline__ int wave_rotation() {
  const int j = blockIdx.x >> 3;  // workgroups go round-robin over the 8 XCDs
  return j + (j >> 5);
}

// Input rows -> tile, in two halves so that the loads of the NEXT tile fly during the last layer of the current one.
// A wave takes 8 rows x 8 consecutive columns per instruction: the LDS store then touches all 32 banks twice (rows
// are 4 banks apart, k % 4 fills them), where consecutive lanes on consecutive columns would hit 4 banks 16 times.
template <int BI>
struct RowLoad {
  float v[BI];
  float* dst;
  int f0;
  bool on;
  // rx: buffer over the whole input (a load past a row's end reads the next row or, past the end, 0; neither lands)
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rx, int ldx, float* T, int TS, int r0, int nr, int wave,
                                        int lane) {
    const int row = 8 * (wave & 3) + (lane >> 3);
    on = row < nr;
    f0 = 8 * (wave >> 2) + (lane & 7);
    const unsigned vo = 4u * ((unsigned)(r0 + min(row, nr - 1)) * (unsigned)ldx + (unsigned)f0);
    dst = T + (row >> 4) * TS + (row & 15) * 4;
#pragma unroll
    for (int j = 0; j < BI; ++j)
      v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo + 64u * j, 0, 0));
  }
  __device__ __forceinline__ void land(int F) {
    if (!on) return;
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int f = f0 + 16 * j;
      if (f < F) dst[(f >> 2) * (4 * kR) + (f & 3)] = v[j];
    }
  }
};

struct NextTile {
  int r0, nr;  // nr = 0: none
};

// BI / BH: 16-k blocks of in_dim and of hidden; L layers (L - 1 hidden + output); RB row blocks of 16 per tile.
// vw: this wave's (rotated) column-tile owner index.  The tile's rows are in Ta on entry (k4-interleaved; padding k
// holds finite values and meets zero weight rows); layer l reads Ta / Tb alternately, so the buffer the last layer
// does NOT read is free for the next tile's rows while the (narrow) output layer runs.
template <int BI, int BH, int L, int RB, int TS>
__device__ __forceinline__ void mlp_fwd_tile(const MlpFwdArgs& p, float* Ta, float* Tb, __amdgpu_buffer_rsrc_t rw,
                                             __amdgpu_buffer_rsrc_t rx,
                                             int r0, int nr, NextTile nx, int vw, int wave, int lane, int tile_no) {
  const int lq = lane >> 4;
  const int F = p.in_dim, Hd = p.hidden;
  WWin w0, w1;
  dense_open<BI>(w0, rw, p.W[0], Hd, vw, lane);
  __syncthreads();  // this tile's rows have landed; the previous tile's last layer is done with Tb
#pragma unroll
  for (int l = 0; l < L; ++l) {
    WWin& wc = (l & 1) ? w1 : w0;
    WWin& wn = (l & 1) ? w0 : w1;
    const float* src = (l & 1) ? Tb : Ta;
    float* dst = (l & 1) ? Ta : Tb;
    const bool last = l == L - 1;
    const int N = last ? p.out_dim : Hd;
    if (l + 1 < L) dense_open<BH>(wn, rw, p.W[l + 1], (l + 1 == L - 1) ? p.out_dim : Hd, vw, lane);
    RowLoad<BI> nxt;
    if (last && nx.nr > 0) nxt.issue(rx, p.ldx, dst, TS, nx.r0, nx.nr, wave, lane);
    MLP_STAMP(0);
    if (wc.act) {
      const int m = lane & 15;
      f32x4v bv[2];  // hidden layers: this lane's bias quads, requested ahead of the MFMA chain
      if (!last) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          bv[t] = *reinterpret_cast<const f32x4v*>(p.b[l] + min((vw + t * kW) * 16 + 4 * lq, Hd - 4));
      }
      f32x4v acc[RB][2];
      zero_acc<RB>(acc);
      if (l == 0)
        mrun<BI, RB>(acc, src, TS, wc, rw, lane);
      else
        mrun<BH, RB>(acc, src, TS, wc, rw, lane);
      MLP_STAMP(1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n0 = (vw + t * kW) * 16 + 4 * lq;
        if ((t == 0 || wc.two) && n0 < N) {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            const int row = rb * kR + m;
            if (last) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (n0 + r < N && row < nr) p.out[(size_t)(r0 + row) * p.ldo + n0 + r] = acc[rb][t][r] + p.b[l][n0 + r];
            } else {  // hidden % 4 == 0: the quad is whole
              f32x4v v;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = elu(acc[rb][t][r] + bv[t][r]);
              *reinterpret_cast<f32x4v*>(dst + rb * TS + ai(n0, m)) = v;
              if (row < nr) *reinterpret_cast<f32x4v*>(p.hid[l] + (size_t)(r0 + row) * Hd + n0) = v;
            }
          }
        }
      }
    }
    MLP_STAMP(2);
    if (last && nx.nr > 0) nxt.land(F);
    if (!last) __syncthreads();
    MLP_STAMP(3);
  }
}

template <int BI, int BH, int L>
__global__ __launch_bounds__(512, 4) void mlp_fwd_kernel(MlpFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TS = 16 * (BI > BH ? BI : BH) * kR;  // floats per 16-row tile
  float* Ta = lds;
  float* Tb = lds + 2 * TS;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const __amdgpu_buffer_rsrc_t rw = wrsrc(p.wpack, p.wbytes);
  const __amdgpu_buffer_rsrc_t rx = wrsrc(p.x, 4u * ((unsigned)(p.rows - 1) * (unsigned)p.ldx + (unsigned)p.in_dim));
  for (int i = tid0; i < 4 * TS / 4; i += 512) reinterpret_cast<f32x4v*>(lds)[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  Blocks mb = my_blocks(p.rows);
  {
    RowLoad<BI> first;
    const int r0 = mb.first * kR;
    first.issue(rx, p.ldx, Ta, TS, r0, min((mb.count >= 2 ? 2 : 1) * kR, p.rows - r0), wave, tid0 & 63);
    first.land(p.in_dim);
  }
  int tile_no = 0;
  for (int it = wave_rotation(); mb.count > 0; ++it) {
    const int vw = (wave + it) & (kW - 1);
    const int r0 = mb.first * kR;
    int lane = tid0 & 63;
    asm volatile("" : "+v"(lane));  // per-tile lane arithmetic stays inside the tile: hoisted out of this loop it spills
    const int rb = mb.count >= 2 ? 2 : 1;
    mb.first += rb, mb.count -= rb;
    NextTile nx{mb.first * kR, 0};
    if (mb.count > 0) nx.nr = min((mb.count >= 2 ? 2 : 1) * kR, p.rows - nx.r0);
    if (rb == 2)
      mlp_fwd_tile<BI, BH, L, 2, TS>(p, Ta, Tb, rw, rx, r0, min(2 * kR, p.rows - r0), nx, vw, wave, lane, tile_no);
    else
      mlp_fwd_tile<BI, BH, L, 1, TS>(p, Ta, Tb, rw, rx, r0, min(kR, p.rows - r0), nx, vw, wave, lane, tile_no);
    ++tile_no;
    if (L & 1) {  // the next tile's rows went where an odd chain's last layer does not read
      float* t = Ta;
      Ta = Tb;
      Tb = t;
    }
  }
}

// stage s = 0 .. L-1 handles layer l = L-1-s: T = D_l * W_l; s < L-1: D_{l-1} = T * elu'(h_{l-1}); s = L-1: dx = T.
// BO: 16-k blocks of out_dim.
template <int BI, int BH, int BO, int L, int RB, int TS>
__device__ __forceinline__ void mlp_bwd_tile(const MlpBwdArgs& p, float* T0, float* T1, __amdgpu_buffer_rsrc_t rw,
                                             int r0, int nr, int vw, int tid) {
  const int lane = tid & 63, lq = lane >> 4;
  const int F = p.in_dim, Hd = p.hidden, O = p.out_dim;
  WWin w0, w1;
  dense_open<BO>(w0, rw, p.W[L - 1], Hd, vw, lane);
  const bool dual = p.dout_w != nullptr;   // (uniform; out_dim == 1)
  for (int i = tid; i < RB * kR * O; i += 512) {
    const int row = i / O, f = i % O;
    if (row < nr) T0[(row >> 4) * TS + ai(f, row & 15)] = dual ? 1.f : p.dout[(size_t)(r0 + row) * p.lddout + f];
  }
  float sxr[RB], swr[RB];   // this lane's rows: the scalars of the two upstream gradients
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int row = r0 + rb * kR + (lane & 15);
    sxr[rb] = (dual && row < r0 + nr) ? p.dout[(size_t)row * p.lddout] : 1.f;
    swr[rb] = dual ? (row < p.rows_w ? p.dout_w[row] : 0.f) : 1.f;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < L; ++s) {
    const int l = L - 1 - s;
    WWin& wc = (s & 1) ? w1 : w0;
    WWin& wn = (s & 1) ? w0 : w1;
    const float* src = (s & 1) ? T1 : T0;
    float* dst = (s & 1) ? T0 : T1;
    const bool last = s == L - 1;
    if (last && !p.dx) break;
    const int N = last ? F : Hd;
    if (s + 1 < L) dense_open<BH>(wn, rw, p.W[l > 0 ? l - 1 : 0], (s + 1 == L - 1) ? F : Hd, vw, lane);
    if (wc.act) {
      const int m = lane & 15;
      // the activations this stage's epilogue multiplies by: requested ahead of the MFMA chain
      f32x4v hv[RB][2];
      if (!last) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int n0 = min((vw + t * kW) * 16 + 4 * lq, Hd - 4);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
            hv[rb][t] = *reinterpret_cast<const f32x4v*>(p.hid[l > 0 ? l - 1 : 0] +
                                                         (size_t)(r0 + min(rb * kR + m, nr - 1)) * Hd + n0);
        }
      }
      f32x4v acc[RB][2];
      zero_acc<RB>(acc);
      if (s == 0)
        mrun<BO, RB>(acc, src, TS, wc, rw, lane);
      else
        mrun<BH, RB>(acc, src, TS, wc, rw, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n0 = (vw + t * kW) * 16 + 4 * lq;
        if ((t == 0 || wc.two) && n0 < N) {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            const int row = rb * kR + m;
            if (last) {
              if (row < nr) {
                float* d = p.dx + (size_t)(r0 + row) * p.lddx + n0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (n0 + r < N) d[r] = p.accumulate_dx ? fmaf(acc[rb][t][r], sxr[rb], d[r]) : acc[rb][t][r] * sxr[rb];
              }
            } else {
              f32x4v g;
#pragma unroll
              for (int r = 0; r < 4; ++r) g[r] = acc[rb][t][r] * elu_grad_from_out(hv[rb][t][r]);
              *reinterpret_cast<f32x4v*>(dst + rb * TS + ai(n0, m)) = g;
              float* sv = p.dsave[l > 0 ? l - 1 : 0];
              if (sv && row < nr) *reinterpret_cast<f32x4v*>(sv + (size_t)(r0 + row) * Hd + n0) = g * swr[rb];
            }
          }
        }
      }
    }
    if (!last) __syncthreads();
  }
}

template <int BI, int BH, int BO, int L>
__global__ __launch_bounds__(512, 4) void mlp_bwd_kernel(MlpBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TS = 16 * (BH > BO ? BH : BO) * kR;
  float* T0 = lds;
  float* T1 = lds + 2 * TS;
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const __amdgpu_buffer_rsrc_t rw = wrsrc(p.wpack, p.wbytes);
  for (int i = tid0; i < 4 * TS / 4; i += 512) reinterpret_cast<f32x4v*>(lds)[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  Blocks mb = my_blocks(p.rows);
  for (int it = wave_rotation(); mb.count > 0; ++it) {
    const int vw = (wave + it) & (kW - 1);
    const int r0 = mb.first * kR;
    int tid = tid0;
    asm volatile("" : "+v"(tid));  // per-tile lane arithmetic stays inside the tile: hoisted out of this loop it spills
    if (mb.count >= 2) {
      mlp_bwd_tile<BI, BH, BO, L, 2, TS>(p, T0, T1, rw, r0, min(2 * kR, p.rows - r0), vw, tid);
      mb.first += 2, mb.count -= 2;
    } else {
      mlp_bwd_tile<BI, BH, BO, L, 1, TS>(p, T0, T1, rw, r0, min(kR, p.rows - r0), vw, tid);
      mb.first += 1, mb.count -= 1;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ host side
// Instantiated for the reference heads: in_dim = belief + state = 230 (15 blocks), hidden = 200 (13), 4 or 5 layers,
// out_dim <= 16 (reward / value: 1, actor: 2 A).  Other shapes run layer by layer through the GEMM engine.
constexpr int kBI = 15, kBH = 13;
bool mlp_fused_ok(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers) {
  auto blk = [](int64_t k) { return pad16((int)k) >> 4; };
  return (n_layers == 4 || n_layers == 5) && blk(in_dim) == kBI && blk(hidden) == kBH && hidden % 4 == 0 &&
         out_dim <= 16 &&
         rows * (in_dim > hidden ? in_dim : hidden) < kMaxIdx;  // callers also bound rows * ld of their operands
}
size_t mlp_fused_ws_floats(int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers) {
  // forward and transposed packs have the same padded sizes up to the side that is padded; take the larger of each
  const size_t f = pack_floats(hidden, in_dim) + (n_layers - 2) * pack_floats(hidden, hidden) + pack_floats(out_dim, hidden);
  const size_t b = pack_floats(in_dim, hidden) + (n_layers - 2) * pack_floats(hidden, hidden) + pack_floats(hidden, out_dim);
  return (f > b ? f : b) + 64;
}

// two workgroups per CU fit (2 x 61 KB LDS, <= 128 VGPRs)
static int grid_for(int rows) {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    slots = 2 * cus;
  }
  const int nb = (rows + kR - 1) / kR;
  return nb < slots ? nb : slots;
}

template <int L>
static int launch_fwd(const MlpFwdArgs& a, hipStream_t s) {
  constexpr int lds = 4 * 16 * (kBI > kBH ? kBI : kBH) * kR * 4;
  hipLaunchKernelGGL((mlp_fwd_kernel<kBI, kBH, L>), dim3(grid_for(a.rows)), dim3(512), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

int mlp_fused_fwd(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int L, const float* x, int64_t ldx,
                  const float* const* params, float* const* hidden_out, float* out, int64_t ldo, void* ws,
                  hipStream_t stream) {
  MlpFwdArgs a;
  a.trace = nullptr;
  a.rows = (int)rows, a.in_dim = (int)in_dim, a.hidden = (int)hidden, a.out_dim = (int)out_dim;
  a.ldx = (int)ldx, a.ldo = (int)ldo;
  a.x = x, a.out = out;
  float* w = (float*)ws;
  a.wpack = w;
  PackArgs pa;
  pa.njobs = 0;
  for (int l = 0; l < L; ++l) {
    const int n = l == L - 1 ? (int)out_dim : (int)hidden, k = l == 0 ? (int)in_dim : (int)hidden;
    pa.job[pa.njobs++] = PackJob{params[2 * l], w, n, k, k, 1};
    a.W[l] = (unsigned)((w - a.wpack) * sizeof(float));
    a.b[l] = params[2 * l + 1];
    if (l < L - 1) a.hid[l] = hidden_out[l];
    w += pack_floats(n, k);
  }
  a.wbytes = (unsigned)((w - a.wpack) * sizeof(float));
  int rc = launch_pack(pa, stream);
  if (rc) return rc;
  return L == 4 ? launch_fwd<4>(a, stream) : launch_fwd<5>(a, stream);
}

template <int L>
static int launch_bwd(const MlpBwdArgs& a, hipStream_t s) {
  constexpr int lds = 4 * 16 * kBH * kR * 4;
  hipLaunchKernelGGL((mlp_bwd_kernel<kBI, kBH, 1, L>), dim3(grid_for(a.rows)), dim3(512), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// dsave[l], l < L-1: where the pre-activation gradient of hidden layer l goes (null: not kept)
int mlp_fused_bwd(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int L, const float* const* params,
                  const float* const* hidden_acts, const float* dout, int64_t lddout, float* const* dsave, float* dx,
                  int64_t lddx, int accumulate_dx, void* ws, hipStream_t stream, const float* dout_w, int64_t rows_w) {
  MlpBwdArgs a;
  a.dout_w = dout_w, a.rows_w = (int)rows_w;
  a.rows = (int)rows, a.in_dim = (int)in_dim, a.hidden = (int)hidden, a.out_dim = (int)out_dim;
  a.lddout = (int)lddout, a.lddx = (int)lddx, a.accumulate_dx = accumulate_dx;
  a.dout = dout, a.dx = dx;
  float* w = (float*)ws;
  a.wpack = w;
  PackArgs pa;
  pa.njobs = 0;
  for (int l = (dx ? 0 : 1); l < L; ++l) {
    const int n = l == L - 1 ? (int)out_dim : (int)hidden, k = l == 0 ? (int)in_dim : (int)hidden;
    // transposed: pack column n' = k index, reduction k' = n index
    pa.job[pa.njobs++] = PackJob{params[2 * l], w, k, n, 1, k};
    a.W[l] = (unsigned)((w - a.wpack) * sizeof(float));
    w += pack_floats(k, n);
  }
  if (!dx) a.W[0] = a.W[1];
  for (int l = 0; l < L - 1; ++l) {
    a.hid[l] = hidden_acts[l];
    a.dsave[l] = dsave ? dsave[l] : nullptr;
  }
  a.wbytes = (unsigned)((w - a.wpack) * sizeof(float));
  int rc = launch_pack(pa, stream);
  if (rc) return rc;
  return L == 4 ? launch_bwd<4>(a, stream) : launch_bwd<5>(a, stream);
}

}  // namespace repo
