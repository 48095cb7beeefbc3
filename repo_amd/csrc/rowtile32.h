// Row-tile primitives on the bf16 matrix pipe ("bf16x6", bgemm.h): 32-row activation tiles, v_mfma_f32_32x32x16_bf16.
//
// The 16-row engine (rowtile.h) streams every fp32 weight of a layer chain from L2 once per 16 rows and runs on the
// fp32 MFMA (1/16 of the bf16 rate): a chain of 200-wide layers is bound by (MFMA time + per-layer latency) on a tile
// that is too small to amortise either.  Here
//   * an activation is split ONCE, where it is produced (the epilogue of the layer that writes it), into its three
//     exact bf16 parts, kept as three planes of the LDS tile: element (k, row) of a plane is the bf16 at
//     ((k / 8) * 32 + row) * 16 + (k % 8) * 2 bytes, so the B fragment of a 16-k block -- 8 consecutive k of one row
//     per lane -- is ONE conflict-free ds_read_b128 per plane;
//   * a weight matrix is split once per call by the pack kernel into the same three planes, fragment-ready:
//     [16-k block][plane][k-half][column (padded to 32)][8 bf16]: the A fragment of a block is one 16-byte buffer load
//     per plane, a 32-column tile of a block 3 KB;
//   * a product a*b is the six partial products with i + j <= 4 (bgemm.h) accumulated in fp32: six MFMAs of 8 passes
//     per 16 k and 32 x 32 outputs, against sixteen fp32 MFMAs of 8 passes on the 16-row engine for the same work;
//   * the product is formed transposed (weights = the MFMA's A operand): a lane's 16 accumulator registers are FOUR
//     QUADS, 4 consecutive output columns of ONE row each (columns n0 + 8 i + 4 (lane / 32) + j, row lane % 32):
//     the split of a quad is one 8-byte LDS store per plane into the next layer's tile, its fp32 value one 16-byte
//     global store into a row-major saved activation.
// A 32-row tile halves the weight traffic per row of the 16-row engine (the bf16 planes are 6 bytes per weight
// against 4): the rollout runs on 77 workgroups instead of 154, the CUs it leaves go to the other lane's kernels.
#pragma once
#include "rowtile.h"

namespace repo {

constexpr int kR32 = 32;
typedef __bf16 rt_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rt_f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef unsigned rt_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned rt_u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int pad32(int n) { return (n + 31) & ~31; }

// two floats -> their three bf16 parts, packed pairwise (low half = first element); exact: x = p1 + p2 + p3
__device__ __forceinline__ void rt_split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(rt_f32x2{x0, x1}, rt_bf16x2));
  const float r0 = x0 - __builtin_bit_cast(float, p1 << 16), r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(rt_f32x2{r0, r1}, rt_bf16x2));
  const float s0 = r0 - __builtin_bit_cast(float, p2 << 16), s1 = r1 - __builtin_bit_cast(float, p2 & 0xffff0000u);
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(rt_f32x2{s0, s1}, rt_bf16x2));
}

// ---- activation tile: three planes of KP (multiple of 16) k x 32 rows; ps = plane stride in bytes = 64 * KP
__device__ __forceinline__ int pofs(int k, int row) { return (((k >> 3) * kR32 + row) << 4) + ((k & 7) << 1); }
// a quad: columns c .. c+3 (c % 4 == 0) of one row
__device__ __forceinline__ void stq32(char* T, int ps, int c, int row, const f32x4v& v) {
  unsigned a1, a2, a3, b1, b2, b3;
  rt_split3(v[0], v[1], a1, a2, a3);
  rt_split3(v[2], v[3], b1, b2, b3);
  char* d = T + pofs(c, row);
  *reinterpret_cast<rt_u32x2*>(d) = rt_u32x2{a1, b1};
  *reinterpret_cast<rt_u32x2*>(d + ps) = rt_u32x2{a2, b2};
  *reinterpret_cast<rt_u32x2*>(d + 2 * ps) = rt_u32x2{a3, b3};
}
__device__ __forceinline__ f32x4v ldq32(const char* T, int ps, int c, int row) {
  const char* s = T + pofs(c, row);
  f32x4v v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 2; p >= 0; --p) {  // smallest part first: the sum is exact either way
    const rt_u32x2 u = *reinterpret_cast<const rt_u32x2*>(s + p * ps);
    v[0] += __builtin_bit_cast(float, u[0] << 16);
    v[1] += __builtin_bit_cast(float, u[0] & 0xffff0000u);
    v[2] += __builtin_bit_cast(float, u[1] << 16);
    v[3] += __builtin_bit_cast(float, u[1] & 0xffff0000u);
  }
  return v;
}
// single elements (the element-wise phases: samples, start states)
__device__ __forceinline__ void st1_32(char* T, int ps, int k, int row, float x) {
  unsigned p1, p2, p3;
  rt_split3(x, 0.f, p1, p2, p3);
  unsigned short* d = reinterpret_cast<unsigned short*>(T + pofs(k, row));
  d[0] = (unsigned short)p1;
  d[ps >> 1] = (unsigned short)p2;
  d[ps] = (unsigned short)p3;
}
__device__ __forceinline__ float ld1_32(const char* T, int ps, int k, int row) {
  const unsigned short* s = reinterpret_cast<const unsigned short*>(T + pofs(k, row));
  return __builtin_bit_cast(float, (unsigned)s[ps] << 16) + __builtin_bit_cast(float, (unsigned)s[ps >> 1] << 16) +
         __builtin_bit_cast(float, (unsigned)s[0] << 16);
}

// ---- weight pack.  kind 0: W(n, k) = src[n * sn + k * sk] -> planes, [kb][plane][k-half][NP][8 bf16], NP = pad32(N),
// kb < pad16(K) / 16, zero outside (n < N, k < K);  kind 1: a vector of N floats copied, zero-padded to pad32(N)
struct Pack32Job {
  const float* src;
  char* dst;
  int N, K, sn, sk, kind;
  const float* src2;  // kind 1: added to src (the GRU's b_ih + b_hh of the r and z gates), or null
};
constexpr int kMaxJobs32 = 28;
struct Pack32Args {
  Pack32Job job[kMaxJobs32];
  int njobs;
};
static inline size_t pack32_bytes(int64_t N, int64_t K) { return (size_t)96 * (pad16((int)K) >> 4) * pad32((int)N); }
static inline size_t vec32_bytes(int64_t N) { return (size_t)4 * pad32((int)N); }
static __global__ __launch_bounds__(256) void pack32_kernel(Pack32Args a) {
  const Pack32Job j = a.job[blockIdx.y];
  const int NP = pad32(j.N);
  if (j.kind == 1) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < NP; i += gridDim.x * 256)
      reinterpret_cast<float*>(j.dst)[i] = i < j.N ? j.src[i] + (j.src2 ? j.src2[i] : 0.f) : 0.f;
    return;
  }
  const int units = (pad16(j.K) >> 3) * NP;  // (kb, g, n): 8 consecutive k of one column
  for (int i = blockIdx.x * 256 + threadIdx.x; i < units; i += gridDim.x * 256) {
    const int n = i % NP, kg = i / NP;  // kg = 2 kb + g
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * kg + e;
      x[e] = (n < j.N && k < j.K) ? j.src[(size_t)n * j.sn + (size_t)k * j.sk] : 0.f;
    }
    unsigned p[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) rt_split3(x[2 * e], x[2 * e + 1], p[0][e], p[1][e], p[2][e]);
    const int kb = kg >> 1, g = kg & 1;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<rt_u32x4*>(j.dst + ((size_t)((kb * 3 + pl) * 2 + g) * NP + n) * 16) =
          rt_u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
  }
}
static inline int launch_pack32(Pack32Args& pa, hipStream_t s) {
  int mx = 1;
  for (int i = 0; i < pa.njobs; ++i) {
    const int units = pa.job[i].kind ? pad32(pa.job[i].N) : (pad16(pa.job[i].K) >> 3) * pad32(pa.job[i].N);
    const int blocks = (units + 255) / 256;
    if (blocks > mx) mx = blocks;
  }
  if (mx > 64) mx = 64;
  hipLaunchKernelGGL(pack32_kernel, dim3(mx, pa.njobs), dim3(256), 0, s, pa);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// ---- weight stream of ONE 32-column tile: a window of PD blocks in registers, opened (first PD blocks requested)
// one stage before its use, kept PD blocks ahead by the run.  Addresses: lane part v (VGPR) + scalar offset.
template <int PD>
struct WWin32 {
  rt_bf16x8 f[PD][3];
  f32x4v bias[4];        // the tile's bias quads, requested with the window: the accumulator starts from them
  unsigned v;            // this lane's byte offset of the NEXT block to request (pack offset folded in)
  unsigned pstr, bstr;   // plane / block strides in bytes
  bool act;
};
__device__ __forceinline__ rt_bf16x8 wld32(__amdgpu_buffer_rsrc_t rw, unsigned v, unsigned so) {
  return __builtin_bit_cast(rt_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, v, so, 0));
}
// request the stream's next block into window slot `slot`
template <int PD>
__device__ __forceinline__ void wnext32(WWin32<PD>& w, __amdgpu_buffer_rsrc_t rw, int slot) {
  w.f[slot][0] = wld32(rw, w.v, 0);
  w.f[slot][1] = wld32(rw, w.v, w.pstr);
  w.f[slot][2] = wld32(rw, w.v, 2 * w.pstr);
#ifndef RT32_NO_WLOAD  // (ablation build: every block reads the stream's first one, an L1 hit)
  w.v += w.bstr;
#endif
}
// tile: the column tile (clamped by the caller to an existing one: an idle wave loads too -- conditional stores into
// the window would make the compiler keep it in scratch memory).  W / B: byte offsets of the layer's weight pack and
// of its bias vector (fp32, zero-padded to pad32(N)) in the pack buffer.
// The block addresses advance in a VECTOR register made opaque here: as scalar offsets derived from the loop-invariant
// W the compiler hoists every one of them out of the step loop (several hundred scalars, spilled to vector lanes and
// read back with v_readlane at each use).
template <int NBLK, int PD, bool BIAS = true>
__device__ __forceinline__ void wopen32(WWin32<PD>& w, __amdgpu_buffer_rsrc_t rw, bool act, unsigned W, unsigned B, int N,
                                        int tile, int lane) {
  const int NP = pad32(N);
  w.act = act;
  unsigned v = W + 16u * (unsigned)((lane >> 5) * NP + tile * 32 + (lane & 31));
  asm volatile("" : "+v"(v));
  w.v = v;
  w.pstr = 32u * (unsigned)NP;
  w.bstr = 96u * (unsigned)NP;
#pragma unroll
  for (int b = 0; b < (NBLK < PD ? NBLK : PD); ++b) wnext32(w, rw, b);
  if (BIAS) {
    unsigned bv = B + 4u * (unsigned)(tile * 32 + 4 * (lane >> 5));
    asm volatile("" : "+v"(bv));
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w.bias[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rw, bv + 32u * i, 0, 0));
  }
}
// fp32 quads in LDS (gradient carries, element-wise operands): columns c .. c+3 of one row are 16 bytes at
// ((c / 4) * 32 + row) * 16 -- a lane's accumulator quad is one conflict-free ds_read / ds_write_b128
__device__ __forceinline__ int fq(int c, int row) { return ((c >> 2) * kR32 + row) * 4 + (c & 3); }
template <int PD>
__device__ __forceinline__ f32x16v bias_acc(const WWin32<PD>& w) {
  f32x16v c;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[4 * i + r] = w.bias[i][r];
  return c;
}
// six partial products, smallest first
__device__ __forceinline__ void mfma6(f32x16v& c, const rt_bf16x8 (&w)[3], const rt_bf16x8 (&x)[3]) {
#ifdef RT32_ONE_MFMA  // ablation build (tools/build_variant.sh): bf16-accurate results, time meaningful
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], c, 0, 0, 0);
  asm volatile("" ::"v"(w[1]), "v"(w[2]), "v"(x[1]), "v"(x[2]));
  return;
#endif
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], c, 0, 0, 0);
}
// this lane's B-fragment offset inside a plane (block 0); a block is 1024 bytes further
__device__ __forceinline__ int xfrag32(int lane) { return ((lane >> 5) * kR32 + (lane & 31)) << 4; }

// c += (X * stream w)^T over NBLK 16-k blocks.  X: plane 0 of an activation tile + xfrag32(lane).  The B fragments of
// block b + 1 are requested before the MFMAs of block b (an LDS round trip is ~1/2 of a block's MFMA time).
// NB1 > 0: a stream whose K range continues in a second tile (blocks NB1 .. NBLK-1 read X1: the GRU's [e | h]).
template <int NBLK, int PD, int NB1 = 0>
__device__ __forceinline__ void wrun32(f32x16v& c, const char* X, int ps, WWin32<PD>& w, __amdgpu_buffer_rsrc_t rw,
                                       const char* X1 = nullptr, int ps1 = 0) {
  rt_bf16x8 xc[3], xn[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) xc[p] = *reinterpret_cast<const rt_bf16x8*>(X + p * ps);
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    if (b + 1 < NBLK) {
      const bool second = NB1 > 0 && b + 1 >= NB1;
      const char* src = second ? X1 + (b + 1 - NB1) * 1024 : X + (b + 1) * 1024;
      const int pss = second ? ps1 : ps;
#pragma unroll
      for (int p = 0; p < 3; ++p) xn[p] = *reinterpret_cast<const rt_bf16x8*>(src + p * pss);
    }
    mfma6(c, w.f[b % PD], xc);
    if (b + PD < NBLK) wnext32(w, rw, b % PD);
#pragma unroll
    for (int p = 0; p < 3; ++p) xc[p] = xn[p];
    __builtin_amdgcn_sched_barrier(0);  // keep the window: no hoisting of later blocks' loads
  }
}

// An activation tile -> its row-major fp32 matrix (rows x ncols, ncols % 4 == 0), by ONE wave: eight lanes take 32
// consecutive columns of one row (128 contiguous bytes), a store instruction eight rows.  The compute waves'
// vector-memory queue then holds loads only (on gfx9 a load issued behind a store is not seen complete before the
// store is acknowledged), and the CU's address path sees 128-byte runs instead of the accumulators' 32-byte ones.
// row0: the tile's first global row; nr: its valid rows; ld: floats per global row
// g0 / gstep: this wave takes column groups g0, g0 + gstep, ... (a layer that leaves several waves idle shares the tile out)
__device__ __forceinline__ void store_tile32(const char* T, int ps, int ncols, const float* dst, unsigned row0,
                                             int nr, unsigned ld, unsigned soff, int lane, int g0 = 0, int gstep = 1) {
#ifdef RT32_NO_TILESTORE
  return;
#endif
  // the descriptor is built here, not kept in scalar registers across the step loop; 2 GB window: a lane without a
  // row stores out of range (dropped: no branch around the stores)
  unsigned long long da = reinterpret_cast<unsigned long long>(dst);
  unsigned dlo = (unsigned)da, dhi = (unsigned)(da >> 32);
  asm volatile("" : "+s"(dlo), "+s"(dhi));
  dlo = __builtin_amdgcn_readfirstlane(dlo), dhi = __builtin_amdgcn_readfirstlane(dhi);
  const __amdgpu_buffer_rsrc_t q = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<float*>(((unsigned long long)dhi << 32) | dlo), 0, 0x7fffffff, 0x00020000);
  const int piece = lane & 7, rsub = lane >> 3;
  unsigned vo[4];
  const char* src[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int row = rg * 8 + rsub;
    vo[rg] = row < nr ? 4u * ((row0 + (unsigned)row) * ld + 4u * (unsigned)piece) : 0x80000000u;
    src[rg] = T + pofs(4 * piece, row);
  }
  // column group g: the quad at columns 32 g + 4 piece; its planes are 2048 g bytes further in the tile
  const int ng = (ncols + 31) >> 5;
  const bool mine_last = (ng - 1) * 32 + 4 * piece < ncols;  // the last group may be ragged (ncols % 4 == 0)
  if (g0 >= ng) return;
  f32x4v cur[4], nxt[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) cur[rg] = ldq32(src[rg] + 2048 * g0, ps, 0, 0);
  for (int g = g0; g < ng; g += gstep) {
    if (g + gstep < ng) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) nxt[rg] = ldq32(src[rg] + 2048 * (g + gstep), ps, 0, 0);
    }
    const bool ok = g + 1 < ng || mine_last;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, cur[rg]), q,
                                             ok ? vo[rg] + 128u * (unsigned)g : 0x80000000u, soff, 0);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) cur[rg] = nxt[rg];
  }
}

// quad i of a 32 x 32 accumulator: columns n0 + 8 i + 4 (lane / 32) .. + 3 of row lane % 32
__device__ __forceinline__ f32x4v quad(const f32x16v& c, int i) {
  return f32x4v{c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]};
}
__device__ __forceinline__ void set_quad(f32x16v& c, int i, const f32x4v& v) {
  c[4 * i] = v[0], c[4 * i + 1] = v[1], c[4 * i + 2] = v[2], c[4 * i + 3] = v[3];
}
__device__ __forceinline__ f32x16v zero16() {
  f32x16v c;
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  return c;
}

}  // namespace repo
