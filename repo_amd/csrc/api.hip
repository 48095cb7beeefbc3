// ABI bookkeeping entry points.
#include "common.h"

#include <atomic>
#include <string.h>

extern "C" int repo_abi_version(void) { return REPO_ABI_VERSION; }

namespace repo {
// 0 = not queried yet, 1 = gfx950, 2 = some other architecture.  Written once per device ordinal; the only
// mutable global state of the library ("once-initialised per-device constants").
static std::atomic<int> g_arch[64];

static int query_arch(int dev) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return (int)e;
  // gcnArchName is e.g. "gfx950:sramecc+:xnack-"
  const bool ok = strncmp(prop.gcnArchName, "gfx950", 6) == 0 &&
                  (prop.gcnArchName[6] == '\0' || prop.gcnArchName[6] == ':');
  return ok ? REPO_OK : REPO_E_ARCH;
}

int arch_status() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= 64) return query_arch(dev);
  const int seen = g_arch[dev].load(std::memory_order_relaxed);
  if (seen == 1) return REPO_OK;
  if (seen == 2) return REPO_E_ARCH;
  const int rc = query_arch(dev);
  if (rc == REPO_OK) g_arch[dev].store(1, std::memory_order_relaxed);
  else if (rc == REPO_E_ARCH) g_arch[dev].store(2, std::memory_order_relaxed);
  return rc;
}
}  // namespace repo

extern "C" int repo_device_check(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return (int)e;
  if (device < 0 || device >= n) return REPO_E_BADARG;
  return repo::query_arch(device);
}

extern "C" const char* repo_strerror(int code) {
  switch (code) {
    case REPO_OK: return "ok";
    case REPO_E_BADARG: return "bad argument (null pointer or unsupported option)";
    case REPO_E_SHAPE: return "unsupported shape or index range";
    case REPO_E_ALIGN: return "misaligned pointer";
    case REPO_E_WS_TOO_SMALL: return "workspace too small";
    case REPO_E_ARCH: return "device is not gfx950";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

namespace repo {
int philox_fill(float* out, long n, uint64_t seed, uint64_t offset, hipStream_t s);  // imagine.hip
}
extern "C" int repo_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(n >= 0, REPO_E_SHAPE);
  REPO_REQUIRE(out || n == 0, REPO_E_BADARG);
  return repo::philox_fill(out, (long)n, seed, offset, stream);
}

// Debug aid for the parity tests: fill the LDS of every CU with NaN bit patterns, so that a kernel which
// reads LDS it never wrote (and, say, multiplies it by zero) shows up deterministically instead of once
// in a dozen runs.  LDS contents persist until a later workgroup on that CU overwrites them.
namespace repo {
__global__ __launch_bounds__(256) void poison_lds_kernel(int words, unsigned* sink) {
  extern __shared__ unsigned plds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) plds[i] = 0x7fc00000u + (unsigned)i;
  __syncthreads();
  if (plds[(threadIdx.x * 7919) % words] == 0u && sink) *sink = 1u;  // keep the stores alive
}
}  // namespace repo

extern "C" int repo_debug_poison_lds(hipStream_t stream) {
  REPO_ARCH_GUARD();
  const int bytes = 160 * 1024;
  hipError_t e = hipFuncSetAttribute((const void*)repo::poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     bytes);
  if (e != hipSuccess) return (int)e;
  // one workgroup occupies a whole CU's LDS; several waves of workgroups cover all 256 CUs
  hipLaunchKernelGGL(repo::poison_lds_kernel, dim3(1024), dim3(256), bytes, stream, bytes / 4, (unsigned*)nullptr);
  e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// One update's copy of the scans' sticky status word, taken and cleared in one launch (stream-ordered behind the scans).
namespace repo {
__global__ void take_status_kernel(unsigned* __restrict__ sticky, unsigned* __restrict__ taken) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *taken = *sticky;
    *sticky = 0u;
  }
}
}  // namespace repo
extern "C" int repo_take_status(unsigned* sticky, unsigned* taken, hipStream_t stream) {
  REPO_ARCH_GUARD();
  REPO_REQUIRE(sticky && taken && sticky != taken, REPO_E_BADARG);
  hipLaunchKernelGGL(repo::take_status_kernel, dim3(1), dim3(64), 0, stream, sticky, taken);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? REPO_OK : (int)e;
}

// Test aid: the spin limit of the column-split scans' exchanges (scan_cs.hip reads it per launch).  Like the engine
// switches (repo_debug_bgemm / _bconv / _rowtile32) it is THREAD-LOCAL: a setting belongs to the host thread that
// made it and governs the launches that thread issues afterwards -- another thread (another stream's driver) keeps its
// own, default, setting, so the library has no process-global mutable state beyond the per-device arch cache above.
namespace repo {
static thread_local int t_scan_spin_limit = 1 << 22;
int scan_cs_spin_limit() { return t_scan_spin_limit; }
}  // namespace repo
extern "C" int repo_debug_scan_spin_limit(int polls) {
  const int prev = repo::t_scan_spin_limit;
  repo::t_scan_spin_limit = polls < 0 ? (1 << 22) : polls;
  return prev;
}
