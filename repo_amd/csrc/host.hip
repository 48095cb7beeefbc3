// Host-side helper of the replay path (no device code): the gather of a batch's rows out of the replay ring.
//
// Reference: SequenceReplayBuffer._get_samples (/root/reference/common/buffers.py:186-191) -- `ring[batch_inds]`, a
// single-threaded NumPy fancy-index of 2500 scattered 12 KB frames (30.7 MB) per batch, 10+ ms: longer than the
// update it feeds.  Here the rows are copied by a few host threads straight into the page-locked staging slot the
// hipMemcpyAsync reads from; ctypes releases the GIL for the call, so the Python thread that enqueues the update's
// kernels is not held up either.
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/repo_hip.h"

extern "C" int repo_host_gather_rows(const void* src, int64_t src_rows, int64_t row_bytes, const int64_t* idx,
                                     int64_t n, void* dst, int nthreads) {
  if (!src || !idx || !dst || src_rows <= 0 || row_bytes <= 0 || n < 0) return REPO_E_BADARG;
  for (int64_t i = 0; i < n; ++i)
    if (idx[i] < 0 || idx[i] >= src_rows) return REPO_E_SHAPE;  // NumPy raises IndexError; never copy out of range
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  // small batches (the action / reward / done fields: a few bytes per row) are not worth a thread
  if ((int64_t)nthreads > n || n * row_bytes < (1 << 20)) nthreads = 1;
  const char* s = (const char*)src;
  char* d = (char*)dst;
  auto work = [=](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) memcpy(d + i * row_bytes, s + idx[i] * row_bytes, (size_t)row_bytes);
  };
  if (nthreads == 1) {
    work(0, n);
    return REPO_OK;
  }
  std::vector<std::thread> th;
  th.reserve(nthreads - 1);
  const int64_t per = (n + nthreads - 1) / nthreads;
  for (int t = 1; t < nthreads; ++t) {
    const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  work(0, per < n ? per : n);
  for (auto& t : th) t.join();
  return REPO_OK;
}
