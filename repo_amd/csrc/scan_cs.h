// Column-split, weight-stationary observe scan (scan_cs.hip): the entry points rssm.hip dispatches to.
#pragma once
#include "common.h"

namespace repo {

// floats of workspace the column-split forward scan needs for (B, A, D, Hd, S): weight packs, the exchange buffers
// of ceil(B/16) row groups, their flags and one error word
size_t scan_cs_fwd_ws_floats(int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S);
bool scan_cs_ok(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S);

struct ScanCsFwd {
  int64_t T, B, A, D, Hd, S, E;
  const float* const* params;  // the 14 TransitionModel tensors, state_dict order
  const float *prev_belief, *prev_state, *actions, *nonterms, *eemb;
  NoiseSrc eps_post;
  float min_std;
  float *featx, *post_mean, *post_std, *xsa, *e, *gates, *hq;
};
// packs the weights, clears the flags and launches the scan; eemb (the hoisted embed product) must be complete on `s`
int scan_cs_fwd(const ScanCsFwd& a, void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace repo
