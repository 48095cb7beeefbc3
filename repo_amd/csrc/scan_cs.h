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
  unsigned* status;  // nullable: the caller's sticky status word (REPO_SCAN_STATUS_* bits)
};
// packs the weights, clears the flags and launches the scan; eemb (the hoisted embed product) must be complete on `s`
int scan_cs_fwd(const ScanCsFwd& a, void* ws, size_t ws_bytes, hipStream_t s);

// ---- reverse scan (posterior path; the prior head's share of d belief_t comes in as `dbx`, computed for all steps
// by the caller: it is off the recurrence)
struct ScanCsBwd {
  int64_t T, B, A, D, Hd, S, E;
  const float* const* params;
  const float* nonterms;
  NoiseSrc eps_post;
  float min_std;
  const float *featx, *post_std, *e, *gates, *hq;  // saved by the forward scan
  const float *dfeat, *dqm, *dqs, *dbx;            // upstream, each nullable: (T,B,D+S), (T,B,S) x2, (T,B,D)
  float *doutq, *dhq, *dgi, *dgh, *de;             // per-step deltas: (T,B,2S) (T,B,Hd) (T,B,3D) x2 (T,B,D)
  float *dprev_belief, *dprev_state;               // nullable
  unsigned* status;                                // nullable, as in ScanCsFwd
};
// polls a spin-wait makes before giving up (1 << 22 unless repo_debug_scan_spin_limit changed it)
int scan_cs_spin_limit();
size_t scan_cs_bwd_ws_floats(int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S);
int scan_cs_bwd(const ScanCsBwd& a, void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace repo
