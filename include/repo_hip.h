/* librepo_hip.so -- C ABI of the MI355X (gfx950) kernels behind RePo's world-model +
 * imagination update.
 *
 * The reference (zchuning/repo) has no native/FFI layer: its hot path is PyTorch calls
 * inside Python classes.  Each entry point below therefore cites
 * the reference *call site* whose arithmetic it replaces; the Python classes in
 * repo_amd/ (same names and signatures as the reference's) bind these through ctypes
 * (see INTEGRATION.md for the binding a maintainer of the reference would add).
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless a parameter is documented "host".
 *  - All arithmetic is fp32; replay observations may be uint8 (normalised in-kernel with
 *    the reference's expression ((x/255)*2)-1, common/utils.py:79).
 *  - Activations are NCHW / row-major exactly like the reference's tensors; time-major
 *    (T,B,...) sequences are passed flattened to rows = T*B.
 *  - The library never allocates or frees device memory, never synchronises the stream
 *    and keeps no pointer after returning.  Scratch comes from the caller (workspace).
 *  - Return value: 0 on success, a negative REPO_E_* code for argument errors, a positive
 *    value = raw hipError_t from a launch.  Nothing throws or aborts.
 *  - Reentrant; no global mutable state.  Work is stream-ordered on `stream`.
 */
#ifndef REPO_HIP_H
#define REPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define REPO_ABI_VERSION 8

#define REPO_OK 0
#define REPO_E_BADARG (-1)
#define REPO_E_SHAPE (-2)
#define REPO_E_ALIGN (-3)
#define REPO_E_WS_TOO_SMALL (-4)
#define REPO_E_ARCH (-5)

/* epilogues of repo_gemm */
#define REPO_EPI_NONE 0
#define REPO_EPI_ELU 1       /* F.elu(alpha=1)                                  */
#define REPO_EPI_RELU 2      /* F.relu                                          */
#define REPO_EPI_MUL_DELU 3  /* multiply by elu'(x) given aux = elu(x)          */
#define REPO_EPI_MUL_DRELU 4 /* multiply by relu'(x) given aux = relu(x)        */
#define REPO_EPI_MUL_MASK4 5 /* multiply by relu'(x) given aux = the QUAD MASK of relu(x) (bytes): bit (o & 3) of
                                byte (o >> 2) is relu(x)[o] > 0, o = flat element index; written by
                                repo_decoder_out_nll for the decoder's last hidden activation, read by
                                repo_conv_down (8.8 MB of mask instead of 282 MB of activations at 2450 frames)   */

#define REPO_EPI_MUL_CMASK 6 /* multiply by relu'(x) given aux = the CHANNEL-QUAD MASK of relu(x) (bytes): for an NCHW
                                activation x (n, C, P pixels), byte ((n * C/4 + c/4) * P + p) has bit (c & 3) set iff
                                relu(x)[n][c][p] > 0.  Written by repo_conv_down (relu_cmask: the encoder's forward),
                                read by repo_conv_up on the scatter kernels (layers 1-3: the encoder's data gradients,
                                whose drain owns four channels of a pixel): 19 MB of mask instead of 301 MB of
                                activations for the encoder's first layer at 2450 frames                           */

#define REPO_EPI_FILM_RELU 7 /* FiLM + ReLU of the multitask agents' conv stacks (models/encoder.py:75-88, decoder.py:108-123):
                                relu(scale[n][c] * (x + bias[c]) + shift[n][c]) with aux = this layer's FiLM TABLE, fp32
                                (nimg, 2, C): row n = [scale = 1 + gamma (C) | shift = beta (C)] (repo_film_tables).
                                Accepted by repo_conv_down, repo_conv_up (scatter kernels) and repo_gemm (c = n /
                                bias_div, ldaux = 2 C): the modulation is an epilogue, the pre-FiLM tensor is never
                                written; the backward pass recovers it from the output (repo_film_bwd_h)             */

int repo_abi_version(void);
const char* repo_strerror(int code);
/* REPO_OK if HIP device `device` is gfx950 (MI355X), REPO_E_ARCH if it is another architecture, REPO_E_BADARG
 * for an ordinal that does not exist, a positive hipError_t if the runtime cannot be queried.  Every entry
 * point below makes the same check for the calling thread's current device (once per device, then cached)
 * and returns REPO_E_ARCH before launching anything; the reference's counterpart is the device choice in
 * common/utils.py:10-24 (set_gpu_mode / get_device), which has no notion of an unsupported GPU. */
int repo_device_check(int device);
/* Test aid: fills every CU's LDS with NaN patterns (see tests/test_ops_gpu.py::test_no_uninitialised_lds). */
int repo_debug_poison_lds(hipStream_t stream);
/* *taken = *sticky; *sticky = 0 -- one update's copy of the scans' sticky asynchronous status word (repo_rssm_observe_fwd),
 * stream-ordered behind the scans that may have written it; the copy is what that update's optimiser steps take as
 * `skip_if_nonzero` (repo_clip_adam) and what the caller reads back.  Device pointers, one launch. */
int repo_take_status(unsigned* sticky, unsigned* taken, hipStream_t stream);

/* The four repo_debug_* switches below are TEST AIDS and THREAD-LOCAL: a setting belongs to the calling host thread and
 * governs the launches that thread issues afterwards (it is read when an entry point is called, never by a kernel);
 * other threads -- e.g. the driver of another stream -- keep their own, default, settings.  The library therefore has
 * no process-global mutable state beyond the once-initialised per-device architecture cache.
 * Consequence for callers that split a forward and its backward over host threads (torch.autograd runs a Function's
 * backward on its own device thread): a setting made on the thread that ran the forward does NOT reach the backward.
 * repo_amd's autograd wrappers (algorithms/repo/autograd.py) therefore record the four settings in forward and re-apply
 * them around their backward (ops.debug_snapshot / ops.debug_scope); a workspace size queried under one setting must be
 * consumed under the same one.
 *
 * Polls a spin-wait of the column-split scans makes before it gives up (default 1 << 22; `polls` < 0 restores it);
 * tests/test_rssm_gpu.py::test_scan_timeout_* set 0 to see the status word of repo_rssm_observe_fwd / _bwd raised;
 * returns the previous value. */
int repo_debug_scan_spin_limit(int polls);
/* Test aid: enable (default) / disable the bf16x6 dense engine (csrc/bgemm.h: big products of repo_gemm /
 * repo_gemm_wgrad formed as six exact bf16 partial products per fp32 multiply on the bf16 matrix pipe -- same inputs,
 * outputs and accuracy as the fp32-MFMA engines); for A/B runs; returns the calling thread's previous setting. */
int repo_debug_bgemm(int enable);
/* The same switch for the bf16x6 stride-2 "down" convolution kernel (csrc/bconv.h). */
int repo_debug_bconv(int enable);
/* The same switch for the bf16x6 32-row-tile engines (csrc/rowtile32.h: the imagination rollout and the dense heads);
 * 0 = the 16-row fp32-MFMA engines of rowtile.h. */
int repo_debug_rowtile32(int enable);

/* ------------------------------------------------------------------ reparameterisation noise
 * The reference draws its noise from torch's global generator (torch.randn_like in models/rssm.py:49,61-63;
 * Normal.rsample in models/actor_critic.py:97-102 and models/utils.py:161).  Here every op that consumes noise
 * takes the tensor(s) as explicit inputs -- OR null pointers plus (noise_seed, noise_offset): the kernel then
 * DRAWS standard normals itself from a counter-based generator (Philox4x32-10 keyed by the seed, Box-Muller),
 * normal number noise_offset + i standing in for element i of the tensor (per-op layouts below).  Nothing is
 * written to or read from memory for it, the backward op re-draws the same values from the same (seed, offset),
 * and repo_philox_normal materialises out[i] = normal number offset + i for tests.  The caller owns the
 * counter: advance the offset by the number of normals an op consumed before the next draw. */
int repo_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t stream);

/* ------------------------------------------------------------------ dense layers
 * C[m][n] (+)= epi( sum_k opA(m,k) * opB(k,n) + bias[n / bias_div] )
 *   opA(m,k) = transa ? A[k*lda + m] : A[m*lda + k]
 *   opB(k,n) = transb ? B[n*ldb + k] : B[k*ldb + n]
 * Replaces every nn.Linear forward / autograd backward-data on the path, e.g.
 * F.linear in models/rssm.py:36-64, models/decoder.py:42,191-194,
 * models/actor_critic.py:21-25,77-82 (transa=0, transb=1), their input gradients
 * (transb=0), and the 1x1 -> 5x5 first transposed convolution of the decoder
 * (models/decoder.py:44), which is a plain GEMM against the (1024, 128*25) weight.
 * bias may be NULL; aux (ld = ldaux) is read only by the MUL_* / FILM epilogues.
 * bias_div < 0 (ABI v8, REPO_EPI_FILM_RELU only): the bias is per output column (as bias_div = 1) while the FiLM table's
 * channel of column n is n / -bias_div -- the decoder's COMPOSED first layers (repo_amd/functional.py, dec_head_compose:
 * fc1 and the 1 x 1 -> 5 x 5 transposed conv are linear in sequence, models/decoder.py:41-44) carry one bias per
 * output element and one FiLM pair per 25 of them. */
int repo_gemm(int transa, int transb, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
              const float* B, int64_t ldb, const float* bias, int64_t bias_div, float* C, int64_t ldc,
              int epi, const float* aux, int64_t ldaux, int accumulate, hipStream_t stream);

/* Weight (+bias) gradient of a dense layer:  dW[n][k] (+)= sum_m dY[m][n] * X[m][k],
 * db[n] (+)= sum_m dY[m][n]  (db may be NULL).  Split over rows into deterministic
 * partial slabs in `ws`, then reduced in a fixed order (bitwise reproducible).
 * Replaces autograd's weight/bias gradients of nn.Linear (model_loss.backward(),
 * algorithms/repo/repo.py:88; actor/value backward, dreamer.py:357,371). */
size_t repo_gemm_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K);
int repo_gemm_wgrad(int64_t M, int64_t N, int64_t K, const float* dY, int64_t lddy, const float* X,
                    int64_t ldx, float* dW, int64_t lddw, float* db, int accumulate, void* ws,
                    size_t ws_bytes, hipStream_t stream);

/* ------------------------------------------------------------------ convolutions
 * `layer` selects one of the seven stride-2, no-padding geometries of the reference:
 *   0..3  encoder conv1..4  (models/encoder.py:26-29): 3x64x64 -> 32x31x31 -> 64x14x14
 *                                                     -> 128x6x6 -> 256x2x2, k=4
 *   4..6  decoder conv2..4  (models/decoder.py:37-39): 128x5x5 -> 64x13x13 (k5)
 *                                                     -> 32x30x30 (k6) -> 3x64x64 (k6)
 * or of the BUILD-DEFINED 128 x 128 stack (BASELINE config 4's frame size; the reference's encoder
 * hard-codes the 64 x 64 flatten, encoder.py:39, so there is no reference model at this size):
 *   7..10 encoder conv1..4 at 128: 3x128x128 -> 32x63x63 -> 64x30x30 -> 128x14x14 -> 256x6x6, k=4
 *   11,12 decoder conv4, conv5 at 128 (after layers 4, 5): 32x30x30 -> 16x64x64 (k6) -> 3x128x128 (k2)
 *   13    TIAObservationModel.conv4 (models/decoder.py:165): 32x30x30 -> 6x64x64 (k6), [recon(3) | mask(3)]
 * Every geometry is a pair (big, small) with big = 2*small + k - 2 and one weight tensor
 * indexed w[small_ch][big_ch][ky][kx] -- which is both nn.Conv2d's (out,in,kh,kw) for the
 * encoder and nn.ConvTranspose2d's (in,out,kh,kw) for the decoder.  Three kernels act on a
 * pair:
 *   down : small = gather(big)      encoder forward / decoder backward-data
 *   up   : big   = scatter(small)   decoder forward / encoder backward-data
 *   wgrad: dw = corr(small, big)    both
 * epi: REPO_EPI_NONE (+bias if given), REPO_EPI_RELU (+bias), REPO_EPI_MUL_DRELU (aux: fp32, the
 * output's shape), REPO_EPI_MUL_MASK4 (aux: the output's quad mask, numel/4 bytes).
 * `big_is_u8` != 0: big is uint8 pixels (encoder conv1 only).
 * dbias_small (nullable, [small_ch]): (+)= the sum over images and pixels of the values written to `small` -- when
 * `small` is the pre-activation gradient of the transposed-conv layer below (decoder backward), that is that layer's
 * bias gradient, taken while the values are in registers instead of by a second pass over the tensor
 * (repo_channel_sum).  Needs ws >= repo_conv_down_workspace_bytes(layer, nimg) then (per-workgroup partial sums,
 * reduced in a fixed order: bit-reproducible).
 * ws also holds the weight pack of the bf16x6 kernel (csrc/bconv.h: the MFMA-bound layers with an even row pitch --
 * 2, 3, 5 -- form their products as six exact bf16 partial products per fp32 multiply on the bf16 matrix pipe, same
 * accuracy); a call without ws (or with too little) runs the fp32-MFMA kernel.
 * relu_cmask (nullable; epi = REPO_EPI_RELU, small_ch % 4 == 0): nimg * small_ch / 4 * small_pixels bytes, the
 * output's channel-quad mask (REPO_EPI_MUL_CMASK) -- taken from the accumulators on their way out. */
size_t repo_conv_down_workspace_bytes(int layer, int64_t nimg);
int repo_conv_down(int layer, int64_t nimg, const void* big, int big_is_u8, const float* w,
                   const float* bias, float* small, int epi, const void* aux, float* dbias_small,
                   int accumulate_dbias, unsigned char* relu_cmask, void* ws, size_t ws_bytes, hipStream_t stream);
/* `up` works from a fragment-ready copy of the layer's weights in `ws` (layers 1..5; 0 bytes for the 3-channel
 * layers 0 and 6): at least repo_conv_up_workspace_bytes(layer).  repo_conv_up_pack writes that copy; repo_conv_up
 * writes it itself first unless ws_is_packed != 0 (the weights change once per optimiser step, not per call). */
size_t repo_conv_up_workspace_bytes(int layer);
int repo_conv_up_pack(int layer, const float* w, void* ws, size_t ws_bytes, hipStream_t stream);
int repo_conv_up(int layer, int64_t nimg, const float* small, const float* w, const float* bias,
                 float* big, int epi, const float* aux, int ws_is_packed, void* ws, size_t ws_bytes,
                 hipStream_t stream);
/* dw (+)= ..., dbias_small[small_ch] (+)= sum over images and pixels of `small` (NULL to skip); dbias_big[big_ch] (+)=
 * the same of `big` (NULL to skip; f32 `big` only) -- the bias gradient of a transposed conv, whose output gradient is
 * the `big` operand here (models/decoder.py:43-47).  Where the weight-gradient engine stages every element of `big`
 * exactly once (decoder conv3) the sums ride along; elsewhere they are the channel-sum pass of repo_channel_sum. */
size_t repo_conv_wgrad_workspace_bytes(int layer, int64_t nimg);
int repo_conv_wgrad(int layer, int64_t nimg, const float* small, const void* big, int big_is_u8,
                    float* dw, float* dbias_small, float* dbias_big, int accumulate, void* ws, size_t ws_bytes,
                    hipStream_t stream);

/* ------------------------------------------------------------------ replay batch gather (HOST memory, no GPU work)
 * dst[i] = src[idx[i]] for n rows of row_bytes bytes, copied by `nthreads` host threads -- the reference's
 * `ring[batch_inds]` (SequenceReplayBuffer._get_samples, common/buffers.py:186-191) written straight into the
 * page-locked staging slot the asynchronous host-to-device copy reads from.  REPO_E_SHAPE if an index is outside
 * [0, src_rows).  Needs no device; safe to call concurrently with kernel launches from another thread. */
int repo_host_gather_rows(const void* src, int64_t src_rows, int64_t row_bytes, const int64_t* idx,
                          int64_t n, void* dst, int nthreads);

/* Final decoder layer fused with the pixel likelihood (models/decoder.py:47 +
 * -Normal(recon,1).log_prob(obs[1:]).sum((2,3,4)).mean((0,1)), repo.py:46-53):
 *   recon = up(layer 6)(h3) + bias;  d = recon - target
 *   dpre[img][c][y][x] = d * grad_scale          (gradient w.r.t. recon of the mean loss)
 *   loss_sum += 0.5*d*d   (per-workgroup partials in ws, reduced into *loss_sum in order)
 * recon and dpre may each be NULL.  target is uint8 (target_is_u8) or fp32 in [-1,1].
 * relu_mask4 (nullable, nimg*32*900/4 bytes): the quad mask of h3 (REPO_EPI_MUL_MASK4) -- the kernel has every
 * element of h3 in registers on its way to LDS anyway, and the layer's data gradient (repo_conv_down, layer 6)
 * then needs h3 for nothing else.
 * dbias (nullable, 3 floats): (+)= the channel sums of dpre -- the output layer's bias gradient (accumulate_dbias != 0
 * adds): the kernel has every d in registers; the separate pass re-read 120 MB of dpre at 2450 frames. */
size_t repo_decoder_out_nll_workspace_bytes(int64_t nimg);
int repo_decoder_out_nll(int64_t nimg, const float* h3, const float* w, const float* bias,
                         const void* target, int target_is_u8, float grad_scale, float* recon,
                         float* dpre, unsigned char* relu_mask4, float* loss_sum, float* dbias,
                         int accumulate_dbias, void* ws, size_t ws_bytes, hipStream_t stream);

/* A 3-channel transposed conv fused with the pixel likelihood on the gather engine, any output size: layer 12 (the
 * 128 x 128 stack's output layer) or 6 (the reference's: repo_decoder_out_nll is the specialised kernel for it).
 * Same outputs as repo_decoder_out_nll without the mask. */
size_t repo_conv_up_nll_workspace_bytes(int layer, int64_t nimg);
int repo_conv_up_nll(int layer, int64_t nimg, const float* small, const float* w, const float* bias,
                     const void* target, int target_is_u8, float grad_scale, float* recon, float* dpre,
                     float* loss_sum, void* ws, size_t ws_bytes, hipStream_t stream);

/* out[c] (+)= sum over n and p of x[n][c][p]   (bias gradient of an NCHW activation) */
size_t repo_channel_sum_workspace_bytes(int64_t nimg, int64_t C, int64_t P);
int repo_channel_sum(int64_t nimg, int64_t C, int64_t P, const float* x, float* out, int accumulate,
                     void* ws, size_t ws_bytes, hipStream_t stream);

/* y = dy * (h > 0)  (ReLU backward through a saved output), n elements */
int repo_relu_mask(int64_t n, const float* dy, const float* h, float* y, hipStream_t stream);

/* ------------------------------------------------------------------ RSSM observe scan
 * Fused per-timestep GRU + prior/posterior cell, persistent over the T steps
 * (TransitionModel.observe with observations and nonterminals,
 * models/rssm.py:76-146; cell = rssm.py:34-64).  rows are time-major: row = t*B + b.
 *
 * params: HOST array of 14 device pointers in the module's state_dict order:
 *   0 fc_embed_state_action.weight (D,S+A)  1 .bias      2 rnn.weight_ih (3D,D)
 *   3 rnn.weight_hh (3D,D)   4 rnn.bias_ih   5 rnn.bias_hh
 *   6 fc_embed_belief_prior.weight (Hd,D)   7 .bias      8 fc_state_prior.weight (2S,Hd)  9 .bias
 *  10 fc_embed_belief_posterior.weight (Hd,D+E)  11 .bias  12 fc_state_posterior.weight (2S,Hd)  13 .bias
 * Inputs : prev_belief (B,D), prev_state (B,S), actions (T,B,A), nonterms (T,B),
 *          embeds (T,B,E), eps_prior / eps_post (T,B,S) standard-normal noise in the
 *          reference's draw order (rssm.py:49,61-63); both NULL => drawn in-kernel: eps_prior = normals
 *          [noise_offset, +T*B*S), eps_post = the next T*B*S (2*T*B*S consumed).
 * Outputs: featx (T+1,B,D+S): slot 0 = [prev_belief|prev_state], slot t+1 = [belief_t|post_t]
 *          (so beliefs = featx[1:,:,:D], posterior_states = featx[1:,:,D:]);
 *          prior_state/mean/std, post_mean/std (T,B,S).
 * Saved for backward: xsa (T,B,S+A), e (T,B,D), gates (T,B,4D) = r|z|n|W_hn h+b_hn,
 *          hp, hq (T,B,Hd); eemb (T,B,Hd) is scratch for the hoisted embedding GEMM.
 * prior_only == 2: the scan leaves the prior head out (repo_rssm_prior_head below computes it for all steps).
 * prior_only == 3: as 2, on the COLUMN-SPLIT, WEIGHT-STATIONARY engine (csrc/scan_cs.hip): ceil(D/16) workgroups per
 *          16 batch rows each keep their 16-column slices of W_ih / W_hh / W_bq (and the small replicated layers) in
 *          REGISTERS for all T steps, the row tile runs on v_mfma_f32_16x16x4_f32 and the belief / posterior-hidden
 *          activations are all-gathered through L2 twice per step (sc1 write-through stores, data-tagged granules).
 *          ~8.5 us per step in the kernel for any B <= 64 (the row scan: 17-19); shapes D, Hd in (192, 208], S + A in
 *          (32, 48], S <= 32 only (else REPO_E_SHAPE).  The workspace must not be shared with a concurrent call.
 * prior_only == 1: the reference's `observations=None` branch (rssm.py:118): step t+1 is fed the PRIOR sample of
 *          step t, featx[t+1][D:] = prior sample; the posterior outputs are then computed from whatever
 *          `embeds` holds and mean nothing (forward only: repo_rssm_observe_bwd assumes prior_only == 0).
 * status: ASYNCHRONOUS errors.  A launch error is this call's return value; what can only be known once the kernel runs
 *          is reported through `status`, a caller-owned device word (nullable; zero it once, it is sticky): the
 *          column-split engine's exchanges are cross-workgroup spin-waits, and a group whose peers do not answer within
 *          the spin limit (2^22 polls; a peer workgroup that never became resident) ORs REPO_SCAN_STATUS_FWD_TIMEOUT
 *          (the reverse scan: REPO_SCAN_STATUS_BWD_TIMEOUT) into it, writes NaN into its outputs and leaves -- it never
 *          hangs.  The caller reads the word whenever it next copies results to the host (the agents: inside their one
 *          per-update scalar copy, algorithms/repo/dreamer.py `_log_update`, raising RepoHipError).  The row-scan
 *          engine has no cross-workgroup waits and never touches it. */
#define REPO_SCAN_STATUS_FWD_TIMEOUT 1u
#define REPO_SCAN_STATUS_BWD_TIMEOUT 2u
size_t repo_rssm_observe_fwd_workspace_bytes(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd,
                                             int64_t S, int64_t E);
int repo_rssm_observe_fwd(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S, int64_t E,
                          const float* const* params, const float* prev_belief,
                          const float* prev_state, const float* actions, const float* nonterms,
                          const float* embeds, const float* eps_prior, const float* eps_post,
                          uint64_t noise_seed, uint64_t noise_offset,
                          float min_std, float* featx, float* prior_state, float* prior_mean,
                          float* prior_std, float* post_mean, float* post_std, float* xsa, float* e,
                          float* gates, float* hp, float* hq, float* eemb, int prior_only, unsigned* status,
                          void* ws, size_t ws_bytes, hipStream_t stream);

/* The prior head (fc_embed_belief_prior, fc_state_prior, softplus + sample: rssm.py:42-50) of ALL T steps at once.
 * It depends on belief_t only, i.e. it is off the recurrence: repo_rssm_observe_fwd(prior_only = 2) leaves it out of
 * the scan (7-10 % of the weights a step streams), and this call -- two (T*B)-row GEMMs and an elementwise kernel --
 * can run on another stream beside whatever consumes the posterior first (the decoder).  featx as written by the
 * scan; hp (T,B,Hd) and the three prior tensors (T,B,S) as repo_rssm_observe_fwd would have written them (the same
 * noise: eps_prior, or the Philox stream (seed, offset) of the scan's call). */
size_t repo_rssm_prior_head_workspace_bytes(int64_t T, int64_t B, int64_t S);
int repo_rssm_prior_head(int64_t T, int64_t B, int64_t D, int64_t Hd, int64_t S, const float* const* params,
                         const float* featx, const float* eps_prior, uint64_t noise_seed, uint64_t noise_offset,
                         float min_std, float* hp, float* prior_state, float* prior_mean, float* prior_std, void* ws,
                         size_t ws_bytes, hipStream_t stream);

/* Reverse scan (BPTT) + deferred weight gradients.  Upstream gradients (each nullable):
 * dfeat (T,B,D+S) w.r.t. featx[1:], dprior_state, dpm, dps, dqm, dqs (T,B,S) w.r.t. the
 * prior sample and the four distribution parameters.  dparams: HOST array of 14 device
 * pointers (same order/shapes as params) receiving the gradients ((+)= if accumulate);
 * dembeds (T,B,E), dprev_belief (B,D), dprev_state (B,S) are nullable.
 * Replaces autograd's traversal of the 49-step graph in model_loss.backward()
 * (algorithms/repo/repo.py:88 / dreamer.py:287).
 * `accumulate`: bit 0 = accumulate into dparams; bit 1 (value 2) = run the reverse scan on the column-split,
 * weight-stationary engine (csrc/scan_cs.hip; D, Hd in (192, 208], S + A in (32, 48], S <= 32 only, else
 * REPO_E_SHAPE), the counterpart of repo_rssm_observe_fwd(prior_only = 3). */
size_t repo_rssm_observe_bwd_workspace_bytes(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd,
                                             int64_t S, int64_t E);
int repo_rssm_observe_bwd(int64_t T, int64_t B, int64_t A, int64_t D, int64_t Hd, int64_t S, int64_t E,
                          const float* const* params, const float* nonterms, const float* embeds,
                          const float* eps_prior, const float* eps_post, uint64_t noise_seed,
                          uint64_t noise_offset, float min_std,
                          const float* featx, const float* prior_std, const float* post_std,
                          const float* xsa, const float* e, const float* gates, const float* hp,
                          const float* hq, const float* dfeat, const float* dprior_state,
                          const float* dpm, const float* dps, const float* dqm, const float* dqs,
                          float* const* dparams, float* dembeds, float* dprev_belief,
                          float* dprev_state, int accumulate, unsigned* status, void* ws, size_t ws_bytes,
                          hipStream_t stream);

/* ------------------------------------------------------------------ ELU-MLP heads
 * n_layers nn.Linear layers, ELU between, last layer linear (RewardModel / ValueModel:
 * 4 layers, out_dim 1, models/decoder.py:189-195, models/actor_critic.py:20-26; ActorModel
 * trunk: 5 layers, out_dim 2A, models/actor_critic.py:76-82).  Input rows are [belief|state]
 * (the torch.cat of the reference is a row of the caller's feature buffer, ld = ldx).
 * params / dparams: HOST arrays of 2*n_layers device pointers (fc1.weight, fc1.bias, ...).
 * hidden_out / hidden_acts: HOST arrays of n_layers-1 device pointers to (rows, hidden). */
/* The reference head shapes (in_dim = 230, hidden = 200, 4 or 5 layers, out_dim <= 16) run the whole chain of a row
 * tile in one kernel over packed weights kept in ws (csrc/mlp16.hip); other shapes run layer by layer and need
 * no workspace (the query returns 0). */
size_t repo_mlp_fwd_workspace_bytes(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim,
                                    int n_layers);
int repo_mlp_fwd(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers,
                 const float* x, int64_t ldx, const float* const* params, float* const* hidden_out,
                 float* out, int64_t ldo, void* ws, size_t ws_bytes, hipStream_t stream);
/* dparams NULL: frozen weights (FreezeParameters, dreamer.py:306-317); dx NULL: detached input.
 * dout_w (ABI v8, nullable; out_dim == 1, lddout == 1, dparams and dx given): TWO upstream gradients through ONE reverse
 * chain -- dx is the input gradient of `dout` over all rows, dparams the weight gradients of `dout_w` over the first
 * rows_w <= rows rows.  A scalar head's reverse chain is, per row, the chain of a unit upstream times that row's scalar,
 * so the chain runs once (on upstream 1) and its two products leave scaled by dout[row] / dout_w[row].  The
 * actor-critic update differentiates the value head twice per update -- the actor's loss through the lambda-returns
 * into the imagined states (dreamer.py:343-359, weights frozen) and the critic's own loss into its weights
 * (dreamer.py:362-373, inputs detached): same rows, same activations, two scalars per row. */
size_t repo_mlp_bwd_workspace_bytes(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim,
                                    int n_layers);
int repo_mlp_bwd(int64_t rows, int64_t in_dim, int64_t hidden, int64_t out_dim, int n_layers,
                 const float* x, int64_t ldx, const float* const* params,
                 const float* const* hidden_acts, const float* dout, int64_t lddout,
                 float* const* dparams, int accumulate_w, float* dx, int64_t lddx, int accumulate_dx,
                 const float* dout_w, int64_t rows_w, void* ws, size_t ws_bytes, hipStream_t stream);

/* Actor distribution head (models/actor_critic.py:84-87,89-102): raw (rows,2A) ->
 * mean = mean_scale*tanh(raw_m/mean_scale), std = softplus(raw_s+init_std)+min_std.
 * With eps (rows,A): also action = tanh(mean+std*eps) (TanhBijector rsample) and
 * xsa[row] = [state(row) (S, ld ldstate) | action] -- the torch.cat([state, action]) input of
 * fc_embed_state_action.  Backward maps (dmean,dstd) and/or d action to d raw. */
int repo_actor_head_fwd(int64_t rows, int64_t A, int64_t S, const float* raw, const float* eps,
                        const float* state, int64_t ldstate, float min_std, float init_std,
                        float mean_scale, float* mean, float* std, float* xsa, hipStream_t stream);
int repo_actor_head_bwd(int64_t rows, int64_t A, const float* dmean, const float* dstd,
                        const float* daction, int64_t ldda, const float* action, int64_t ldact,
                        const float* eps, const float* mean, const float* std, float min_std,
                        float mean_scale, float* draw, int accumulate, hipStream_t stream);

/* ------------------------------------------------------------------ imagination rollout
 * TransitionModel.imagine(prev_belief, prev_state, actor, horizon) (models/rssm.py:148-184)
 * with policy.get_action = rsample of the tanh-Normal on DETACHED inputs (rssm.py:170,
 * actor_critic.py:97-102).  Hm = horizon-1 steps over N independent rows.
 * rssm_params: 14 pointers as in repo_rssm_observe_fwd; actor_params: 2*n_actor_layers.
 * eps_act (Hm,N,A) then eps_prior (Hm,N,S) per step, in the reference's draw order; both NULL => drawn
 * in-kernel: eps_act = normals [noise_offset, +Hm*N*A), eps_prior = the next Hm*N*S.
 * Outputs: featx (Hm+1,N,D+S), slot 0 = start, slot t+1 = [belief|prior sample];
 *          prior_mean/std (Hm,N,S).
 * Saved  : a_hidden (n_actor_layers-1, a_layer_rows, Hd) with a_layer_rows >= Hm*N (row stride of a
 *          layer, so the caller can keep one spare step slot behind the rollout's rows), a_raw
 *          (Hm*N,2A), a_mean/a_std (Hm*N,A), xsa (Hm*N,S+A+C), e (Hm*N,D), gates (Hm*N,4D), hp (Hm*N,Hd).
 * cond (N,C), C > 0: the CONDITIONED rollout of the multitask agents (ConditionalTransitionModel.imagine,
 *          models/rssm.py:221-249; ConditionalActorModel, models/actor_critic.py:104-139): the policy sees
 *          [belief | state | cond], the belief update the pseudo-action [action | cond]; then actor_params[0] is
 *          (Hd, D+S+C), rssm_params[0] is (D, S+A+C), and the saved xsa rows are [state | action | cond].  The
 *          condition rides in the K padding of the persistent engine's tiles: D+S+C <= 240 and S+A+C <= 48 at the
 *          reference's widths, else REPO_E_SHAPE.  C = 0 (cond ignored): the reference's unconditioned rollout. */
size_t repo_rssm_imagine_fwd_workspace_bytes(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd,
                                             int64_t S);
int repo_rssm_imagine_fwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S,
                          int n_actor_layers, const float* const* rssm_params,
                          const float* const* actor_params, const float* belief0, const float* state0,
                          const float* cond, int64_t C,
                          const float* eps_act, const float* eps_prior, uint64_t noise_seed,
                          uint64_t noise_offset, float min_std, float a_min_std,
                          float a_init_std, float a_mean_scale, float* featx, float* prior_mean,
                          float* prior_std, float* a_hidden, int64_t a_layer_rows, float* a_raw,
                          float* a_mean, float* a_std, float* xsa, float* e, float* gates, float* hp,
                          void* ws, size_t ws_bytes, hipStream_t stream);
/* Reverse pass with frozen world-model weights: dfeat (Hm,N,D+S) is the gradient w.r.t.
 * featx[1:] (from the heads and the entropy term), dprior_mean/std nullable.  Emits
 * d_araw (Hm*N,2A), the gradient at the actor trunk's output of every step (the caller
 * finishes with repo_mlp_bwd over all Hm*N rows), and optionally dfeat0 (N,D+S). */
size_t repo_rssm_imagine_bwd_workspace_bytes(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd,
                                             int64_t S);
int repo_rssm_imagine_bwd(int64_t Hm, int64_t N, int64_t A, int64_t D, int64_t Hd, int64_t S, int64_t C,
                          const float* const* rssm_params, const float* eps_act,
                          const float* eps_prior, uint64_t noise_seed, uint64_t noise_offset,
                          float min_std, float a_min_std, float a_mean_scale,
                          const float* featx, const float* prior_std, const float* a_mean,
                          const float* a_std, const float* xsa, const float* e, const float* gates,
                          const float* hp, const float* dfeat, const float* dprior_mean,
                          const float* dprior_std, float* d_araw, float* dfeat0, void* ws,
                          size_t ws_bytes, hipStream_t stream);

/* ------------------------------------------------------------------ losses and regularisers
 * All reductions use repo_reduce_workspace_bytes() of scratch and write device scalars
 * (no host synchronisation; the caller batches its .item() reads).
 * ABI v8 -- a reduction whose grid has at most 64 blocks (the KL up to 2560 rows, the scalar heads' NLL up to 65536
 * elements, the lambda-returns up to 16384 rows) is ONE launch: its last block sums the per-block partials in the fixed
 * order of the follow-up launch, which larger grids still get (tickets are same-address atomics: 1000 of them cost more
 * than the launch they would save -- measured).  The scratch of these calls (and of repo_tia_blend_nll,
 * repo_grad_sqnorm) therefore begins with a 256-byte HEADER whose FIRST WORD must be zero on entry and is left zero on
 * return (the blocks' ticket; repo_amd keeps repo_film_bwd_h's epoch word at byte 32 of the same header): allocate the buffer zeroed, keep it for these calls only, one per stream (two reductions in
 * flight on two streams must not share it).  repo_amd.ops.reduce_ws does exactly that. */
size_t repo_reduce_workspace_bytes(void);

/* KL(q||p) of diagonal Gaussians over rows x S.
 * mode 0 (RePo, algorithms/repo/repo.py:64-83): *kl_sum = sum_rows KL; gradients of
 *   exp(*log_beta) * (alpha*KL(sg q||p) + (1-alpha)*KL(q||sg p)) * scale.
 * mode 1 (Dreamer, algorithms/repo/dreamer.py:278-282): *kl_sum = sum_rows max(KL, free_nats);
 *   gradients of that * scale.   Gradient outputs are nullable. */
int repo_kl_balance(int64_t rows, int64_t S, const float* pm, const float* ps, const float* qm,
                    const float* qs, int mode, float alpha, const float* log_beta, float free_nats,
                    float scale, float* dpm, float* dps, float* dqm, float* dqs, float* kl_sum,
                    void* ws, size_t ws_bytes, hipStream_t stream);
/* Lagrangian dual ascent on log_beta (repo.py:83,93-105): grad = -(kl_sum/rows - target_kl),
 * one Adam step (state exp_avg/exp_avg_sq on device, `step` = 1-based count) if apply.
 * scalars_out[4] = {kl_div, kl_loss = beta_old*viol, beta_loss = -log_beta_old*viol, beta_new}.
 * skip_if_nonzero: see repo_clip_adam -- a faulted update leaves log_beta and its moments untouched (the scalars are
 * still written: they are what the caller logs, NaN included). */
int repo_dual_step(float* log_beta, float* exp_avg, float* exp_avg_sq, const float* kl_sum,
                   int64_t rows, float target_kl, float lr, float beta1, float beta2, float eps,
                   int64_t step, int apply, float* scalars_out, const unsigned* skip_if_nonzero,
                   hipStream_t stream);
/* Unit-variance Gaussian NLL of a scalar head (reward repo.py:58-61, value dreamer.py:365-368):
 * sums2[0] = sum 0.5*(pred-target)^2*mask, sums2[1] = sum mask (mask NULL = ones);
 * dpred = (pred-target)*mask*scale (nullable). */
int repo_scalar_nll(int64_t n, const float* pred, const float* target, const float* mask, float scale,
                    float* dpred, float* sums2, void* ws, size_t ws_bytes, hipStream_t stream);
/* TIA's masked blend of two decoders + unit-variance pixel NLL (tia.py:123-133; TIAObservationModel,
 * models/decoder.py:154-175): t_out, d_out (nimg, 6, pixels) = [recon(3) | mask(3)] of the task / distractor
 * decoder; mask_wb[7] = mask_head's Conv2d(6,1,1) weight (t_mask(3), d_mask(3)) and bias;
 *   m = sigmoid(bias + w . [t_mask | d_mask]);  recon = t_recon*m + d_recon*(1-m)
 * sums8 = {sum 0.5*(recon-target)^2, d/dw[0..5], d/dbias} of grad_scale * that sum (the loss itself unscaled);
 * dt_out / dd_out (nullable together; may alias t_out / d_out) = grad_scale * d sum / d t_out, d_out;
 * recon (nimg, 3, pixels) nullable.  target (nimg, 3, pixels) float32 in [-1,1] or uint8 (normalised in-kernel).
 * pixels % 4 == 0. */
size_t repo_tia_blend_nll_workspace_bytes(void);
int repo_tia_blend_nll(int64_t nimg, int64_t pixels, const float* t_out, const float* d_out,
                       const float* mask_wb, const void* target, int target_is_u8, float grad_scale,
                       float* dt_out, float* dd_out, float* recon, float* sums8, void* ws, size_t ws_bytes,
                       hipStream_t stream);
/* SampleDist.entropy of the tanh-Normal policy (models/utils.py:126-134,160-163):
 * eps (samples, rows, A), or NULL => drawn in-kernel, sample s of element e = row*A + a being normal number
 * noise_offset + e*samples + s (sample-fastest; rows*A*samples consumed).  *ent_sum = sum_rows entropy_row;
 * dmean/dstd (rows,A) = gscale * d ent_sum / d(mean,std) (nullable). */
int repo_tanh_normal_entropy(int64_t rows, int64_t A, int64_t samples, const float* mean,
                             const float* std, const float* eps, uint64_t noise_seed,
                             uint64_t noise_offset, float gscale, float* dmean,
                             float* dstd, float* ent_sum, void* ws, size_t ws_bytes, hipStream_t stream);
/* SampleDist.mode (models/utils.py:149-158): per row, the tanh-Normal sample with the highest
 * log-probability among `samples` draws (first maximum, like torch.argmax); eps (samples,rows,A). */
int repo_tanh_normal_mode(int64_t rows, int64_t A, int64_t samples, const float* mean, const float* std,
                          const float* eps, float* action, hipStream_t stream);
/* Independent(Normal).entropy summed over n elements (dreamer.py:327-328): dstd = gscale/std. */
int repo_normal_entropy(int64_t n, const float* std, float gscale, float* dstd, float* ent_sum,
                        void* ws, size_t ws_bytes, hipStream_t stream);
/* lambda_return (common/utils.py:61-71) as called at dreamer.py:342-349: rewards, values
 * (Hm,N); returns (Hm-1,N); *ret_sum = sum(returns); drewards/dvalues (Hm,N) = gradient of
 * gret*sum(returns) (both or neither). */
int repo_lambda_return(int64_t Hm, int64_t N, const float* rewards, const float* values, float gamma,
                       float lambda_, float gret, float* returns, float* drewards, float* dvalues,
                       float* ret_sum, void* ws, size_t ws_bytes, hipStream_t stream);

/* ------------------------------------------------------------------ multitask (task-conditioned) agents
 * MultitaskDreamer / MultitaskRePo (algorithms/repo/dreamer_mt.py, repo_mt.py) condition every module on the task
 * one-hot: dense layers by concatenation -- K columns of the caller's rows, no kernel of their own (heads:
 * repo_mlp_fwd with in_dim = D+S+C; observe scan: pseudo-actions [action | task] = repo_rssm_observe_fwd with
 * A + C "actions"; rollout: repo_rssm_imagine_fwd's `cond`) -- and the conv stacks by FiLM:
 *   h[n][c][p] = relu((1 + gamma[n][c]) * y[n][c][p] + beta[n][c])
 * (ConditionalVisualEncoder.mod, models/encoder.py:75-88; ConditionalVisualObservationModel.mod, models/decoder.py:
 * 108-123), y = the layer's conv output incl. bias (repo_conv_down / repo_conv_up / repo_gemm with REPO_EPI_NONE),
 * gamma = film[n*ldfilm + gamma_off + c], beta = film[n*ldfilm + beta_off + c] with film (nimg, ldfilm) = the FiLM
 * linear layer's output (film(condition).chunk(2).split(...) of the reference, by offsets).  P = pixels per plane.
 * Backward: dh = the gradient at the ReLU's input (the ReLU mask already applied by the producer: a data-gradient
 * kernel with REPO_EPI_MUL_DRELU on h, or repo_relu_mask);  dy = dh * (1 + gamma);
 * dfilm[n][gamma_off + c] = sum_p dh * y,  dfilm[n][beta_off + c] = sum_p dh  (written, fixed summation order). */
int repo_film_fwd(int64_t nimg, int64_t C, int64_t P, const float* y, const float* film, int64_t ldfilm,
                  int64_t gamma_off, int64_t beta_off, float* out, hipStream_t stream);
int repo_film_bwd(int64_t nimg, int64_t C, int64_t P, const float* dh, const float* y, const float* film,
                  int64_t ldfilm, int64_t gamma_off, int64_t beta_off, float* dy, float* dfilm, hipStream_t stream);
/* The FiLM tables of the layers of one conv stack in ONE launch: film (nimg, ldfilm) is the FiLM Linear's output,
 * [gammas of all layers | betas of all layers] (film(condition).chunk(2) then .split(channels), encoder.py:80-82); layer l
 * with channels[l] channels at column offset sum(channels[:l]) gets tables + nimg * 2 * sum(channels[:l]) =
 * (nimg, 2, channels[l]) = [1 + gamma | beta] per image (REPO_EPI_FILM_RELU's aux).  nlayers <= 4; `channels` is a HOST array. */
int repo_film_tables(int64_t nimg, int nlayers, const int* channels, const float* film, int64_t ldfilm, float* tables,
                     hipStream_t stream);
/* repo_film_bwd when the layer ran with REPO_EPI_FILM_RELU and only its OUTPUT h = relu((1 + gamma) y + beta) exists:
 * y = (h - beta) / (1 + gamma) wherever dh != 0 (there h > 0) -- for planes with |1 + gamma| >= 1e-3 (the recovered y
 * then carries a relative error of about eps |beta| / |(1 + gamma) y|: at most ~6e-5 for |beta| ~ |y|).  A plane below
 * that -- a channel the FiLM layer has gated off for that image's task, where the recovery loses y -- RECOMPUTES its y
 * exactly from the layer's own input, weights and bias (ABI v8), described by
 *   conv_kind 1: the stride-2 convolution of repo_conv_down, geo = {CB, CS, HB, KS}, x = its `big` input (uint8 frames if
 *                x_is_u8, normalised like repo_conv_down does), w (CS, CB, KS, KS), bias (CS); planes = small channels;
 *   conv_kind 2: its transpose (repo_conv_up), x = the `small` input, same w, bias (CB); planes = big channels;
 *   conv_kind 3: dense (the decoder's 1 x 1 -> 5 x 5 first layer), geo = {K, per_element_bias}: y[n][c*P + p] = bias[c] (or
 *                bias[c*P + p] if per_element_bias) + sum_k x[n][k] w[k][c*P + p];
 *   gated_epoch / epoch (nullable / non-zero): one device word, zero-initialised once, and a number that grows from call
 *                to call -- the streaming pass stamps the word with this call's epoch when it meets a gated-off plane,
 *                and the exact pass (a second launch) returns at once unless the word carries it: a layer without
 *                gated-off planes pays a few microseconds.  NULL: the exact pass always walks every plane's scale.
 *   conv_kind 0: no description (geo / x / w / bias unused): every plane is recovered, and one whose 1 + gamma is
 *                exactly 0 contributes no gamma gradient -- the round-5 behaviour, kept for callers without the inputs.
 * The reference differentiates the saved conv output itself (models/encoder.py:84-87, models/decoder.py:117-122). */
int repo_film_bwd_h(int64_t nimg, int64_t C, int64_t P, const float* dh, const float* h, const float* film,
                    int64_t ldfilm, int64_t gamma_off, int64_t beta_off, float* dy, float* dfilm, int conv_kind,
                    const int64_t* geo, const void* x, int x_is_u8, const float* w, const float* bias,
                    unsigned* gated_epoch, unsigned epoch, hipStream_t stream);
/* MultitaskRePo's KL balance (repo_mt.py:75-93): the Lagrange multiplier is PER ROW, beta_row = exp(lb_row) with
 * lb_row = tasks[row] . log_beta (tasks (rows, C) one-hot, log_beta (C), C <= 13).  Gradients (nullable) of
 *   scale * sum_rows beta_row * (alpha*KL(sg q||p) + (1-alpha)*KL(q||sg p)),
 * sums[3 + C] = { sum KL_row, sum beta_row*viol_row, sum lb_row*viol_row, sum tasks[row][i]*viol_row (i < C) } with
 * viol_row = KL_row - target_kl: this rank's partial sums (a data-parallel job all-reduces them) for
 * repo_dual_step_tasks. */
size_t repo_kl_balance_tasks_workspace_bytes(void);
int repo_kl_balance_tasks(int64_t rows, int64_t S, int64_t C, const float* pm, const float* ps, const float* qm,
                          const float* qs, float alpha, const float* log_beta, const float* tasks, float target_kl,
                          float scale, float* dpm, float* dps, float* dqm, float* dqs, float* sums, void* ws,
                          size_t ws_bytes, hipStream_t stream);
/* Dual ascent on the per-task log_beta vector (repo_mt.py:95-112): beta_loss = -mean_rows(lb_row * viol_row), so
 * grad[i] = -sums[3+i] / rows; ONE Adam step on the C-vector if apply (`step` = 1-based count, state on device).
 * rows = the GLOBAL row count.  scalars_out[3 + C] = { kl_div, kl_loss, beta_loss, exp(log_beta[i]) after the step }.
 * skip_if_nonzero: as repo_dual_step. */
int repo_dual_step_tasks(int64_t C, float* log_beta, float* exp_avg, float* exp_avg_sq, const float* sums,
                         int64_t rows, float lr, float beta1, float beta2, float eps, int64_t step, int apply,
                         float* scalars_out, const unsigned* skip_if_nonzero, hipStream_t stream);

/* dst[c * ldd + r] = src[r * lds + c] for r < rows, c < cols; columns [rows, ldd) of dst are written as zeros
 * (ldd - rows < 64; lds, ldd multiples of 4, 16-byte aligned pointers).  The host side uses it to hand the bf16x6 dense
 * engine (repo_gemm) k-contiguous operands for the decoder's 1024 -> 3200 layer: forward on W^T (the reference's
 * ConvTranspose2d weight keeps its (in, out, kH, kW) layout: models/decoder.py:43), weight gradient on the transposed
 * activations -- torch's counterpart is the .t() view that at::mm resolves inside the BLAS call. */
int repo_transpose(int64_t rows, int64_t cols, const float* src, int64_t lds, float* dst, int64_t ldd,
                   hipStream_t stream);
/* 1 if a product of this size (C (M x N) over K) is one repo_gemm runs on the bf16x6 engine -- the engine whose NT form
 * (both operands k-contiguous) is worth a transposing copy of an operand (ABI v8) -- under the calling thread's
 * repo_debug_bgemm setting, else 0.  Sizes only: the caller still checks what the engine asks of the operands themselves
 * (leading dimensions % 4 == 0, 16-byte aligned pointers) and leaves them as they are otherwise -- repo_gemm then takes
 * the fp32-MFMA tile engines on the untransposed operands, as it did before the NT forms existed. */
int repo_gemm_nt_pays(int64_t M, int64_t N, int64_t K);

/* ------------------------------------------------------------------ optimiser
 * *sqnorm = sum g^2 over a flat, 16-byte aligned buffer (global norm of
 * nn.utils.clip_grad_norm_, repo.py:89).  Fixed-order two-level sum (two launches); ws: a reduction workspace (its
 * header is left untouched: see "losses and regularisers"). */
size_t repo_grad_sqnorm_workspace_bytes(void);
int repo_grad_sqnorm(int64_t n, const float* g, float* sqnorm, void* ws, size_t ws_bytes,
                     hipStream_t stream);
/* g *= min(1, max_norm/(sqrt(*sqnorm)+1e-6)) fused with torch.optim.Adam's update
 * (betas, eps, no weight decay; `step` = 1-based count).  sqnorm NULL = no clipping.
 * skip_if_nonzero (ABI v6, nullable): a device word -- the update's copy of the scans' asynchronous `status`
 * (repo_rssm_observe_fwd), reduced over the ranks of a data-parallel job.  If it is non-zero when the kernel runs,
 * NOTHING is written: parameters and both moments keep their values, so that an update whose scan timed out (its
 * gradients are NaN-poisoned by then) costs the caller one retry, not the model.  The reference has no counterpart:
 * its optimiser steps cannot be reached by a failed kernel (an exception unwinds repo.py:86-90 first). */
int repo_clip_adam(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   const float* sqnorm, float max_norm, float lr, float beta1, float beta2, float eps,
                   int64_t step, const unsigned* skip_if_nonzero, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REPO_HIP_H */
