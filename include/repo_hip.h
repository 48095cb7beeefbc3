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

#define REPO_ABI_VERSION 1

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

int repo_abi_version(void);
const char* repo_strerror(int code);

/* ------------------------------------------------------------------ dense layers
 * C[m][n] (+)= epi( sum_k opA(m,k) * opB(k,n) + bias[n / bias_div] )
 *   opA(m,k) = transa ? A[k*lda + m] : A[m*lda + k]
 *   opB(k,n) = transb ? B[n*ldb + k] : B[k*ldb + n]
 * Replaces every nn.Linear forward / autograd backward-data on the path, e.g.
 * F.linear in models/rssm.py:36-64, models/decoder.py:42,191-194,
 * models/actor_critic.py:21-25,77-82 (transa=0, transb=1), their input gradients
 * (transb=0), and the 1x1 -> 5x5 first transposed convolution of the decoder
 * (models/decoder.py:44), which is a plain GEMM against the (1024, 128*25) weight.
 * bias may be NULL; aux (ld = ldaux) is read only by the MUL_* epilogues. */
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
 * Every geometry is a pair (big, small) with big = 2*small + k - 2 and one weight tensor
 * indexed w[small_ch][big_ch][ky][kx] -- which is both nn.Conv2d's (out,in,kh,kw) for the
 * encoder and nn.ConvTranspose2d's (in,out,kh,kw) for the decoder.  Three kernels act on a
 * pair:
 *   down : small = gather(big)      encoder forward / decoder backward-data
 *   up   : big   = scatter(small)   decoder forward / encoder backward-data
 *   wgrad: dw = corr(small, big)    both
 * epi: REPO_EPI_NONE (+bias if given), REPO_EPI_RELU (+bias), REPO_EPI_MUL_DRELU (aux has
 * the output's shape).  `big_is_u8` != 0: big is uint8 pixels (encoder conv1 only). */
int repo_conv_down(int layer, int64_t nimg, const void* big, int big_is_u8, const float* w,
                   const float* bias, float* small, int epi, const float* aux, hipStream_t stream);
int repo_conv_up(int layer, int64_t nimg, const float* small, const float* w, const float* bias,
                 float* big, int epi, const float* aux, hipStream_t stream);
/* dw (+)= ..., dbias_small[small_ch] (+)= sum over images and pixels of `small` (NULL to skip). */
size_t repo_conv_wgrad_workspace_bytes(int layer, int64_t nimg);
int repo_conv_wgrad(int layer, int64_t nimg, const float* small, const void* big, int big_is_u8,
                    float* dw, float* dbias_small, int accumulate, void* ws, size_t ws_bytes,
                    hipStream_t stream);

/* Final decoder layer fused with the pixel likelihood (models/decoder.py:47 +
 * -Normal(recon,1).log_prob(obs[1:]).sum((2,3,4)).mean((0,1)), repo.py:46-53):
 *   recon = up(layer 6)(h3) + bias;  d = recon - target
 *   dpre[img][c][y][x] = d * grad_scale          (gradient w.r.t. recon of the mean loss)
 *   loss_sum += 0.5*d*d   (per-workgroup partials in ws, reduced into *loss_sum in order)
 * recon and dpre may each be NULL.  target is uint8 (target_is_u8) or fp32 in [-1,1]. */
size_t repo_decoder_out_nll_workspace_bytes(int64_t nimg);
int repo_decoder_out_nll(int64_t nimg, const float* h3, const float* w, const float* bias,
                         const void* target, int target_is_u8, float grad_scale, float* recon,
                         float* dpre, float* loss_sum, void* ws, size_t ws_bytes, hipStream_t stream);

/* out[c] (+)= sum over n and p of x[n][c][p]   (bias gradient of an NCHW activation) */
size_t repo_channel_sum_workspace_bytes(int64_t nimg, int64_t C, int64_t P);
int repo_channel_sum(int64_t nimg, int64_t C, int64_t P, const float* x, float* out, int accumulate,
                     void* ws, size_t ws_bytes, hipStream_t stream);

/* y = dy * (h > 0)  (ReLU backward through a saved output), n elements */
int repo_relu_mask(int64_t n, const float* dy, const float* h, float* y, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REPO_HIP_H */
