"""Tensor-level wrappers over the C ABI (include/repo_hip.h).

Each function takes torch CUDA tensors, passes raw device pointers, leading dimensions
and the current HIP stream to librepo_hip.so, and returns the output tensor.  PyTorch is
used for device memory and streams only -- there is no eager/CPU fallback here: a missing
library or a CPU tensor raises.
"""
import torch

from ._lib import check, lib

EPI_NONE, EPI_ELU, EPI_RELU, EPI_MUL_DELU, EPI_MUL_DRELU = 0, 1, 2, 3, 4

# layer ids of repo_conv_* (include/repo_hip.h)
ENC1, ENC2, ENC3, ENC4, DEC2, DEC3, DEC4 = range(7)
# (CB, CS, HB, KS) per layer; HS = (HB-KS)//2+1
CONV_GEO = {
    ENC1: (3, 32, 64, 4),
    ENC2: (32, 64, 31, 4),
    ENC3: (64, 128, 14, 4),
    ENC4: (128, 256, 6, 4),
    DEC2: (64, 128, 13, 5),
    DEC3: (32, 64, 30, 6),
    DEC4: (3, 32, 64, 6),
}


def conv_shapes(layer):
    cb, cs, hb, ks = CONV_GEO[layer]
    hs = (hb - ks) // 2 + 1
    return (cb, hb, hb), (cs, hs, hs)


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("repo_amd ops need CUDA (HIP) tensors; there is no CPU fallback")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_ws = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (stream-ordered reuse on the current stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


def _ld(t):
    """Leading dimension of a 2-D view whose rows are contiguous."""
    assert t.dim() == 2 and t.dtype == torch.float32 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def gemm(A, B, transa=False, transb=False, bias=None, bias_div=1, out=None, epi=EPI_NONE, aux=None, accumulate=False):
    """C = epi(opA @ opB + bias).  A is (M,K) [or (K,M) if transa], B is (K,N) [or (N,K) if transb]."""
    M, K = (A.shape[1], A.shape[0]) if transa else (A.shape[0], A.shape[1])
    N = B.shape[0] if transb else B.shape[1]
    assert (B.shape[1] if transb else B.shape[0]) == K, (A.shape, B.shape, transa, transb)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    check(
        lib().repo_gemm(
            int(transa), int(transb), M, N, K, _ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(bias), bias_div,
            _ptr(out), _ld(out), epi, _ptr(aux), _ld(aux) if aux is not None else 0, int(accumulate), _stream(),
        ),
        "repo_gemm",
    )
    return out


def linear(x, w, b=None, epi=EPI_NONE, out=None):
    """F.linear(x, w, b) with a fused activation."""
    return gemm(x, w, transb=True, bias=b, epi=epi, out=out)


def gemm_wgrad(dY, X, dW=None, db=None, accumulate=False, want_bias=True):
    """dW[n][k] = sum_m dY[m][n] X[m][k]; db[n] = sum_m dY[m][n]."""
    M, N = dY.shape
    K = X.shape[1]
    assert X.shape[0] == M
    if dW is None:
        dW = torch.empty(N, K, dtype=torch.float32, device=dY.device)
    if db is None and want_bias:
        db = torch.empty(N, dtype=torch.float32, device=dY.device)
    nb = lib().repo_gemm_wgrad_workspace_bytes(M, N, K)
    ws = workspace(nb, dY.device)
    check(
        lib().repo_gemm_wgrad(
            M, N, K, _ptr(dY), _ld(dY), _ptr(X), _ld(X), _ptr(dW), _ld(dW), _ptr(db), int(accumulate),
            _ptr(ws), ws.numel(), _stream(),
        ),
        "repo_gemm_wgrad",
    )
    return dW, db


def conv_down(layer, big, w, bias=None, epi=EPI_NONE, aux=None, out=None):
    nimg = big.shape[0]
    (cb, hb, _), (cs, hs, _) = conv_shapes(layer)
    assert tuple(big.shape[1:]) == (cb, hb, hb) and big.is_contiguous(), big.shape
    is_u8 = big.dtype == torch.uint8
    assert is_u8 or big.dtype == torch.float32
    if out is None:
        out = torch.empty(nimg, cs, hs, hs, dtype=torch.float32, device=big.device)
    check(
        lib().repo_conv_down(layer, nimg, _ptr(big), int(is_u8), _ptr(_f32c(w)), _ptr(bias), _ptr(out), epi,
                             _ptr(aux), _stream()),
        "repo_conv_down",
    )
    return out


def conv_up(layer, small, w, bias=None, epi=EPI_NONE, aux=None, out=None):
    nimg = small.shape[0]
    (cb, hb, _), (cs, hs, _) = conv_shapes(layer)
    assert tuple(small.shape[1:]) == (cs, hs, hs) and small.is_contiguous(), small.shape
    if out is None:
        out = torch.empty(nimg, cb, hb, hb, dtype=torch.float32, device=small.device)
    check(
        lib().repo_conv_up(layer, nimg, _ptr(_f32c(small)), _ptr(_f32c(w)), _ptr(bias), _ptr(out), epi, _ptr(aux),
                           _stream()),
        "repo_conv_up",
    )
    return out


def conv_wgrad(layer, small, big, dw=None, db=None, accumulate=False, want_bias=True):
    nimg = small.shape[0]
    (cb, hb, _), (cs, hs, _) = conv_shapes(layer)
    ks = CONV_GEO[layer][3]
    assert big.is_contiguous() and small.is_contiguous()
    is_u8 = big.dtype == torch.uint8
    if dw is None:
        dw = torch.empty(cs, cb, ks, ks, dtype=torch.float32, device=small.device)
    if db is None and want_bias:
        db = torch.empty(cs, dtype=torch.float32, device=small.device)
    nb = lib().repo_conv_wgrad_workspace_bytes(layer, nimg)
    ws = workspace(nb, small.device)
    check(
        lib().repo_conv_wgrad(layer, nimg, _ptr(_f32c(small)), _ptr(big), int(is_u8), _ptr(dw), _ptr(db),
                              int(accumulate), _ptr(ws), ws.numel(), _stream()),
        "repo_conv_wgrad",
    )
    return dw, db


def decoder_out_nll(h3, w, bias, target, grad_scale, want_recon=False, want_dpre=True):
    """Final transposed conv fused with 0.5*(recon-target)^2 summed over everything.
    Returns (loss_sum (1,), dpre or None, recon or None)."""
    nimg = h3.shape[0]
    is_u8 = target.dtype == torch.uint8
    assert target.is_contiguous() and target.numel() == nimg * 3 * 64 * 64
    dev = h3.device
    recon = torch.empty(nimg, 3, 64, 64, dtype=torch.float32, device=dev) if want_recon else None
    dpre = torch.empty(nimg, 3, 64, 64, dtype=torch.float32, device=dev) if want_dpre else None
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    nb = lib().repo_decoder_out_nll_workspace_bytes(nimg)
    ws = workspace(nb, dev)
    check(
        lib().repo_decoder_out_nll(nimg, _ptr(_f32c(h3)), _ptr(_f32c(w)), _ptr(bias), _ptr(target), int(is_u8),
                                   float(grad_scale), _ptr(recon), _ptr(dpre), _ptr(loss), _ptr(ws), ws.numel(),
                                   _stream()),
        "repo_decoder_out_nll",
    )
    return loss, dpre, recon


def channel_sum(x, out=None, accumulate=False):
    nimg, C = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = lib().repo_channel_sum_workspace_bytes(nimg, C, P)
    ws = workspace(nb, x.device)
    check(
        lib().repo_channel_sum(nimg, C, P, _ptr(_f32c(x)), _ptr(out), int(accumulate), _ptr(ws), ws.numel(), _stream()),
        "repo_channel_sum",
    )
    return out


def relu_mask(dy, h, out=None):
    if out is None:
        out = torch.empty_like(dy)
    check(lib().repo_relu_mask(dy.numel(), _ptr(_f32c(dy)), _ptr(_f32c(h)), _ptr(out), _stream()), "repo_relu_mask")
    return out
