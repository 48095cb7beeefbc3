"""Tensor-level wrappers over the C ABI (include/repo_hip.h).

Each function takes torch CUDA tensors, passes raw device pointers, leading dimensions
and the current HIP stream to librepo_hip.so, and returns the output tensor.  PyTorch is
used for device memory and streams only -- there is no eager/CPU fallback here: a missing
library or a CPU tensor raises.
"""
import contextlib
import os
import threading

import torch

from ._lib import check, lib

EPI_NONE, EPI_ELU, EPI_RELU, EPI_MUL_DELU, EPI_MUL_DRELU, EPI_MUL_MASK4, EPI_MUL_CMASK, EPI_FILM_RELU = 0, 1, 2, 3, 4, 5, 6, 7

# layer ids of repo_conv_* (include/repo_hip.h): 0..6 the reference's 64 x 64 stack, 7..12 the build-defined 128 x 128 one
ENC1, ENC2, ENC3, ENC4, DEC2, DEC3, DEC4 = range(7)
X_ENC1, X_ENC2, X_ENC3, X_ENC4, X_DEC4, X_DEC5 = range(7, 13)
T_DEC4 = 13  # TIAObservationModel.conv4 (32 -> 6 = [recon | mask])
# (CB, CS, HB, KS) per layer; HS = (HB-KS)//2+1
CONV_GEO = {
    ENC1: (3, 32, 64, 4),
    ENC2: (32, 64, 31, 4),
    ENC3: (64, 128, 14, 4),
    ENC4: (128, 256, 6, 4),
    DEC2: (64, 128, 13, 5),
    DEC3: (32, 64, 30, 6),
    DEC4: (3, 32, 64, 6),
    X_ENC1: (3, 32, 128, 4),
    X_ENC2: (32, 64, 63, 4),
    X_ENC3: (64, 128, 30, 4),
    X_ENC4: (128, 256, 14, 4),
    X_DEC4: (16, 32, 64, 6),
    X_DEC5: (3, 16, 128, 2),
    T_DEC4: (6, 32, 64, 6),
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


_raw_stream = torch._C._cuda_getCurrentRawStream  # the handle only: torch.cuda.current_stream() builds a Stream
                                                   # object per call (7 us x 120 calls per update on the host)


def _stream():
    return _raw_stream(torch.cuda.current_device())


_ws = {}
_scope = threading.local()   # .pool: the scratch dict of the innermost scratch_scope() of this thread, if any


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, current stream): reuse is stream-ordered, and work
    running concurrently on a side stream never shares scratch with the main stream.

    The buffers live in a process-wide dict -- except inside `scratch_scope()`, whose dict the caller owns.  A HIP
    graph bakes the POINTERS of the scratch its kernels were captured with, so a captured region must never see the
    process-wide dict: a later, larger request on a stream with the same raw handle (PyTorch hands out 32 handles
    round-robin) replaces that entry, the old block returns to the allocator, and every replay of the graph then
    scribbles (the scans start with a 0xFF fill = NaN) over whatever tensor has been given the block since.  Use
    `capture_graph()`, which owns its scratch for the graph's lifetime."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _raw_stream(idx))
    pool = getattr(_scope, "pool", None)
    if pool is None:
        pool = _ws
    buf = pool.get(key)
    if buf is None or buf.numel() < nbytes:
        if pool is _ws and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("repo_amd.ops: scratch requested under stream capture outside ops.scratch_scope(); "
                               "capture through ops.capture_graph() so that the graph owns its scratch")
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        pool[key] = buf
    return buf


@contextlib.contextmanager
def scratch_scope(pool=None):
    """Every workspace() request of this thread inside the block is served from `pool` (a dict the caller keeps alive
    for as long as anything launched inside may still run -- a graph: for as long as it can be replayed)."""
    pool = {} if pool is None else pool
    prev = getattr(_scope, "pool", None)
    _scope.pool = pool
    try:
        yield pool
    finally:
        _scope.pool = prev


def capture_graph(fn, device=None, warmup=2):
    """Capture `fn()` (a no-grad launch sequence over STATIC input tensors that returns a tuple of output tensors) into
    a HIP graph.  Returns (graph, outputs, keep): replay with graph.replay(); `keep` owns every scratch buffer the
    captured kernels hold pointers to and must live as long as the graph.  Warm-up (lazy state, scratch sizing) runs
    on a side stream, as torch asks for before a capture; warm-up and capture see ONLY the graph's own scratch."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    keep = {}
    with scratch_scope(keep), torch.no_grad():
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = tuple(fn())
    return graph, outs, keep


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


def _ld(t):
    """Leading dimension of a 2-D view whose rows are contiguous."""
    assert t.dim() == 2 and t.dtype == torch.float32 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _nt_forms():
    return os.environ.get("REPO_GEMM_NT", "1") == "1"


def gemm(A, B, transa=False, transb=False, bias=None, bias_div=1, out=None, epi=EPI_NONE, aux=None, accumulate=False):
    """C = epi(opA @ opB + bias).  A is (M,K) [or (K,M) if transa], B is (K,N) [or (N,K) if transb]."""
    M, K = (A.shape[1], A.shape[0]) if transa else (A.shape[0], A.shape[1])
    N = B.shape[0] if transb else B.shape[1]
    assert (B.shape[1] if transb else B.shape[0]) == K, (A.shape, B.shape, transa, transb)
    if not transa and not transb and _nt_pays(M, N, K, A, B):
        B, transb = transpose(B), True     # (K, N) -> (N, K): both operands k-contiguous
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


def transpose(src, out=None):
    """(R, C) row-major -> (C, R) with the row pitch padded to a multiple of 4 (pad columns zero); returns the (C, R) view."""
    R, C = src.shape
    ld = (R + 3) // 4 * 4
    if out is None:
        out = torch.empty(C, ld, dtype=torch.float32, device=src.device)
    check(lib().repo_transpose(R, C, _ptr(src), _ld(src), _ptr(out), out.stride(0), _stream()), "repo_transpose")
    return out[:, :R]


# Big products on the bf16x6 engine run fastest with BOTH operands k-contiguous (csrc/bgemm.h: the "NT" form; an
# m/n-contiguous operand is transposed element-wise while it is staged): 2450 x 3200 x 1024 forward 153 -> 117 us, the
# 1024 x 3200 weight gradient over 2450 rows 184 -> 121 us.  Above this size a transposing copy (one streaming pass,
# repo_transpose) is cheaper than the slower form.
_NT_MIN = 512


def _aligned(t):
    return t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def _nt_pays(M, N, K, *operands):
    """The transposing copy is made only for a product the bf16x6 engine will take in its NT form (the same predicate as
    the library's dispatch: repo_gemm_nt_pays -- sizes, enough tiles to fill the chip, this thread's repo_debug_bgemm)
    and only from operands repo_transpose and the engine accept; anything else -- N = 513, a misaligned view, a
    data-parallel shard whose 637 rows make 125 tiles -- stays on the untransposed fp32-MFMA forms."""
    return (min(M, N, K) >= _NT_MIN and _nt_forms() and all(_aligned(t) for t in operands)
            and bool(lib().repo_gemm_nt_pays(M, N, K)))


# ----------------------------------------------------------------------------- test-aid switches (thread-local in the library)
_DEBUG_SWITCHES = ("repo_debug_scan_spin_limit", "repo_debug_bgemm", "repo_debug_bconv", "repo_debug_rowtile32")


def debug_snapshot():
    """The calling thread's four repo_debug_* settings (each setter returns the previous value: set, then restore)."""
    out = []
    for name in _DEBUG_SWITCHES:
        fn = getattr(lib(), name)
        prev = fn(1)
        fn(prev)
        out.append(prev)
    return tuple(out)


@contextlib.contextmanager
def debug_scope(state):
    """Apply a debug_snapshot() on THIS thread for the block (the autograd wrappers' backward runs on torch's device
    thread, which would otherwise see the library's defaults)."""
    prev = [getattr(lib(), name)(v) for name, v in zip(_DEBUG_SWITCHES, state)]
    try:
        yield
    finally:
        for name, v in zip(_DEBUG_SWITCHES, prev):
            getattr(lib(), name)(v)


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
    if db is None and not want_bias and _nt_pays(N, K, M, dY, X):
        # dW = dY^T X as ONE product over k = rows with both operands k-contiguous
        return gemm(transpose(dY), transpose(X), transb=True, out=dW, accumulate=accumulate), None
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


def conv_down(layer, big, w, bias=None, epi=EPI_NONE, aux=None, out=None, dbias=None, accumulate_dbias=False,
              want_cmask=False):
    """dbias: optional [small_ch] tensor that receives (accumulate_dbias: is added) the per-channel sum of the output
    -- the bias gradient of the transposed-conv layer whose pre-activation gradient this call produces.
    want_cmask (epi = EPI_RELU): also returns the output's channel-quad mask (uint8, numel / 4: EPI_MUL_CMASK of the
    layer's data gradient, conv_up) -- (out, cmask)."""
    nimg = big.shape[0]
    (cb, hb, _), (cs, hs, _) = conv_shapes(layer)
    assert tuple(big.shape[1:]) == (cb, hb, hb) and big.is_contiguous(), big.shape
    is_u8 = big.dtype == torch.uint8
    assert is_u8 or big.dtype == torch.float32
    if out is None:
        out = torch.empty(nimg, cs, hs, hs, dtype=torch.float32, device=big.device)
    # workspace: the channel-sum partials (dbias) and the bf16x6 kernel's weight pack (include/repo_hip.h)
    nb = lib().repo_conv_down_workspace_bytes(layer, nimg)
    ws = workspace(nb, big.device) if nb else None
    cmask = torch.empty(nimg * (cs // 4) * hs * hs, dtype=torch.uint8, device=big.device) if want_cmask else None
    check(
        lib().repo_conv_down(layer, nimg, _ptr(big), int(is_u8), _ptr(_f32c(w)), _ptr(bias), _ptr(out), epi,
                             _ptr(aux), _ptr(dbias), int(accumulate_dbias), _ptr(cmask), _ptr(ws), nb, _stream()),
        "repo_conv_down",
    )
    return (out, cmask) if want_cmask else out


def conv_up_pack(layer, w):
    """The fragment-ready weight copy conv_up works from (None for the 3-channel layers): pass it as `pack=`."""
    nb = lib().repo_conv_up_workspace_bytes(layer)
    if not nb:
        return None
    pack = torch.empty(nb, dtype=torch.uint8, device=w.device)
    check(lib().repo_conv_up_pack(layer, _ptr(_f32c(w)), _ptr(pack), nb, _stream()), "repo_conv_up_pack")
    return pack


def conv_up(layer, small, w, bias=None, epi=EPI_NONE, aux=None, out=None, pack=None):
    nimg = small.shape[0]
    (cb, hb, _), (cs, hs, _) = conv_shapes(layer)
    assert tuple(small.shape[1:]) == (cs, hs, hs) and small.is_contiguous(), small.shape
    if out is None:
        out = torch.empty(nimg, cb, hb, hb, dtype=torch.float32, device=small.device)
    nb = lib().repo_conv_up_workspace_bytes(layer)
    ws = pack if pack is not None else (workspace(nb, small.device) if nb else None)
    check(
        lib().repo_conv_up(layer, nimg, _ptr(_f32c(small)), _ptr(_f32c(w)), _ptr(bias), _ptr(out), epi, _ptr(aux),
                           int(pack is not None), _ptr(ws), ws.numel() if ws is not None else 0, _stream()),
        "repo_conv_up",
    )
    return out


def conv_wgrad(layer, small, big, dw=None, db=None, accumulate=False, want_bias=True, dbig=None):
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
        lib().repo_conv_wgrad(layer, nimg, _ptr(_f32c(small)), _ptr(big), int(is_u8), _ptr(dw), _ptr(db), _ptr(dbig),
                              int(accumulate), _ptr(ws), ws.numel(), _stream()),
        "repo_conv_wgrad",
    )
    return dw, db


def decoder_out_nll(h3, w, bias, target, grad_scale, want_recon=False, want_dpre=True, want_mask=False, dbias=None,
                    accumulate_dbias=False):
    """Final transposed conv fused with 0.5*(recon-target)^2 summed over everything.
    Returns (loss_sum (1,), dpre or None, recon or None) and, with want_mask, the quad mask of h3
    (uint8, numel/4: EPI_MUL_MASK4 of the layer's data gradient) as a fourth element.
    dbias (3,): (+)= the channel sums of dpre, the layer's bias gradient."""
    nimg = h3.shape[0]
    is_u8 = target.dtype == torch.uint8
    assert target.is_contiguous() and target.numel() == nimg * 3 * 64 * 64
    dev = h3.device
    recon = torch.empty(nimg, 3, 64, 64, dtype=torch.float32, device=dev) if want_recon else None
    dpre = torch.empty(nimg, 3, 64, 64, dtype=torch.float32, device=dev) if want_dpre else None
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    mask = torch.empty(h3.numel() // 4, dtype=torch.uint8, device=dev) if want_mask else None
    nb = lib().repo_decoder_out_nll_workspace_bytes(nimg)
    ws = workspace(nb, dev)
    check(
        lib().repo_decoder_out_nll(nimg, _ptr(_f32c(h3)), _ptr(_f32c(w)), _ptr(bias), _ptr(target), int(is_u8),
                                   float(grad_scale), _ptr(recon), _ptr(dpre), _ptr(mask), _ptr(loss), _ptr(dbias),
                                   int(accumulate_dbias), _ptr(ws), ws.numel(), _stream()),
        "repo_decoder_out_nll",
    )
    return (loss, dpre, recon, mask) if want_mask else (loss, dpre, recon)


def conv_up_nll(layer, small, w, bias, target, grad_scale, want_recon=False, want_dpre=True):
    """A 3-channel transposed conv (layer 6 or 12) fused with 0.5*(recon-target)^2 summed over everything, on the
    gather engine (any output size).  Returns (loss_sum (1,), dpre or None, recon or None)."""
    nimg = small.shape[0]
    (cb, hb, _), (cs, hs, _) = conv_shapes(layer)
    assert tuple(small.shape[1:]) == (cs, hs, hs) and small.is_contiguous(), small.shape
    is_u8 = target.dtype == torch.uint8
    assert target.is_contiguous() and target.numel() == nimg * cb * hb * hb
    dev = small.device
    recon = torch.empty(nimg, cb, hb, hb, dtype=torch.float32, device=dev) if want_recon else None
    dpre = torch.empty(nimg, cb, hb, hb, dtype=torch.float32, device=dev) if want_dpre else None
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    nb = lib().repo_conv_up_nll_workspace_bytes(layer, nimg)
    ws = workspace(nb, dev)
    check(
        lib().repo_conv_up_nll(layer, nimg, _ptr(_f32c(small)), _ptr(_f32c(w)), _ptr(bias), _ptr(target), int(is_u8),
                               float(grad_scale), _ptr(recon), _ptr(dpre), _ptr(loss), _ptr(ws), ws.numel(), _stream()),
        "repo_conv_up_nll",
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


# ----------------------------------------------------------------------------- pointer arrays
import ctypes  # noqa: E402


def ptr_array(tensors):
    """HOST array of device pointers (kept alive by the caller for the duration of the call)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else _ptr(t)
    return arr


_REDUCE_WS_BYTES = 64 << 10   # >= every reduction's request (repo_reduce / _tia_blend_nll / _grad_sqnorm _workspace_bytes)


def reduce_ws(device, nbytes=0):
    """The reduction workspace of the current stream: allocated ZEROED, used only by the reductions (the small-grid ones
    finish in their own launch: their last block takes the ticket in word 0 of the 256-byte header and leaves it zero
    again: include/repo_hip.h, "losses and regularisers") and by film_bwd_h, which keeps its epoch word at byte 32 of the
    header.  Per (device, stream) like workspace(), and owned by the innermost scratch_scope() if there is one."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = ("reduce", idx, _raw_stream(idx))
    pool = getattr(_scope, "pool", None)
    if pool is None:
        pool = _ws
    buf = pool.get(key)
    if buf is None:
        assert nbytes <= _REDUCE_WS_BYTES
        buf = pool[key] = torch.zeros(_REDUCE_WS_BYTES, dtype=torch.uint8, device=device)
    return buf


# ----------------------------------------------------------------------------- RSSM observe
_scan_status = {}


def scan_status(device):
    """The device's sticky status word for ASYNCHRONOUS scan errors (include/repo_hip.h, repo_rssm_observe_fwd: the
    column-split engine ORs REPO_SCAN_STATUS_* bits into it when a spin-wait on a peer workgroup times out).  One int32
    per device, zeroed once; every scan launch of this process passes it.  The agents append it to their per-update
    scalar copy (`Dreamer._log_update`) -- no extra transfer -- and raise; anyone else calls `check_scan_status`."""
    device = torch.device(device)
    # (a host word for CPU devices: the data-parallel status protocol is exercised over gloo without a GPU)
    idx = "cpu" if device.type == "cpu" else device.index if device.index is not None else torch.cuda.current_device()
    w = _scan_status.get(idx)
    if w is None:
        w = _scan_status[idx] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


SCAN_STATUS_TEXT = {1: "forward", 2: "reverse", 3: "forward and reverse"}


def raise_scan_status(word, consequence="the outputs of that call are NaN-poisoned"):
    """word: the status value read on the host (0 = fine)."""
    if word:
        from ._lib import RepoHipError
        raise RepoHipError(
            f"column-split observe scan ({SCAN_STATUS_TEXT.get(word & 3, word)}): a spin-wait on a peer workgroup timed "
            f"out (the group's workgroups were not co-resident); {consequence}.  "
            "REPO_SCAN_CS=0 selects the row-scan engine, which has no cross-workgroup waits.")


def check_scan_status(device):
    """Synchronising read of the status word (tests, callers outside the agents' update loop).  The word is cleared
    before the exception leaves: the fault is reported once, the caller may recover (REPO_SCAN_CS=0) and go on."""
    w = scan_status(device)
    word = int(w.item())
    if word:
        w.zero_()
    raise_scan_status(word)


def take_scan_status(device, dp=None):
    """The per-UPDATE status word: a copy of the device's sticky word, which is cleared in the same stream order --
    call it on the stream that has joined every scan of the update, before the first optimiser step.  Every step of
    the update then takes the copy as `skip` (repo_clip_adam: a faulted update leaves parameters, moments and the dual
    variable untouched) and the agent appends it to the update's scalar copy.  Data parallel: the copy is MAX-reduced
    over the ranks first, so that a timeout on ONE rank makes EVERY rank skip the same steps and raise in the same
    update (its NaN gradient has been summed into every replica's buffer by then)."""
    w = scan_status(device)
    st = torch.empty_like(w)
    take_status_into(w, st)
    if dp is not None:
        dp.all_reduce_status(st)
    return st


def take_status_into(sticky, taken):
    """*taken = *sticky; *sticky = 0 in ONE launch on a HIP device (repo_take_status); plain tensor ops on the host."""
    if sticky.is_cuda:
        check(lib().repo_take_status(_ptr(sticky), _ptr(taken), _stream()), "repo_take_status")
    else:
        taken.copy_(sticky)
        sticky.zero_()


class ObserveSaved:
    __slots__ = ("T", "B", "A", "D", "Hd", "S", "E", "featx", "prior_state", "prior_mean", "prior_std", "post_mean",
                 "post_std", "xsa", "e", "gates", "hp", "hq", "nonterms", "embeds", "eps_prior", "eps_post", "noise",
                 "prior_ready", "cs")


def rssm_observe_fwd(params, prev_belief, prev_state, actions, nonterms, embeds, eps_prior, eps_post, min_std=0.1,
                     noise=(0, 0), prior_only=False, prior_stream=None):
    """params: list of the 14 TransitionModel tensors in state_dict order.  Time-major inputs.
    eps_prior = eps_post = None: the kernel draws its noise from Philox stream noise = (seed, offset).
    prior_stream: a side stream -- the scan then leaves the prior head out (it depends on belief_t only, not on the
    recurrence) and repo_rssm_prior_head evaluates it for all steps on that stream, beside whatever the caller
    enqueues next; `sv.prior_ready` (a stream to wait for) guards sv.prior_* and sv.hp."""
    T, B, A = actions.shape
    D, S = prev_belief.shape[1], prev_state.shape[1]
    Hd = params[6].shape[0]
    E = embeds.shape[-1]
    dev = actions.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    sv = ObserveSaved()
    sv.T, sv.B, sv.A, sv.D, sv.Hd, sv.S, sv.E = T, B, A, D, Hd, S, E
    sv.featx = f(T + 1, B, D + S)
    sv.prior_state, sv.prior_mean, sv.prior_std = f(T, B, S), f(T, B, S), f(T, B, S)
    sv.post_mean, sv.post_std = f(T, B, S), f(T, B, S)
    sv.xsa, sv.e, sv.gates, sv.hp, sv.hq = f(T, B, S + A), f(T, B, D), f(T, B, 4 * D), f(T, B, Hd), f(T, B, Hd)
    eemb = f(T, B, Hd)
    sv.nonterms = _f32c(nonterms.reshape(T, B))
    sv.embeds = _f32c(embeds)
    sv.eps_prior = _f32c(eps_prior) if eps_prior is not None else None
    sv.eps_post = _f32c(eps_post) if eps_post is not None else None
    sv.noise = (int(noise[0]), int(noise[1]))
    nb = lib().repo_rssm_observe_fwd_workspace_bytes(T, B, A, D, Hd, S, E)
    ws = workspace(nb, dev)
    pa = ptr_array(params)
    # engine: the column-split, weight-stationary MFMA scan (csrc/scan_cs.hip, mode 3): 12.5 us per step for any
    # B <= 64 (13 workgroups per 16 rows) against the row scan's 18-20; measured alone (fwd 880-980 -> 590-640 us,
    # reverse 890-1030 -> 655-750) and inside the update (B=50: 8.51 -> 8.35 ms, the B=7 shard 3.37 -> 2.97 ms;
    # profiles/r03_scan_cs.txt).  REPO_SCAN_CS=1 / 0 forces it on / off for any B.  It always leaves the prior head to
    # repo_rssm_prior_head.
    cs_env = os.environ.get("REPO_SCAN_CS", "auto")
    use_cs = (not prior_only and cs_env != "0" and (cs_env == "1" or B <= 64)
              and (D + 15) // 16 == 13 and (Hd + 15) // 16 == 13 and (S + A + 15) // 16 == 3 and S <= 32 and D % 4 == 0)
    hoist = (prior_stream is not None or use_cs) and not prior_only
    check(
        lib().repo_rssm_observe_fwd(
            T, B, A, D, Hd, S, E, pa, _ptr(_f32c(prev_belief)), _ptr(_f32c(prev_state)), _ptr(_f32c(actions)),
            _ptr(sv.nonterms), _ptr(sv.embeds), _ptr(sv.eps_prior), _ptr(sv.eps_post), sv.noise[0], sv.noise[1],
            float(min_std), _ptr(sv.featx), _ptr(sv.prior_state), _ptr(sv.prior_mean), _ptr(sv.prior_std), _ptr(sv.post_mean),
            _ptr(sv.post_std), _ptr(sv.xsa), _ptr(sv.e), _ptr(sv.gates), _ptr(sv.hp), _ptr(sv.hq), _ptr(eemb),
            3 if use_cs else 2 if hoist else int(prior_only), _ptr(scan_status(dev)), _ptr(ws), ws.numel(), _stream(),
        ),
        "repo_rssm_observe_fwd",
    )
    sv.prior_ready = None
    sv.cs = use_cs   # the reverse scan takes the same engine
    if hoist and prior_stream is None:   # the prior head of all steps, in line
        nbp = lib().repo_rssm_prior_head_workspace_bytes(T, B, S)
        wsp = workspace(nbp, dev)
        check(
            lib().repo_rssm_prior_head(T, B, D, Hd, S, pa, _ptr(sv.featx), _ptr(sv.eps_prior), sv.noise[0], sv.noise[1],
                                       float(min_std), _ptr(sv.hp), _ptr(sv.prior_state), _ptr(sv.prior_mean),
                                       _ptr(sv.prior_std), _ptr(wsp), nbp, _stream()),
            "repo_rssm_prior_head",
        )
    elif hoist:
        main = torch.cuda.current_stream(dev)
        prior_stream.wait_stream(main)
        nbp = lib().repo_rssm_prior_head_workspace_bytes(T, B, S)
        with torch.cuda.stream(prior_stream):
            wsp = torch.empty(nbp, dtype=torch.uint8, device=dev)  # its own scratch: the shared workspace is the main stream's
            check(
                lib().repo_rssm_prior_head(T, B, D, Hd, S, pa, _ptr(sv.featx), _ptr(sv.eps_prior), sv.noise[0], sv.noise[1],
                                           float(min_std), _ptr(sv.hp), _ptr(sv.prior_state), _ptr(sv.prior_mean),
                                           _ptr(sv.prior_std), _ptr(wsp), nbp, prior_stream.cuda_stream),
                "repo_rssm_prior_head",
            )
        for t in (sv.hp, sv.prior_state, sv.prior_mean, sv.prior_std):
            t.record_stream(prior_stream)
        sv.prior_ready = prior_stream
    return sv


def rssm_observe_bwd(params, sv, dparams, dfeat=None, dprior_state=None, dpm=None, dps=None, dqm=None, dqs=None,
                     dembeds=None, dprev_belief=None, dprev_state=None, accumulate=False, min_std=0.1):
    dev = sv.featx.device
    nb = lib().repo_rssm_observe_bwd_workspace_bytes(sv.T, sv.B, sv.A, sv.D, sv.Hd, sv.S, sv.E)
    ws = workspace(nb, dev)
    pa, ga = ptr_array(params), ptr_array(dparams)
    check(
        lib().repo_rssm_observe_bwd(
            sv.T, sv.B, sv.A, sv.D, sv.Hd, sv.S, sv.E, pa, _ptr(sv.nonterms), _ptr(sv.embeds), _ptr(sv.eps_prior),
            _ptr(sv.eps_post), sv.noise[0], sv.noise[1], float(min_std), _ptr(sv.featx), _ptr(sv.prior_std), _ptr(sv.post_std), _ptr(sv.xsa),
            _ptr(sv.e), _ptr(sv.gates), _ptr(sv.hp), _ptr(sv.hq), _ptr(dfeat), _ptr(dprior_state), _ptr(dpm),
            _ptr(dps), _ptr(dqm), _ptr(dqs), ga, _ptr(dembeds), _ptr(dprev_belief), _ptr(dprev_state),
            int(bool(accumulate)) | (2 if getattr(sv, "cs", False) else 0), _ptr(scan_status(dev)), _ptr(ws), ws.numel(),
            _stream(),
        ),
        "repo_rssm_observe_bwd",
    )


# ----------------------------------------------------------------------------- MLP heads
def mlp_fwd(params, x, out=None, hid=None):
    """params: [w1,b1,...,wL,bL]; x (rows, in_dim) view with contiguous rows.  Returns (out, hidden list).
    `hid`: optional caller-provided (rows, hidden) buffers for the L-1 hidden activations."""
    L = len(params) // 2
    rows, in_dim = x.shape
    hidden = params[0].shape[0] if L > 1 else 0
    out_dim = params[-2].shape[0]
    dev = x.device
    if hid is None:
        hid = [torch.empty(rows, hidden, dtype=torch.float32, device=dev) for _ in range(L - 1)]
    if out is None:
        out = torch.empty(rows, out_dim, dtype=torch.float32, device=dev)
    pa, ha = ptr_array(params), ptr_array(hid)
    nb = lib().repo_mlp_fwd_workspace_bytes(rows, in_dim, max(hidden, 1), out_dim, L)
    ws = workspace(nb, dev) if nb else None
    check(
        lib().repo_mlp_fwd(rows, in_dim, max(hidden, 1), out_dim, L, _ptr(x), _ld(x), pa, ha, _ptr(out), _ld(out),
                           _ptr(ws), nb, _stream()),
        "repo_mlp_fwd",
    )
    return out, hid


def mlp_bwd(params, x, hid, dout, dparams=None, accumulate_w=False, dx=None, accumulate_dx=False, dout_w=None):
    """dout_w (rows_w <= rows, 1): a scalar head differentiated for TWO losses in one reverse chain -- dx from `dout`
    over all rows, dparams from `dout_w` over the first rows_w rows (include/repo_hip.h, repo_mlp_bwd)."""
    L = len(params) // 2
    rows, in_dim = x.shape
    hidden = params[0].shape[0] if L > 1 else 1
    out_dim = params[-2].shape[0]
    nb = lib().repo_mlp_bwd_workspace_bytes(rows, in_dim, hidden, out_dim, L)
    ws = workspace(nb, x.device)
    pa, ha = ptr_array(params), ptr_array(hid)
    ga = ptr_array(dparams) if dparams is not None else None
    rows_w = 0
    if dout_w is not None:
        rows_w = dout_w.shape[0]
        assert dout_w.is_contiguous() and dout_w.numel() == rows_w and dout.shape[1] == 1
    check(
        lib().repo_mlp_bwd(rows, in_dim, hidden, out_dim, L, _ptr(x), _ld(x), pa, ha, _ptr(dout), _ld(dout), ga,
                           int(accumulate_w), _ptr(dx), _ld(dx) if dx is not None else 0, int(accumulate_dx),
                           _ptr(dout_w), rows_w, _ptr(ws), ws.numel(), _stream()),
        "repo_mlp_bwd",
    )


def actor_head_fwd(raw, min_std=0.1, init_std=0.0, mean_scale=5.0, eps=None, state=None, mean=None, std=None):
    rows, A2 = raw.shape
    A = A2 // 2
    dev = raw.device
    if mean is None:
        mean = torch.empty(rows, A, dtype=torch.float32, device=dev)
    if std is None:
        std = torch.empty(rows, A, dtype=torch.float32, device=dev)
    xsa = None
    S = 0
    if eps is not None:
        S = state.shape[1]
        xsa = torch.empty(rows, S + A, dtype=torch.float32, device=dev)
    check(
        lib().repo_actor_head_fwd(rows, A, S, _ptr(_f32c(raw)), _ptr(eps), _ptr(state),
                                  _ld(state) if state is not None else 0, min_std, init_std, mean_scale, _ptr(mean),
                                  _ptr(std), _ptr(xsa), _stream()),
        "repo_actor_head_fwd",
    )
    return mean, std, xsa


def actor_head_bwd(mean, std, dmean=None, dstd=None, daction=None, action=None, eps=None, min_std=0.1, mean_scale=5.0,
                   out=None, accumulate=False):
    rows, A = mean.shape
    draw = out if out is not None else torch.empty(rows, 2 * A, dtype=torch.float32, device=mean.device)
    check(
        lib().repo_actor_head_bwd(rows, A, _ptr(dmean), _ptr(dstd), _ptr(daction),
                                  _ld(daction) if daction is not None else 0, _ptr(action),
                                  _ld(action) if action is not None else 0, _ptr(eps), _ptr(mean), _ptr(std), min_std,
                                  mean_scale, _ptr(draw), int(accumulate), _stream()),
        "repo_actor_head_bwd",
    )
    return draw


# ----------------------------------------------------------------------------- imagination
class ImagineSaved:
    __slots__ = ("Hm", "N", "A", "D", "Hd", "S", "featx", "prior_mean", "prior_std", "a_hidden", "a_raw", "a_mean",
                 "a_std", "xsa", "e", "gates", "hp", "eps_act", "eps_prior", "noise", "C")


def rssm_imagine_fwd(rssm_params, actor_params, belief0, state0, eps_act, eps_prior, min_std=0.1, a_min_std=0.1,
                     a_init_std=0.0, a_mean_scale=5.0, spare_slot=False, noise=(0, 0), horizon=None, cond=None):
    """spare_slot: allocate the saved actor tensors with one extra step slot ((Hm+1)*N rows) so the
    caller can evaluate the actor on the final imagined state into the same buffers.
    eps_act = eps_prior = None (+ horizon = Hm): the kernel draws its noise from Philox stream noise = (seed, offset).
    cond (N, C): the multitask agents' conditioned rollout (include/repo_hip.h); the actor's fc1 and W_sa carry C more
    input columns."""
    C = 0 if cond is None else cond.shape[1]
    if eps_act is None:
        Hm, N, A = int(horizon), belief0.shape[0], actor_params[-1].shape[0] // 2
    else:
        Hm, N, A = eps_act.shape
    D, S = belief0.shape[1], state0.shape[1]
    Hd = rssm_params[6].shape[0]
    La = len(actor_params) // 2
    dev = belief0.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    sv = ImagineSaved()
    sv.Hm, sv.N, sv.A, sv.D, sv.Hd, sv.S = Hm, N, A, D, Hd, S
    sv.C = C
    sv.featx = f(Hm + 1, N, D + S)
    sv.prior_mean, sv.prior_std = f(Hm, N, S), f(Hm, N, S)
    ar = (Hm + 1) * N if spare_slot else Hm * N
    sv.a_hidden = f(La - 1, ar, Hd)
    sv.a_raw, sv.a_mean, sv.a_std = f(ar, 2 * A), f(ar, A), f(ar, A)
    sv.xsa, sv.e, sv.gates, sv.hp = f(Hm * N, S + A + C), f(Hm * N, D), f(Hm * N, 4 * D), f(Hm * N, Hd)
    sv.eps_act = _f32c(eps_act) if eps_act is not None else None
    sv.eps_prior = _f32c(eps_prior) if eps_prior is not None else None
    sv.noise = (int(noise[0]), int(noise[1]))
    nb = lib().repo_rssm_imagine_fwd_workspace_bytes(Hm, N, A, D, Hd, S)
    ws = workspace(nb, dev)
    ra, aa = ptr_array(rssm_params), ptr_array(actor_params)
    check(
        lib().repo_rssm_imagine_fwd(
            Hm, N, A, D, Hd, S, La, ra, aa, _ptr(_f32c(belief0)), _ptr(_f32c(state0)),
            _ptr(_f32c(cond)) if C else None, C, _ptr(sv.eps_act),
            _ptr(sv.eps_prior), sv.noise[0], sv.noise[1], min_std, a_min_std, a_init_std, a_mean_scale, _ptr(sv.featx), _ptr(sv.prior_mean),
            _ptr(sv.prior_std), _ptr(sv.a_hidden), ar, _ptr(sv.a_raw), _ptr(sv.a_mean), _ptr(sv.a_std), _ptr(sv.xsa),
            _ptr(sv.e), _ptr(sv.gates), _ptr(sv.hp), _ptr(ws), ws.numel(), _stream(),
        ),
        "repo_rssm_imagine_fwd",
    )
    return sv


def rssm_imagine_bwd(rssm_params, sv, dfeat, dprior_mean=None, dprior_std=None, want_dfeat0=False, min_std=0.1,
                     a_min_std=0.1, a_mean_scale=5.0, d_araw=None):
    dev = dfeat.device
    if d_araw is None:
        d_araw = torch.empty(sv.Hm * sv.N, 2 * sv.A, dtype=torch.float32, device=dev)
    dfeat0 = torch.empty(sv.N, sv.D + sv.S, dtype=torch.float32, device=dev) if want_dfeat0 else None
    nb = lib().repo_rssm_imagine_bwd_workspace_bytes(sv.Hm, sv.N, sv.A, sv.D, sv.Hd, sv.S)
    ws = workspace(nb, dev)
    ra = ptr_array(rssm_params)
    check(
        lib().repo_rssm_imagine_bwd(
            sv.Hm, sv.N, sv.A, sv.D, sv.Hd, sv.S, getattr(sv, "C", 0), ra, _ptr(sv.eps_act), _ptr(sv.eps_prior),
            sv.noise[0], sv.noise[1], min_std, a_min_std, a_mean_scale, _ptr(sv.featx), _ptr(sv.prior_std), _ptr(sv.a_mean), _ptr(sv.a_std), _ptr(sv.xsa),
            _ptr(sv.e), _ptr(sv.gates), _ptr(sv.hp), _ptr(_f32c(dfeat)), _ptr(dprior_mean), _ptr(dprior_std),
            _ptr(d_araw), _ptr(dfeat0), _ptr(ws), ws.numel(), _stream(),
        ),
        "repo_rssm_imagine_bwd",
    )
    return d_araw, dfeat0


# ----------------------------------------------------------------------------- losses
def kl_balance(pm, ps, qm, qs, mode, alpha, log_beta, free_nats, scale, want_grads=True):
    S = pm.shape[-1]
    rows = pm.numel() // S
    dev = pm.device
    g = [torch.empty_like(pm) for _ in range(4)] if want_grads else [None] * 4
    out = torch.empty(1, dtype=torch.float32, device=dev)
    ws = reduce_ws(dev)
    check(
        lib().repo_kl_balance(rows, S, _ptr(_f32c(pm)), _ptr(_f32c(ps)), _ptr(_f32c(qm)), _ptr(_f32c(qs)), mode,
                              float(alpha), _ptr(log_beta), float(free_nats), float(scale), _ptr(g[0]), _ptr(g[1]),
                              _ptr(g[2]), _ptr(g[3]), _ptr(out), _ptr(ws), ws.numel(), _stream()),
        "repo_kl_balance",
    )
    return out, g


def dual_step(log_beta, exp_avg, exp_avg_sq, kl_sum, rows, target_kl, lr, step, apply=True, betas=(0.9, 0.999),
              eps=1e-8, out=None, skip=None):
    """skip: the update's status word (int32 device tensor, see `take_scan_status`): non-zero = leave log_beta alone."""
    if out is None:
        out = torch.empty(4, dtype=torch.float32, device=log_beta.device)
    check(
        lib().repo_dual_step(_ptr(log_beta), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(kl_sum), rows, float(target_kl),
                             float(lr), betas[0], betas[1], eps, step, int(apply), _ptr(out), _ptr(skip), _stream()),
        "repo_dual_step",
    )
    return out


def scalar_nll(pred, target, mask, scale, want_grad=True, out=None):
    n = pred.numel()
    dev = pred.device
    dpred = torch.empty(n, dtype=torch.float32, device=dev) if want_grad else None
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=dev)
    ws = reduce_ws(dev)
    check(
        lib().repo_scalar_nll(n, _ptr(_f32c(pred)), _ptr(_f32c(target)), _ptr(mask), float(scale), _ptr(dpred),
                              _ptr(out), _ptr(ws), ws.numel(), _stream()),
        "repo_scalar_nll",
    )
    return out, dpred


def tia_blend_nll(t_out, d_out, mask_wb, target, grad_scale, want_grads=True, want_recon=False, inplace=False):
    """TIA's masked blend of the task / distractor decoder outputs + pixel NLL (tia.py:123-133).
    t_out, d_out (n,6,H,W); mask_wb (7,) = mask_head weight(6) + bias; target (n,3,H,W) uint8 | float32.
    Returns (sums8 = [loss_sum, d/dw(6), d/db], dt_out, dd_out, recon); inplace: the gradients overwrite t_out, d_out."""
    n = t_out.shape[0]
    pixels = t_out.shape[2] * t_out.shape[3]
    assert t_out.shape == d_out.shape and t_out.shape[1] == 6 and t_out.is_contiguous() and d_out.is_contiguous()
    assert target.is_contiguous() and target.numel() == n * 3 * pixels and mask_wb.numel() == 7
    dev = t_out.device
    dt = (t_out if inplace else torch.empty_like(t_out)) if want_grads else None
    dd = (d_out if inplace else torch.empty_like(d_out)) if want_grads else None
    recon = torch.empty(n, 3, *t_out.shape[2:], dtype=torch.float32, device=dev) if want_recon else None
    sums = torch.empty(8, dtype=torch.float32, device=dev)
    ws = reduce_ws(dev, lib().repo_tia_blend_nll_workspace_bytes())
    check(
        lib().repo_tia_blend_nll(n, pixels, _ptr(_f32c(t_out)), _ptr(_f32c(d_out)), _ptr(_f32c(mask_wb)), _ptr(target),
                                 int(target.dtype == torch.uint8), float(grad_scale), _ptr(dt), _ptr(dd), _ptr(recon),
                                 _ptr(sums), _ptr(ws), ws.numel(), _stream()),
        "repo_tia_blend_nll",
    )
    return sums, dt, dd, recon


def tanh_normal_entropy(mean, std, eps, gscale=0.0, want_grads=True, noise=(0, 0), samples=None):
    """eps (samples, rows, A), or None (+ samples): drawn in-kernel from Philox stream noise = (seed, offset)."""
    rows, A = mean.shape
    NS = eps.shape[0] if eps is not None else int(samples)
    assert eps is None or eps.numel() == NS * rows * A
    dev = mean.device
    dmean = torch.empty_like(mean) if want_grads else None
    dstd = torch.empty_like(std) if want_grads else None
    out = torch.empty(1, dtype=torch.float32, device=dev)
    ws = reduce_ws(dev)
    check(
        lib().repo_tanh_normal_entropy(rows, A, NS, _ptr(_f32c(mean)), _ptr(_f32c(std)),
                                       _ptr(_f32c(eps)) if eps is not None else None, int(noise[0]), int(noise[1]),
                                       float(gscale), _ptr(dmean), _ptr(dstd), _ptr(out), _ptr(ws), ws.numel(),
                                       _stream()),
        "repo_tanh_normal_entropy",
    )
    return out, dmean, dstd


def tanh_normal_mode(mean, std, eps):
    rows, A = mean.shape
    action = torch.empty_like(mean)
    check(
        lib().repo_tanh_normal_mode(rows, A, eps.shape[0], _ptr(_f32c(mean)), _ptr(_f32c(std)), _ptr(_f32c(eps)),
                                    _ptr(action), _stream()),
        "repo_tanh_normal_mode",
    )
    return action


def normal_entropy(std, gscale=0.0, want_grad=False):
    dev = std.device
    dstd = torch.empty_like(std) if want_grad else None
    out = torch.empty(1, dtype=torch.float32, device=dev)
    ws = reduce_ws(dev)
    check(
        lib().repo_normal_entropy(std.numel(), _ptr(_f32c(std)), float(gscale), _ptr(dstd), _ptr(out), _ptr(ws),
                                  ws.numel(), _stream()),
        "repo_normal_entropy",
    )
    return out, dstd


def lambda_return(rewards, values, gamma, lambda_, gret=0.0, want_grads=True):
    Hm, N = rewards.shape
    dev = rewards.device
    returns = torch.empty(Hm - 1, N, dtype=torch.float32, device=dev)
    dr = torch.empty(Hm, N, dtype=torch.float32, device=dev) if want_grads else None
    dv = torch.empty(Hm, N, dtype=torch.float32, device=dev) if want_grads else None
    out = torch.empty(1, dtype=torch.float32, device=dev)
    ws = reduce_ws(dev)
    check(
        lib().repo_lambda_return(Hm, N, _ptr(_f32c(rewards)), _ptr(_f32c(values)), float(gamma), float(lambda_),
                                 float(gret), _ptr(returns), _ptr(dr), _ptr(dv), _ptr(out), _ptr(ws), ws.numel(),
                                 _stream()),
        "repo_lambda_return",
    )
    return returns, dr, dv, out


# ----------------------------------------------------------------------------- optimiser
def grad_sqnorm(g, out=None):
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=g.device)
    ws = reduce_ws(g.device, lib().repo_grad_sqnorm_workspace_bytes())
    check(lib().repo_grad_sqnorm(g.numel(), _ptr(_f32c(g)), _ptr(out), _ptr(ws), ws.numel(), _stream()),
          "repo_grad_sqnorm")
    return out


def clip_adam(p, g, m, v, sqnorm, max_norm, lr, step, betas=(0.9, 0.999), eps=1e-8, skip=None):
    """skip: the update's status word (int32 device tensor, see `take_scan_status`): if it is non-zero when the kernel
    runs, parameters and moments are left untouched (include/repo_hip.h, repo_clip_adam)."""
    check(
        lib().repo_clip_adam(p.numel(), _ptr(_f32c(p)), _ptr(_f32c(g)), _ptr(_f32c(m)), _ptr(_f32c(v)), _ptr(sqnorm),
                             float(max_norm), float(lr), betas[0], betas[1], eps, step, _ptr(skip), _stream()),
        "repo_clip_adam",
    )


def philox_normal(n, seed, offset, device):
    """The n standard normals [offset, offset+n) of Philox stream `seed` -- what a kernel given (seed, offset)
    instead of a noise tensor draws for elements 0..n-1."""
    out = torch.empty(int(n), dtype=torch.float32, device=device)
    check(lib().repo_philox_normal(_ptr(out), int(n), int(seed), int(offset), _stream()), "repo_philox_normal")
    return out


# ----------------------------------------------------------------------------- multitask (task-conditioned) agents
def film_fwd(y, film, gamma_off, beta_off, out=None):
    """h = relu((1 + gamma) * y + beta) per (image, channel) plane; y (n, C, ...) NCHW, film (n, ld) the FiLM layer's
    output, gamma / beta at column offsets gamma_off / beta_off (models/encoder.py:75-88 of the reference)."""
    n, C = y.shape[:2]
    P = y.numel() // (n * C)
    if out is None:
        out = torch.empty_like(y)
    check(lib().repo_film_fwd(n, C, P, _ptr(y), _ptr(film), film.shape[1], gamma_off, beta_off, _ptr(out), _stream()),
          "repo_film_fwd")
    return out


def film_tables(film, channels):
    """The FiLM tables of one conv stack's modulated layers (EPI_FILM_RELU's aux) in one launch: film (n, 2 * sum(channels))
    -> a list of (n, 2, C_l) tensors [1 + gamma | beta] (views of one buffer)."""
    n = film.shape[0]
    assert film.shape[1] >= 2 * sum(channels) and film.is_contiguous() and 1 <= len(channels) <= 4
    buf = torch.empty(n * 2 * sum(channels), dtype=torch.float32, device=film.device)
    arr = (ctypes.c_int * len(channels))(*channels)
    check(lib().repo_film_tables(n, len(channels), arr, _ptr(film), film.shape[1], _ptr(buf), _stream()), "repo_film_tables")
    out, o = [], 0
    for c in channels:
        out.append(buf[o : o + n * 2 * c].view(n, 2, c))
        o += n * 2 * c
    return out


FILM_CONV_DOWN, FILM_CONV_UP, FILM_DENSE = 1, 2, 3


def film_bwd_h(dh, h, film, gamma_off, beta_off, dfilm, dy=None, exact=None):
    """film_bwd for a layer that ran with EPI_FILM_RELU: only its output h exists, y is recovered from it -- except on
    planes with |1 + gamma| < 1e-3, which recompute y exactly from the layer itself: exact = (FILM_CONV_DOWN | FILM_CONV_UP,
    layer id, x, w, bias) or (FILM_DENSE, K, x (n, K), w (K, C * P), bias (C))."""
    n, C = h.shape[:2]
    P = h.numel() // (n * C)
    if dy is None:
        dy = torch.empty_like(h)
    kind, geo, x, w, bias = 0, None, None, None, None
    if exact is not None:
        kind, which, x, w, bias = exact
        # FILM_DENSE: which = K, or (K, True) when `bias` holds one value per output element (the composed decoder head)
        dense = which if isinstance(which, tuple) else (which, False)
        geo = (ctypes.c_int64 * 4)(*(CONV_GEO[which] if kind != FILM_DENSE else (int(dense[0]), int(bool(dense[1])), 0, 0)))
        assert x.is_contiguous() and w.is_contiguous() and x.dtype in (torch.float32, torch.uint8)
    # the "a gated-off plane was met" word: word 8 of the stream's zeroed reduction workspace, stamped with a call counter
    global _film_epoch
    _film_epoch = _film_epoch % 0xFFFFFFF0 + 1
    word = reduce_ws(h.device)[32:36]
    check(lib().repo_film_bwd_h(n, C, P, _ptr(dh), _ptr(h), _ptr(film), film.shape[1], gamma_off, beta_off, _ptr(dy),
                                _ptr(dfilm), kind, geo, _ptr(x), int(x is not None and x.dtype == torch.uint8),
                                _ptr(_f32c(w)) if w is not None else None, _ptr(bias), _ptr(word), _film_epoch,
                                _stream()), "repo_film_bwd_h")
    return dy


_film_epoch = 0


def film_bwd(dh, y, film, gamma_off, beta_off, dfilm, dy=None):
    """dh: gradient at the ReLU's input (mask applied).  Returns dy = dh * (1 + gamma); writes the (n, C) gamma and
    beta gradient blocks of dfilm (n, ld)."""
    n, C = y.shape[:2]
    P = y.numel() // (n * C)
    if dy is None:
        dy = torch.empty_like(y)
    check(lib().repo_film_bwd(n, C, P, _ptr(dh), _ptr(y), _ptr(film), film.shape[1], gamma_off, beta_off, _ptr(dy),
                              _ptr(dfilm), _stream()), "repo_film_bwd")
    return dy


def kl_balance_tasks(pm, ps, qm, qs, alpha, log_beta, tasks, target_kl, scale):
    """MultitaskRePo's KL balance with the per-row multiplier exp(tasks_row . log_beta) (repo_mt.py:75-93).
    Returns (sums (3 + C,), [dpm, dps, dqm, dqs])."""
    S = pm.shape[-1]
    rows = pm.numel() // S
    C = log_beta.numel()
    dev = pm.device
    g = [torch.empty_like(pm) for _ in range(4)]
    sums = torch.empty(3 + C, dtype=torch.float32, device=dev)
    nb = lib().repo_kl_balance_tasks_workspace_bytes()
    ws = workspace(nb, dev)
    check(
        lib().repo_kl_balance_tasks(rows, S, C, _ptr(_f32c(pm)), _ptr(_f32c(ps)), _ptr(_f32c(qm)), _ptr(_f32c(qs)),
                                    float(alpha), _ptr(log_beta), _ptr(_f32c(tasks)), float(target_kl), float(scale),
                                    _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(g[3]), _ptr(sums), _ptr(ws), ws.numel(),
                                    _stream()),
        "repo_kl_balance_tasks",
    )
    return sums, g


def dual_step_tasks(log_beta, exp_avg, exp_avg_sq, sums, rows, lr, betas, eps, step, apply=True, out=None, skip=None):
    C = log_beta.numel()
    if out is None:
        out = torch.empty(3 + C, dtype=torch.float32, device=log_beta.device)
    check(lib().repo_dual_step_tasks(C, _ptr(log_beta), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(sums), int(rows), float(lr),
                                     float(betas[0]), float(betas[1]), float(eps), int(step), int(bool(apply)), _ptr(out),
                                     _ptr(skip), _stream()), "repo_dual_step_tasks")
    return out
