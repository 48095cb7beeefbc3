"""Op-level parity of the HIP kernels (through the C ABI) against fp64 PyTorch on CPU.

fp32 tolerance: the kernels accumulate in fp32 (v_mfma_f32_32x32x2_f32 = chained fmaf),
so the normwise error against an fp64 reference is bounded by ~K * 2^-24 * sum|a*b|;
1e-5 relative to the largest output is used throughout (stated per assert).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import log, relerr, rnd

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from repo_amd import ops as o

    return o


def dev(t):
    return t.cuda()


@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(50, 200, 36), (2450, 200, 230), (784, 1024, 230), (33, 60, 200), (700, 640, 130), (1, 1, 1), (65, 129, 17), (3000, 200, 1), (300, 7, 1), (1, 200, 230), (5, 60, 200), (8, 600, 1224)])
def test_gemm_layouts(ops, ta, tb, M, N, K):
    rs = np.random.RandomState(M * 7 + N * 3 + K)
    A = rnd(rs, K, M) if ta else rnd(rs, M, K)
    B = rnd(rs, N, K) if tb else rnd(rs, K, N)
    bias = rnd(rs, N)
    want = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double()) + bias.double()
    got = ops.gemm(dev(A), dev(B), ta, tb, bias=dev(bias))
    e = relerr(got, want)
    log(f"gemm ta={ta} tb={tb} {M}x{N}x{K}: relerr {e:.2e}")
    assert e < TOL


def test_gemm_strided_and_epilogues(ops):
    rs = np.random.RandomState(5)
    M, N, K = 300, 200, 230
    feat = rnd(rs, M, 260)  # x is a column slice of a wider buffer
    x = feat[:, 10:240]
    W, b = rnd(rs, N, K, scale=0.1), rnd(rs, N)
    pre = x.double() @ W.double().t() + b.double()
    fd = dev(feat)
    for epi, fn in [(ops.EPI_ELU, F.elu), (ops.EPI_RELU, F.relu), (ops.EPI_NONE, lambda v: v)]:
        got = ops.gemm(fd[:, 10:240], dev(W), transb=True, bias=dev(b), epi=epi)
        e = relerr(got, fn(pre))
        log(f"gemm epi={epi}: relerr {e:.2e}")
        assert e < TOL
    # backward-data with the activation derivative fused + accumulate into a strided slice
    h = F.elu(pre).float()
    dy = rnd(rs, M, N)
    want = (dy.double() * torch.where(h > 0, torch.ones_like(h), h + 1).double()) 
    got = ops.gemm(dev(dy), dev(torch.eye(N)), epi=ops.EPI_MUL_DELU, aux=dev(h))
    assert relerr(got, want) < TOL
    dx_want = dy.double() @ W.double()
    outbuf = torch.ones(M, 260).cuda()
    ops.gemm(dev(dy), dev(W), out=outbuf[:, 10:240], accumulate=True)
    assert relerr(outbuf[:, 10:240], dx_want + 1) < TOL
    assert float((outbuf[:, :10] - 1).abs().max()) == 0 and float((outbuf[:, 240:] - 1).abs().max()) == 0
    hr = F.relu(pre).float()
    got = ops.gemm(dev(dy), dev(torch.eye(N)), epi=ops.EPI_MUL_DRELU, aux=dev(hr))
    assert relerr(got, dy.double() * (hr > 0)) < TOL
    # bias_div (channel bias over 25 pixels)
    W2, b2 = rnd(rs, K, 75), rnd(rs, 3)
    got = ops.gemm(fd[:, 10:240], dev(W2), bias=dev(b2), bias_div=25, epi=ops.EPI_RELU)
    want = F.relu(x.double() @ W2.double() + b2.double().repeat_interleave(25))
    assert relerr(got, want) < TOL


BG_SHAPES = [
    # (ta, tb, M, N, K): the decoder's three big products and ragged variants of each operand layout
    (False, False, 2450, 3200, 1024),   # conv1 forward: h1 = h0 @ W1            (256 x 128 tiles)
    (False, True, 2450, 1024, 3200),    # conv1 data gradient: dh0 = d1 @ W1^T   (128 x 128 tiles, K 32 per stage)
    (True, False, 1024, 3200, 2450),    # conv1 weight gradient through repo_gemm's layout: A[k][m], B[k][n]
    (False, True, 1301, 1932, 1001),    # nothing divides anything: M, N ragged in the last tiles, K % 16 = 9, K % 4 = 1
    (False, False, 1300, 1940, 130),     # shortest K the engine takes (two register sets, 5 stages of 32)
    (True, True, 1284, 1937, 777),       # A[k][m] with B[n][k]
    (True, False, 1280, 2048, 16),      # below the engine's K: stays on the fp32 tiles (dispatch boundary)
]


@pytest.mark.parametrize("ta,tb,M,N,K", BG_SHAPES)
def test_bgemm_matches_fp64_and_the_fp32_engine(ops, ta, tb, M, N, K):
    """The bf16x6 dense engine (csrc/bgemm.h: six exact bf16 partial products per fp32 multiply, fp32 accumulation) on
    every operand layout, ragged tiles and K tails, with bias / bias_div / ReLU / accumulate, against fp64 -- and against
    the fp32-MFMA tile engine on the SAME operands (repo_debug_bgemm(0)): its error is not allowed to exceed that
    engine's by more than 25 % (measured: equal or smaller; both are set by the fp32 accumulation over K)."""
    from repo_amd._lib import lib

    rs = np.random.RandomState(M + 3 * N + 7 * K)
    A = rnd(rs, K, M) if ta else rnd(rs, M, K)
    B = rnd(rs, N, K) if tb else rnd(rs, K, N)
    A[::7] *= 40.0      # a wide dynamic range inside every dot product
    bias = rnd(rs, (N + 24) // 25)
    opA, opB = (A.double().t() if ta else A.double()), (B.double().t() if tb else B.double())
    pre = opA @ opB + bias.double().repeat_interleave(25)[:N]
    mag = opA.abs() @ opB.abs()
    # k-contiguous operands as column slices of wider buffers: the leading dimension stays a multiple of 4 (what the
    # engine asks for) while K itself is ragged
    def widen(t):
        wide = torch.zeros(t.shape[0], (t.shape[1] + 7) // 4 * 4)
        wide[:, : t.shape[1]] = t
        return dev(wide)[:, : t.shape[1]]

    dA = dev(A) if ta else widen(A)
    dB = widen(B) if tb else dev(B)
    dbias = dev(bias)
    errs = {}
    for engine in (1, 0):
        prev = lib().repo_debug_bgemm(engine)
        try:
            got = ops.gemm(dA, dB, ta, tb, bias=dbias, bias_div=25, epi=ops.EPI_RELU)
            acc = torch.full((M, N), 0.5).cuda()
            ops.gemm(dA, dB, ta, tb, out=acc, accumulate=True)
        finally:
            lib().repo_debug_bgemm(prev)
        assert relerr(got, F.relu(pre)) < TOL
        assert relerr(acc, opA @ opB + 0.5) < TOL
        errs[engine] = float(((acc.double().cpu() - (opA @ opB + 0.5)).abs() / mag).max())
    log(f"bgemm ta={ta} tb={tb} {M}x{N}x{K}: max |err| / sum|a||b|  bf16x6 {errs[1]:.2e}  fp32 MFMA {errs[0]:.2e}")
    assert errs[1] <= 1.25 * errs[0] + 1e-9, errs


def test_bgemm_weight_gradient_without_slabs(ops):
    """repo_gemm_wgrad without a bias column at the decoder conv1's size takes the bf16x6 engine (one product, no
    split-K slabs): value, accumulate into a strided destination, and bit-reproducibility."""
    rs = np.random.RandomState(9)
    M, N, K = 2450, 1024, 3200
    dY, X = rnd(rs, M, N), rnd(rs, M, K)
    want = dY.double().t() @ X.double()
    dW, _ = ops.gemm_wgrad(dev(dY), dev(X), want_bias=False)
    assert relerr(dW, want) < TOL
    wide = torch.full((N, K + 8), 2.0).cuda()
    ops.gemm_wgrad(dev(dY), dev(X), dW=wide[:, 4 : 4 + K], db=None, accumulate=True, want_bias=False)
    assert relerr(wide[:, 4 : 4 + K], want + 2) < TOL
    assert float((wide[:, :4] - 2).abs().max()) == 0 and float((wide[:, 4 + K :] - 2).abs().max()) == 0
    again, _ = ops.gemm_wgrad(dev(dY), dev(X), want_bias=False)
    assert torch.equal(again, dW)


@pytest.mark.parametrize("M,N,K", [(2450, 200, 230), (50, 60, 200), (3000, 200, 1024), (7, 12, 200), (343, 1, 200)])
def test_gemm_wgrad(ops, M, N, K):
    rs = np.random.RandomState(M + N + K)
    dY, X = rnd(rs, M, N), rnd(rs, M, K)
    dW, db = ops.gemm_wgrad(dev(dY), dev(X))
    e1 = relerr(dW, dY.double().t() @ X.double())
    e2 = relerr(db, dY.double().sum(0))
    log(f"gemm_wgrad {M}x{N}x{K}: dW {e1:.2e} db {e2:.2e}")
    assert e1 < TOL and e2 < TOL
    # accumulate + strided destination (dW is a column block of a wider gradient)
    wide = torch.full((N, K + 50), 2.0).cuda()
    dbacc = torch.full((N,), 3.0).cuda()
    ops.gemm_wgrad(dev(dY), dev(X), dW=wide[:, 20 : 20 + K], db=dbacc, accumulate=True)
    assert relerr(wide[:, 20 : 20 + K], dY.double().t() @ X.double() + 2) < TOL
    assert relerr(dbacc, dY.double().sum(0) + 3) < TOL
    assert float((wide[:, :20] - 2).abs().max()) == 0


def test_gemm_nt_transposition_is_gated_like_the_engine(ops, monkeypatch):
    """ops.gemm / ops.gemm_wgrad make their transposing copy (the bf16x6 engine's NT form) only for products that engine
    takes and from operands repo_transpose accepts: N = 513 (ldb % 4 != 0), a misaligned view and a data-parallel
    shard's 637 rows (125 tiles < 150) stay on the untransposed forms -- and all of them still compute the product."""
    calls = []
    real = ops.transpose
    monkeypatch.setattr(ops, "transpose", lambda *a, **k: (calls.append(a[0].shape), real(*a, **k))[1])
    rs = np.random.RandomState(11)

    def product(M, N, K, view=False):
        A, B = rnd(rs, M, K), rnd(rs, K, N + (1 if view else 0))
        Bd = dev(B)[:, 1:] if view else dev(B)      # view: same ld, data pointer 4 bytes off a 16-byte boundary
        Bh = B[:, 1:] if view else B
        got = ops.gemm(dev(A), Bd)
        assert relerr(got, A.double() @ Bh.double()) < TOL, (M, N, K, view)

    n0 = len(calls)
    product(2450, 3200, 1024)                  # the decoder's first layer: transposed
    assert len(calls) == n0 + 1
    for M, N, K, view in [(640, 513, 512, False), (2560, 1027, 512, True), (637, 3200, 1024, False), (600, 600, 100, False)]:
        n0 = len(calls)
        product(M, N, K, view)
        assert len(calls) == n0, (M, N, K, view, calls[n0:])
    # the switched-off engine: no copy either
    prev = ops.lib().repo_debug_bgemm(0)
    try:
        n0 = len(calls)
        product(2450, 3200, 1024)
        assert len(calls) == n0
    finally:
        ops.lib().repo_debug_bgemm(prev)
    # weight gradient without a bias column: both operands transposed, or none
    for M, N, K, want in [(2450, 1024, 3200, 2), (637, 1024, 3200, 2), (2450, 513, 640, 0)]:
        dY, X = rnd(rs, M, N), rnd(rs, M, K)
        n0 = len(calls)
        dW, _ = ops.gemm_wgrad(dev(dY), dev(X), want_bias=False)
        assert relerr(dW, dY.double().t() @ X.double()) < TOL, (M, N, K)
        assert len(calls) - n0 == want, (M, N, K, calls[n0:])


def test_debug_switches_follow_an_autograd_function_into_its_backward_thread(ops):
    """The repo_debug_* switches are thread-local; torch.autograd runs a Function's backward on its device thread.  The nn
    wrappers record the forward thread's settings and re-apply them around their backward (ops.debug_scope)."""
    import threading

    from repo_amd import functional as Fn
    from repo_amd.algorithms.repo.models.encoder import VisualEncoder

    torch.manual_seed(0)
    enc = VisualEncoder(1024).cuda()
    seen = {}
    real = Fn.encoder_bwd

    def spy(*a, **k):
        seen["thread"], seen["state"] = threading.get_ident(), ops.debug_snapshot()
        return real(*a, **k)

    prev = (ops.lib().repo_debug_scan_spin_limit(12345), ops.lib().repo_debug_bconv(0))
    try:
        Fn.encoder_bwd = spy
        want = ops.debug_snapshot()
        assert want[0] == 12345 and want[2] == 0
        out = enc(torch.rand(3, 3, 64, 64, device="cuda") - 0.5)
        out.sum().backward()
        torch.cuda.synchronize()
    finally:
        Fn.encoder_bwd = real
        ops.lib().repo_debug_scan_spin_limit(prev[0])
        ops.lib().repo_debug_bconv(prev[1])
    assert seen["state"] == want, (seen, want)
    assert ops.debug_snapshot()[0] == prev[0]
    if seen["thread"] != threading.get_ident():      # the backward thread is back on ITS defaults afterwards
        assert all(torch.isfinite(p.grad).all() for p in enc.parameters())


def _layer_tensors(ops, layer, nimg, seed):
    rs = np.random.RandomState(seed)
    (cb, hb, _), (cs, hs, _) = ops.conv_shapes(layer)
    ks = ops.CONV_GEO[layer][3]
    big = rnd(rs, nimg, cb, hb, hb)
    small = rnd(rs, nimg, cs, hs, hs)
    w = rnd(rs, cs, cb, ks, ks, scale=0.1)
    return big, small, w, rs


ALL_LAYERS = list(range(14))  # 0..6: the reference's 64 x 64 stack; 7..12: the build-defined 128 x 128 stack; 13: TIA conv4


@pytest.mark.parametrize("layer", ALL_LAYERS)
@pytest.mark.parametrize("nimg", [1, 5, 37])  # 37: pixel tiles that straddle several images / ragged last tile
def test_conv_down(ops, layer, nimg):
    if layer >= 7 and nimg == 37:
        nimg = 11
    big, small, w, rs = _layer_tensors(ops, layer, nimg, 100 + layer)
    cs = small.shape[1]
    bias = rnd(rs, cs)
    want = F.conv2d(big.double(), w.double(), bias.double(), stride=2)
    got = ops.conv_down(layer, dev(big), dev(w), dev(bias), epi=ops.EPI_RELU)
    e = relerr(got, F.relu(want))
    log(f"conv_down layer {layer} n={nimg}: {e:.2e}")
    assert e < TOL
    # backward-data form: no bias, masked by a saved activation
    h = F.relu(rnd(rs, *small.shape))
    got = ops.conv_down(layer, dev(big), dev(w), None, epi=ops.EPI_MUL_DRELU, aux=dev(h))
    assert relerr(got, F.conv2d(big.double(), w.double(), None, stride=2) * (h > 0)) < TOL


@pytest.mark.parametrize("nimg", [37, 600])   # 600: workgroups that take two or three images each (persistent grid), ragged
def test_conv_down_decoder_conv3_staging_and_multiplying_waves(ops, nimg):
    """Decoder conv3's data gradient from 32 images up runs on csrc/tconv_down.h (one persistent workgroup per CU: four waves
    stage image b + 1's channel chunks and the weight ring while four multiply image b): against fp64, without and with the
    ReLU mask of the saved activation, and the images must not leak into each other (an image's result is the same alone)."""
    layer = 5
    big, small, w, rs = _layer_tensors(ops, layer, nimg, 4100 + nimg)
    big[::7] *= 20.0
    want = F.conv2d(big.double(), w.double(), None, stride=2)
    got = ops.conv_down(layer, dev(big), dev(w), None, epi=ops.EPI_NONE)
    assert relerr(got, want) < TOL
    h = F.relu(rnd(rs, *small.shape))
    got_m = ops.conv_down(layer, dev(big), dev(w), None, epi=ops.EPI_MUL_DRELU, aux=dev(h))
    assert relerr(got_m, want * (h > 0)) < TOL
    assert torch.equal(got_m.cpu(), torch.where(h > 0, got.cpu(), torch.zeros(())))   # the mask selects, bit for bit
    sub = [0, 1, nimg // 2, nimg - 1] + list(range(3, 3 + 32))    # >= 32 images: the same kernel, other workgroup <-> image map
    got_s = ops.conv_down(layer, dev(big[sub]), dev(w), None, epi=ops.EPI_NONE)
    assert torch.equal(got_s.cpu(), got.cpu()[sub])


def test_conv_down_u8(ops):
    rs = np.random.RandomState(3)
    obs = torch.from_numpy(rs.randint(0, 256, size=(6, 3, 64, 64)).astype(np.uint8))
    w, b = rnd(rs, 32, 3, 4, 4, scale=0.1), rnd(rs, 32)
    x = ((obs.numpy().astype(np.float32) / 255) * 2) - 1.0
    want = F.relu(F.conv2d(torch.from_numpy(x).double(), w.double(), b.double(), stride=2))
    got = ops.conv_down(ops.ENC1, dev(obs), dev(w), dev(b), epi=ops.EPI_RELU)
    assert relerr(got, want) < TOL


@pytest.mark.parametrize("layer", ALL_LAYERS)
@pytest.mark.parametrize("nimg", [1, 5, 37])
def test_conv_up(ops, layer, nimg):
    if layer >= 7 and nimg == 37:
        nimg = 11
    big, small, w, rs = _layer_tensors(ops, layer, nimg, 200 + layer)
    cb = big.shape[1]
    bias = rnd(rs, cb)
    hb = big.shape[2]

    def up(bias_):
        # encoder conv2 (31 -> 14) never reads the last input row/col, so its backward-data
        # (the "up" form) leaves them at zero: pad torch's 30x30 result to the 31x31 buffer
        r = F.conv_transpose2d(small.double(), w.double(), None, stride=2)
        r = F.pad(r, (0, hb - r.shape[3], 0, hb - r.shape[2]))
        return r if bias_ is None else r + bias_.double().view(1, -1, 1, 1)

    want = up(bias)
    got = ops.conv_up(layer, dev(small), dev(w), dev(bias), epi=ops.EPI_RELU)
    e = relerr(got, F.relu(want))
    log(f"conv_up layer {layer} n={nimg}: {e:.2e}")
    assert e < TOL
    h = F.relu(rnd(rs, *big.shape))
    got = ops.conv_up(layer, dev(small), dev(w), None, epi=ops.EPI_MUL_DRELU, aux=dev(h))
    assert relerr(got, up(None) * (h > 0)) < TOL
    got = ops.conv_up(layer, dev(small), dev(w), dev(bias))
    assert relerr(got, want) < TOL


@pytest.mark.parametrize("nimg", [4, 600])
def test_conv_up_gather_form_many_images_per_workgroup(ops, nimg):
    """Encoder conv2's data gradient in gather form (csrc/tconv_up.h) where a workgroup walks SEVERAL images (600 images: three
    per workgroup, the accumulators and the weight / relu-operand prefetch carried across images, the last workgroup ragged)
    and at its smallest batch: all three of its epilogues against fp64, the mask and activation forms bit-identical."""
    import repo_amd.ops as rops
    layer = 1
    big, small, w, rs = _layer_tensors(ops, layer, nimg, 321)
    hb = big.shape[2]
    r = F.conv_transpose2d(small.double(), w.double(), None, stride=2)
    want = F.pad(r, (0, hb - r.shape[3], 0, hb - r.shape[2]))
    bias = rnd(rs, big.shape[1])
    got = ops.conv_up(layer, dev(small), dev(w), dev(bias))
    assert relerr(got, want + bias.double().view(1, -1, 1, 1)) < TOL
    h = F.relu(rnd(rs, *big.shape))
    a = ops.conv_up(layer, dev(small), dev(w), None, epi=ops.EPI_MUL_DRELU, aux=dev(h))
    assert relerr(a, want * (h > 0)) < TOL
    bits = (h > 0).to(torch.uint8).view(nimg, -1, 4, hb * hb)   # channel-quad mask: byte (image, quad, pixel), bit = channel & 3
    cmask = (bits[:, :, 0] | (bits[:, :, 1] << 1) | (bits[:, :, 2] << 2) | (bits[:, :, 3] << 3)).contiguous().view(-1)
    m = ops.conv_up(layer, dev(small), dev(w), None, epi=rops.EPI_MUL_CMASK, aux=dev(cmask))
    assert torch.equal(a, m)


@pytest.mark.parametrize("layer,kind", [(5, "down"), (1, "down"), (2, "down"), (3, "down"), (4, "down"), (5, "up"), (1, "up"), (2, "up"),
                                        (3, "up"), (4, "up"), (5, "wgrad"), (2, "wgrad"), (1, "wgrad")])
def test_bf16x6_conv_kernels_match_fp64_and_the_fp32_kernels(ops, layer, kind):
    """The bf16x6 conv kernels (csrc/bconv.h: decoder conv3 / conv2 data gradients, encoder conv2 / conv3 / conv4 forward;
    csrc/buconv.h: decoder conv3 / conv2 forward, encoder conv2 / conv3 / conv4 data gradients; csrc/bwgrad.h: decoder conv3
    and encoder conv3 weight gradients) against fp64 at a batch that fills several
    pixel tiles and straddles images -- and against the fp32-MFMA kernel of the same layer on the SAME operands
    (repo_debug_bconv(0)): the error relative to sum |a||b| may not exceed that kernel's by more than 25 % (measured: at or
    below it), i.e. the six-product split is an fp32-accurate way of feeding the bf16 pipe, not a reduced precision."""
    from repo_amd._lib import lib

    nimg = 41
    big, small, w, rs = _layer_tensors(ops, layer, nimg, 900 + layer)
    big[::3] *= 25.0
    small[::3] *= 25.0
    if kind == "down":
        want = F.conv2d(big.double(), w.double(), None, stride=2)
        mag = F.conv2d(big.double().abs(), w.double().abs(), None, stride=2)
        run = lambda: ops.conv_down(layer, dev(big), dev(w), None, epi=ops.EPI_NONE)  # noqa: E731
    elif kind == "wgrad":
        def corr(a, b):   # dw[cs][cb][ky][kx] = sum_n corr(small, big)
            ks = w.shape[-1]
            cols = F.unfold(b, ks, stride=2).view(b.shape[0], b.shape[1], ks * ks, -1)       # (n, cb, kk, pix)
            return torch.einsum("nsp,nckp->sck", a.flatten(2), cols).view(a.shape[1], b.shape[1], ks, ks)

        want = corr(small.double(), big.double())
        mag = corr(small.double().abs(), big.double().abs())
        run = lambda: ops.conv_wgrad(layer, dev(small), dev(big), want_bias=False)[0]  # noqa: E731
    else:
        hb = big.shape[2]
        pad = lambda r: F.pad(r, (0, hb - r.shape[3], 0, hb - r.shape[2]))  # noqa: E731
        want = pad(F.conv_transpose2d(small.double(), w.double(), None, stride=2))
        mag = pad(F.conv_transpose2d(small.double().abs(), w.double().abs(), None, stride=2)) + 1e-30
        run = lambda: ops.conv_up(layer, dev(small), dev(w), None, epi=ops.EPI_NONE)  # noqa: E731
    errs = {}
    for engine in (1, 0):
        prev = lib().repo_debug_bconv(engine)
        try:
            got = run()
        finally:
            lib().repo_debug_bconv(prev)
        assert relerr(got, want) < TOL
        errs[engine] = float(((got.double().cpu() - want).abs() / (mag + 1e-30)).max())
    log(f"bf16x6 conv layer {layer} {kind}: max |err| / sum|a||b|  bf16x6 {errs[1]:.2e}  fp32 MFMA {errs[0]:.2e}")
    assert errs[1] <= 1.25 * errs[0] + 1e-9, errs


@pytest.mark.parametrize("layer", ALL_LAYERS)
@pytest.mark.parametrize("nimg", [1, 9, 75])  # 75: several image groups per split, ragged last group and split
def test_conv_wgrad(ops, layer, nimg):
    if layer >= 7 and nimg == 75:
        nimg = 21
    big, small, w, rs = _layer_tensors(ops, layer, nimg, 300 + layer)
    bigd = big.double().requires_grad_(False)
    wd = w.double().requires_grad_(True)
    out = F.conv2d(bigd, wd, None, stride=2)
    (out * small.double()).sum().backward()
    dw, db = ops.conv_wgrad(layer, dev(small), dev(big))
    e1 = relerr(dw, wd.grad)
    e2 = relerr(db, small.double().sum((0, 2, 3)))
    log(f"conv_wgrad layer {layer} n={nimg}: dw {e1:.2e} db {e2:.2e}")
    assert e1 < TOL and e2 < TOL
    dw2 = torch.ones_like(dw)
    ops.conv_wgrad(layer, dev(small), dev(big), dw=dw2, db=None, accumulate=True, want_bias=False)
    assert relerr(dw2, wd.grad + 1) < TOL


@pytest.mark.parametrize("layer", [1, 2, 4, 5, 6])
@pytest.mark.parametrize("nimg", [3, 300])
def test_conv_wgrad_channel_sums_of_big(ops, layer, nimg):
    """repo_conv_wgrad's dbias_big: the channel sums of `big` (the bias gradient of a transposed conv) -- fused into the
    weight-gradient kernel where it stages every element once (decoder conv3: 300 images are 3 per workgroup pair, ragged),
    the channel-sum pass elsewhere; the same numbers from both engines, with and without accumulation."""
    from repo_amd._lib import lib

    big, small, w, rs = _layer_tensors(ops, layer, nimg, 700 + layer)
    want = big.double().sum((0, 2, 3))
    for engine in (1, 0):
        prev = lib().repo_debug_bconv(engine)
        try:
            dbig = torch.full((big.shape[1],), 2.0, device="cuda")
            dw, _ = ops.conv_wgrad(layer, dev(small), dev(big), want_bias=False, dbig=dbig, accumulate=False)
            assert relerr(dbig, want) < TOL
            dw2 = dw.clone()
            ops.conv_wgrad(layer, dev(small), dev(big), dw=dw2, db=None, want_bias=False, dbig=dbig, accumulate=True)
            assert relerr(dbig, 2 * want) < TOL and relerr(dw2, 2 * dw.double().cpu()) < TOL
        finally:
            lib().repo_debug_bconv(prev)


def test_conv_wgrad_u8(ops):
    rs = np.random.RandomState(4)
    obs = torch.from_numpy(rs.randint(0, 256, size=(7, 3, 64, 64)).astype(np.uint8))
    x = torch.from_numpy(((obs.numpy().astype(np.float32) / 255) * 2) - 1.0).double()
    small = rnd(rs, 7, 32, 31, 31)
    wd = torch.zeros(32, 3, 4, 4, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, wd, None, stride=2) * small.double()).sum().backward()
    dw, db = ops.conv_wgrad(ops.ENC1, dev(small), dev(obs))
    assert relerr(dw, wd.grad) < TOL


@pytest.mark.parametrize("u8,nimg", [(True, 6), (False, 6), (True, 300)])  # 300 images: persistent tile loop
def test_decoder_out_nll(ops, u8, nimg):
    rs = np.random.RandomState(9)
    h3 = F.relu(rnd(rs, nimg, 32, 30, 30))
    w, b = rnd(rs, 32, 3, 6, 6, scale=0.05), rnd(rs, 3)
    obs = torch.from_numpy(rs.randint(0, 256, size=(nimg, 3, 64, 64)).astype(np.uint8))
    tgt = torch.from_numpy(((obs.numpy().astype(np.float32) / 255) * 2) - 1.0)
    recon = F.conv_transpose2d(h3.double(), w.double(), b.double(), stride=2)
    d = recon - tgt.double()
    loss, dpre, rec = ops.decoder_out_nll(dev(h3), dev(w), dev(b), dev(obs if u8 else tgt), 0.25, want_recon=True)
    e0, e1, e2 = relerr(rec, recon), relerr(dpre, d * 0.25), abs(loss.item() - (0.5 * d * d).sum().item()) / (0.5 * d * d).sum().item()
    log(f"decoder_out_nll u8={u8}: recon {e0:.2e} dpre {e1:.2e} loss {e2:.2e}")
    assert e0 < TOL and e1 < TOL and e2 < 1e-5
    # the quad mask of h3 (REPO_EPI_MUL_MASK4): bit (o & 3) of byte (o >> 2) is h3.flat[o] > 0, every byte written
    sentinel = torch.full((h3.numel() // 4,), 0xAA, dtype=torch.uint8)
    import repo_amd.ops as rops
    loss2, dpre2, _, mask = ops.decoder_out_nll(dev(h3), dev(w), dev(b), dev(obs if u8 else tgt), 0.25, want_mask=True)
    assert mask.dtype == torch.uint8 and mask.numel() == h3.numel() // 4
    bits = (h3.reshape(-1, 4) > 0).to(torch.uint8)
    want = bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3)
    assert torch.equal(mask.cpu(), want)
    assert torch.equal(dpre2, dpre) and loss2.item() == loss.item()
    # dbias: the channel sums of dpre (the output layer's bias gradient), both engines, overwrite and accumulate
    from repo_amd._lib import lib
    for engine in (1, 0):
        prev = lib().repo_debug_bconv(engine)
        try:
            db = torch.full((3,), 5.0, device="cuda")
            ops.decoder_out_nll(dev(h3), dev(w), dev(b), dev(obs if u8 else tgt), 0.25, dbias=db)
            assert relerr(db, (d * 0.25).sum((0, 2, 3))) < 2e-5
            ops.decoder_out_nll(dev(h3), dev(w), dev(b), dev(obs if u8 else tgt), 0.25, dbias=db, accumulate_dbias=True)
            assert relerr(db, 2 * (d * 0.25).sum((0, 2, 3))) < 2e-5
        finally:
            lib().repo_debug_bconv(prev)
    # ... and the data gradient that reads it equals the one that reads h3 itself, bit for bit
    d4 = dev(rnd(rs, nimg, 3, 64, 64))
    a = ops.conv_down(rops.DEC4, d4, dev(w), None, epi=rops.EPI_MUL_DRELU, aux=dev(h3))
    m = ops.conv_down(rops.DEC4, d4, dev(w), None, epi=rops.EPI_MUL_MASK4, aux=mask)
    assert torch.equal(a, m)
    assert float((a == 0).float().mean()) > 0.3  # the mask really masks
    del sentinel


@pytest.mark.parametrize("layer", ALL_LAYERS)
def test_conv_down_channel_sums(ops, layer):
    """repo_conv_down's dbias: the per-channel sum of the values it writes (the bias gradient of the transposed-conv
    layer below, taken in the epilogue) == a second pass over the output; throughput and latency tiles (<= 512
    output pixels), ragged last tiles, accumulation, run-to-run bit-identical."""
    import repo_amd.ops as rops
    rs = np.random.RandomState(70 + layer)
    (cb, hb, _), (cs, hs, _) = rops.conv_shapes(layer)
    ks = rops.CONV_GEO[layer][3]
    for nimg in (1, 9):
        big = dev(rnd(rs, nimg, cb, hb, hb))
        w = dev(rnd(rs, cs, cb, ks, ks, scale=0.1))
        h = dev(F.relu(rnd(rs, nimg, cs, hs, hs)))
        db = torch.full((cs,), 3.0, device="cuda")
        out = ops.conv_down(layer, big, w, None, epi=rops.EPI_MUL_DRELU, aux=h, dbias=db, accumulate_dbias=True)
        want = out.double().sum((0, 2, 3)) + 3.0
        assert relerr(db, want) < TOL, (layer, nimg, relerr(db, want))
        db2 = torch.empty(cs, device="cuda")
        out2 = ops.conv_down(layer, big, w, None, epi=rops.EPI_MUL_DRELU, aux=h, dbias=db2)
        db3 = torch.empty(cs, device="cuda")
        ops.conv_down(layer, big, w, None, epi=rops.EPI_MUL_DRELU, aux=h, dbias=db3)
        assert torch.equal(out2, out) and torch.equal(db2, db3)
        assert relerr(db2, out.double().sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("layer", [1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12])
def test_conv_down_quad_mask_any_geometry(ops, layer):
    """REPO_EPI_MUL_MASK4 on every geometry (pixel planes of 196, 36, 4, 25, 169 and 900 elements: quads that are
    unaligned in the mask's bytes and quads that run over an image's end) equals REPO_EPI_MUL_DRELU bit for bit."""
    import repo_amd.ops as rops
    rs = np.random.RandomState(40 + layer)
    (cb, hb, _), (cs, hs, _) = rops.conv_shapes(layer)
    ks = rops.CONV_GEO[layer][3]
    for nimg in (1, 7):
        big = dev(rnd(rs, nimg, cb, hb, hb))
        w = dev(rnd(rs, cs, cb, ks, ks, scale=0.1))
        h = F.relu(rnd(rs, nimg, cs, hs, hs))
        flat = h.reshape(-1)
        pad = (-flat.numel()) % 4
        bits = (torch.cat([flat, torch.zeros(pad)]).reshape(-1, 4) > 0).to(torch.uint8)
        mask = (bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3)).cuda()
        a = ops.conv_down(layer, big, w, None, epi=rops.EPI_MUL_DRELU, aux=dev(h))
        m = ops.conv_down(layer, big, w, None, epi=rops.EPI_MUL_MASK4, aux=mask)
        assert torch.equal(a, m), (layer, nimg)


@pytest.mark.parametrize("layer", [0, 1, 2])
@pytest.mark.parametrize("engine", [1, 0], ids=["bf16x6", "fp32"])
def test_channel_quad_mask_written_by_conv_down_read_by_conv_up(ops, layer, engine):
    """REPO_EPI_MUL_CMASK: the encoder layer's forward writes the channel-quad mask of its ReLU from the accumulators
    (repo_conv_down relu_cmask: byte ((n * C/4 + c/4) * P + p), bit c % 4 <=> relu > 0), the data gradient of the
    layer above (repo_conv_up on the scatter kernels: enc2 / enc3 / enc4, 961- / 196- / 36-pixel planes, both
    drains) multiplies by it: the mask equals the activation's signs and the gradient equals REPO_EPI_MUL_DRELU's bit
    for bit; ragged image counts; a 3-channel layer has no channel quads (REPO_E_BADARG)."""
    import repo_amd.ops as rops
    from repo_amd._lib import RepoHipError, lib

    rs = np.random.RandomState(90 + layer)
    (cb, hb, _), (cs, hs, _) = rops.conv_shapes(layer)
    ks = rops.CONV_GEO[layer][3]
    up = layer + 1   # the layer whose data gradient lands on this layer's output
    (ucb, uhb, _), (ucs, uhs, _) = rops.conv_shapes(up)
    assert (ucb, uhb) == (cs, hs)
    uks = rops.CONV_GEO[up][3]
    prev = lib().repo_debug_bconv(engine)
    try:
        for nimg in (1, 5, 37):
            big = dev(rnd(rs, nimg, cb, hb, hb))
            w = dev(rnd(rs, cs, cb, ks, ks, scale=0.2))
            b = dev(rnd(rs, cs, scale=0.1))
            h, cmask = ops.conv_down(layer, big, w, b, epi=rops.EPI_RELU, want_cmask=True)
            assert torch.equal(h, ops.conv_down(layer, big, w, b, epi=rops.EPI_RELU))
            bits = (h > 0).to(torch.uint8).view(nimg, cs // 4, 4, hs * hs)
            want = bits[:, :, 0] | (bits[:, :, 1] << 1) | (bits[:, :, 2] << 2) | (bits[:, :, 3] << 3)
            assert torch.equal(cmask.view(nimg, cs // 4, hs * hs), want), (layer, nimg)
            assert 0.2 < float((h > 0).float().mean()) < 0.8
            d = dev(rnd(rs, nimg, ucs, uhs, uhs))
            uw = dev(rnd(rs, ucs, ucb, uks, uks, scale=0.1))
            a = ops.conv_up(up, d, uw, None, epi=rops.EPI_MUL_DRELU, aux=h)
            m = ops.conv_up(up, d, uw, None, epi=rops.EPI_MUL_CMASK, aux=cmask)
            assert torch.equal(a, m), (layer, nimg)
    finally:
        lib().repo_debug_bconv(prev)
    if layer == 0:
        with pytest.raises(RepoHipError):   # the 3-channel data gradient runs on the gather engine: no channel quads
            ops.conv_up(0, dev(rnd(rs, 1, 32, 31, 31)), dev(rnd(rs, 32, 3, 4, 4)), None, epi=rops.EPI_MUL_CMASK,
                        aux=torch.zeros(64, dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("layer,u8", [(6, True), (6, False), (12, True), (12, False)])
def test_conv_up_nll(ops, layer, u8):
    """The gather engine's fused output layer + pixel NLL (layer 12: the 128 x 128 stack's; layer 6: a second
    implementation of repo_decoder_out_nll) vs fp64 torch, and at layer 6 vs the specialised kernel."""
    import repo_amd.ops as rops
    rs = np.random.RandomState(60 + layer)
    (cb, hb, _), (cs, hs, _) = rops.conv_shapes(layer)
    ks = rops.CONV_GEO[layer][3]
    nimg = 5
    h = F.relu(rnd(rs, nimg, cs, hs, hs))
    w, b = rnd(rs, cs, cb, ks, ks, scale=0.05), rnd(rs, cb)
    obs = torch.from_numpy(rs.randint(0, 256, size=(nimg, cb, hb, hb)).astype(np.uint8))
    tgt = torch.from_numpy(((obs.numpy().astype(np.float32) / 255) * 2) - 1.0)
    recon = F.conv_transpose2d(h.double(), w.double(), b.double(), stride=2)
    d = recon - tgt.double()
    loss, dpre, rec = ops.conv_up_nll(layer, dev(h), dev(w), dev(b), dev(obs if u8 else tgt), 0.25, want_recon=True)
    want_loss = (0.5 * d * d).sum().item()
    assert relerr(rec, recon) < TOL and relerr(dpre, d * 0.25) < TOL and abs(loss.item() - want_loss) / want_loss < 1e-5
    if layer == 6:
        loss2, dpre2, rec2 = ops.decoder_out_nll(dev(h), dev(w), dev(b), dev(obs if u8 else tgt), 0.25, want_recon=True)
        assert relerr(rec2, rec.double()) < TOL and relerr(dpre2, dpre.double()) < TOL


def test_channel_sum_relu_mask(ops):
    rs = np.random.RandomState(1)
    x = rnd(rs, 37, 3, 64, 64)
    assert relerr(ops.channel_sum(dev(x)), x.double().sum((0, 2, 3))) < TOL
    x = rnd(rs, 50, 128, 25)
    acc = torch.ones(128).cuda()
    ops.channel_sum(dev(x), out=acc, accumulate=True)
    assert relerr(acc, x.double().sum((0, 2)) + 1) < TOL
    dy, h = rnd(rs, 1000, 33), F.relu(rnd(rs, 1000, 33))
    assert float((ops.relu_mask(dev(dy), dev(h)).cpu() - dy * (h > 0)).abs().max()) == 0


def test_no_uninitialised_lds(ops):
    """Every direct-conv / GEMM kernel must give the same bits after the LDS of all CUs has been filled
    with NaN patterns: a kernel that reads LDS it did not write (even to multiply it by zero) fails here
    deterministically instead of once in a dozen runs."""
    from repo_amd._lib import lib

    def poison():
        assert lib().repo_debug_poison_lds(torch.cuda.current_stream().cuda_stream) == 0

    rs = np.random.RandomState(77)
    for layer in range(7):
        for nimg in (3, 37):
            big, small, w, _ = _layer_tensors(ops, layer, nimg, 900 + layer)
            big, small, w = dev(big), dev(small), dev(w)
            outs = []
            for p in (False, True):
                if p:
                    poison()
                d = ops.conv_down(layer, big, w)
                if p:
                    poison()
                u = ops.conv_up(layer, small, w)
                if p:
                    poison()
                dw, db = ops.conv_wgrad(layer, small, big)
                outs.append((d, u, dw, db))
            for a, b in zip(*outs):
                assert torch.isfinite(b).all(), (layer, nimg)
                assert torch.equal(a, b), (layer, nimg)
    h3, w4, b4 = dev(rnd(rs, 9, 32, 30, 30)), dev(rnd(rs, 32, 3, 6, 6, scale=0.05)), dev(rnd(rs, 3))
    tgt = dev(torch.from_numpy(rs.randint(0, 256, size=(9, 3, 64, 64)).astype(np.uint8)))
    ref = ops.decoder_out_nll(h3, w4, b4, tgt, 0.5)
    poison()
    got = ops.decoder_out_nll(h3, w4, b4, tgt, 0.5)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    for (M, N, K) in ((333, 200, 230), (70, 1024, 36), (500, 7, 1)):
        A, B = dev(rnd(rs, M, K)), dev(rnd(rs, K, N))
        ref = ops.gemm(A, B)
        poison()
        assert torch.equal(ref, ops.gemm(A, B))
        dy = dev(rnd(rs, M, N))
        r1 = ops.gemm_wgrad(dy, A)
        poison()
        r2 = ops.gemm_wgrad(dy, A)
        assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])


def test_no_uninitialised_lds_row_tile_kernels(ops):
    """The persistent row-tile kernels (rollout forward / reverse on both engines, the fused heads) after the LDS of every
    CU has been filled with NaN patterns: same bits as before -- their tiles' padding columns meet zero weights, so they
    must hold finite values the kernels wrote themselves."""
    from repo_amd._lib import lib
    from oracle import fixtures as fx

    def poison():
        assert lib().repo_debug_poison_lds(torch.cuda.current_stream().cuda_stream) == 0

    rs = np.random.RandomState(78)
    Hm, N, A, D, S = 3, 75, 6, 200, 30
    P = fx.make_params(A, 7)
    rp = [torch.tensor(v).cuda() for v in P["transition_model"].values()]
    ap = [torch.tensor(v).cuda() for v in P["actor_model"].values()]
    vp = [torch.tensor(v).cuda() for v in P["value_model"].values()]
    b0, s0 = dev(rnd(rs, N, D, scale=0.3)), dev(rnd(rs, N, S))
    ea, ep = dev(rnd(rs, Hm, N, A)), dev(rnd(rs, Hm, N, S))
    dfeat = dev(rnd(rs, Hm, N, D + S, scale=0.01))
    for engine in (1, 0):
        prev = lib().repo_debug_rowtile32(engine)
        try:
            outs = []
            for p in (False, True):
                if p:
                    poison()
                sv = ops.rssm_imagine_fwd(rp, ap, b0, s0, ea, ep)
                if p:
                    poison()
                d_araw, dfeat0 = ops.rssm_imagine_bwd(rp, sv, dfeat, want_dfeat0=True)
                outs.append((sv.featx.clone(), sv.gates.clone(), sv.a_hidden.clone(), d_araw.clone(), dfeat0.clone()))
            for a, b in zip(*outs):
                assert torch.isfinite(b).all() and torch.equal(a, b), engine
        finally:
            lib().repo_debug_rowtile32(prev)
    x = dev(rnd(rs, 1000, D + S))
    ref, rh = ops.mlp_fwd(vp, x)
    dx_ref = torch.empty_like(x)
    ops.mlp_bwd(vp, x, rh, dev(rnd(np.random.RandomState(1), 1000, 1)), dparams=None, dx=dx_ref)
    poison()
    got, gh = ops.mlp_fwd(vp, x)
    poison()
    dx = torch.empty_like(x)
    ops.mlp_bwd(vp, x, gh, dev(rnd(np.random.RandomState(1), 1000, 1)), dparams=None, dx=dx)
    assert torch.equal(ref, got) and all(torch.equal(a, b) for a, b in zip(rh, gh)) and torch.equal(dx_ref, dx)


def test_device_check_and_arch_guard():
    """repo_device_check: the MI355X passes, an ordinal that does not exist is a bad argument; every entry
    point runs the same (cached) check before launching (REPO_E_ARCH on anything that is not gfx950)."""
    from repo_amd._lib import lib

    L = lib()
    assert L.repo_device_check(0) == 0
    assert L.repo_device_check(torch.cuda.device_count()) == -1  # REPO_E_BADARG
    assert L.repo_strerror(-5).decode() == "device is not gfx950"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName
