"""Multi-kernel building blocks shared by the nn.Module wrappers and the agents' hand-scheduled
update: conv encoder, transposed-conv decoder (+ fused pixel NLL) and their backward passes,
expressed over repo_amd.ops (the C ABI).  Parameter lists are in the reference modules'
state_dict order; gradients are written in place into caller-provided tensors (views of the
flat gradient buffer), so no autograd graph and no extra accumulation pass are involved.
"""
import os

import torch

from . import ops


# ----------------------------------------------------------------------------- encoder
# layer ids per frame size: the reference's 64 x 64 stack, and the build-defined 128 x 128 one (BASELINE config 4's
# size; same kernel sizes, 256x6x6 flatten + a linear `fc` 9216 -> 1024: p then has ten tensors)
_ENC = {64: (ops.ENC1, ops.ENC2, ops.ENC3, ops.ENC4), 128: (ops.X_ENC1, ops.X_ENC2, ops.X_ENC3, ops.X_ENC4)}


def encoder_fwd(p, obs):
    """VisualEncoder.forward (models/encoder.py:34-41).  obs (n,3,64,64) [or (n,3,128,128)] uint8 or float32.
    p = [conv1.w, conv1.b, ..., conv4.w, conv4.b (, fc.w, fc.b)].  Returns (embeds (n,1024), saved)."""
    L = _ENC[obs.shape[-1]]
    if len(p) > 8:   # the 128 x 128 stack (its data gradients take other engines: no masks)
        h1 = ops.conv_down(L[0], obs, p[0], p[1], epi=ops.EPI_RELU)
        h2 = ops.conv_down(L[1], h1, p[2], p[3], epi=ops.EPI_RELU)
        h3 = ops.conv_down(L[2], h2, p[4], p[5], epi=ops.EPI_RELU)
        h4 = ops.conv_down(L[3], h3, p[6], p[7], epi=ops.EPI_RELU)
        return ops.gemm(h4.view(h4.shape[0], -1), p[8], transb=True, bias=p[9]), (h1, h2, h3, h4)
    # conv2 and conv3 also write the channel-quad mask of their ReLU (1/16 of the activation's bytes): all the data
    # gradient of the layer above needs of it
    # (not h1's: since round 5 conv1 runs on the raw bytes at 100 us and its mask costs 40 us of byte stores, which is
    # what conv2's data gradient saves by reading 19 MB instead of 301: update 6.351 / 6.359 / 6.369 ms without against
    # 6.360 / 6.371 / 6.382 with, alternating on one box; again with conv2's data gradient in gather form, which reads the
    # operand ahead of its use: 6.071 / 6.078 / 6.153 without against 6.070 / 6.081 / 6.118 with)
    h1, m1 = ops.conv_down(L[0], obs, p[0], p[1], epi=ops.EPI_RELU), None
    h2, m2 = ops.conv_down(L[1], h1, p[2], p[3], epi=ops.EPI_RELU, want_cmask=True)
    h3, m3 = ops.conv_down(L[2], h2, p[4], p[5], epi=ops.EPI_RELU, want_cmask=True)
    h4 = ops.conv_down(L[3], h3, p[6], p[7], epi=ops.EPI_RELU)
    return h4.view(h4.shape[0], -1), (h1, h2, h3, h4, m1, m2, m3)


class _Fork:
    """Runs weight-gradient kernels on a side stream while the data-gradient chain (the critical
    path) continues on the current stream.  A weight gradient of layer l and the data gradient
    into layer l-1 both only READ d pre_l, so they are independent; on their own each of these
    kernels leaves the matrix pipe ~40% idle (barrier / load phases), and two resident kernels
    fill each other's gaps."""

    def __init__(self, side, deferred=None):
        self.side = side
        self.main = torch.cuda.current_stream() if side is not None else None
        self.deferred = deferred  # a list: collect the independent work instead of running it

    def run(self, fn):
        if self.deferred is not None:
            self.deferred.append(fn)
            return
        if self.side is None:
            fn()
            return
        self.side.wait_stream(self.main)  # everything fn reads has been enqueued on main
        with torch.cuda.stream(self.side):
            fn()

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)


def encoder_bwd(p, obs, saved, dembeds, g, accumulate=False, side=None):
    """Gradients of all eight encoder tensors into g (same order as p)."""
    h1, h2, h3, h4 = saved[:4]
    masks = saved[4:] if len(saved) > 4 else None
    n = h4.shape[0]
    L = _ENC[obs.shape[-1]]
    fk = _Fork(side)
    if len(p) > 8:  # the 128 x 128 stack's fc: d flatten = d embeds @ W, dW = d embeds^T @ flatten
        flat = h4.view(n, -1)
        dflat = ops.gemm(dembeds, p[8])
        fk.run(lambda: ops.gemm_wgrad(dembeds, flat, dW=g[8], db=g[9], accumulate=accumulate))
        d4 = ops.relu_mask(dflat.view(h4.shape), h4)
    else:
        d4 = ops.relu_mask(dembeds.reshape(h4.shape).contiguous(), h4)
    fk.run(lambda: ops.conv_wgrad(L[3], d4, h3, dw=g[6], db=g[7], accumulate=accumulate))
    # the three weight packs first (independent of the gradients): their launches do not sit between the convs
    pk4, pk3, pk2 = ops.conv_up_pack(L[3], p[6]), ops.conv_up_pack(L[2], p[4]), ops.conv_up_pack(L[1], p[2])
    def up(layer, d, w, h, mask, pack):   # d of the layer below = conv_up(d) * relu'(h), from h or from its mask
        if mask is not None:
            return ops.conv_up(layer, d, w, None, epi=ops.EPI_MUL_CMASK, aux=mask, pack=pack)
        return ops.conv_up(layer, d, w, None, epi=ops.EPI_MUL_DRELU, aux=h, pack=pack)

    m1, m2, m3 = masks if masks else (None, None, None)
    d3 = up(L[3], d4, p[6], h3, m3, pk4)
    fk.run(lambda: ops.conv_wgrad(L[2], d3, h2, dw=g[4], db=g[5], accumulate=accumulate))
    d2 = up(L[2], d3, p[4], h2, m2, pk3)
    fk.run(lambda: ops.conv_wgrad(L[1], d2, h1, dw=g[2], db=g[3], accumulate=accumulate))
    d1 = up(L[1], d2, p[2], h1, m1, pk2)
    ops.conv_wgrad(L[0], d1, obs, dw=g[0], db=g[1], accumulate=accumulate)
    fk.join()


# ----------------------------------------------------------------------------- decoder
# The decoder's first two layers are LINEAR in sequence: `hidden = self.fc1(cat)  # No nonlinearity here`, then
# `act(conv1(hidden.view(-1, E, 1, 1)))` (models/decoder.py:41-44) -- a (230 -> 1024) Linear followed by a
# (1024 -> 128 x 5 x 5) transposed conv on a 1 x 1 input, which is a (1024 -> 3200) Linear.  Their composition is one
# (230 -> 3200) Linear, W01 = W1^T W0 (3200 x 230), b01 = W1^T b0 + b1: the 1024-wide hidden never has to exist, and
# neither do the three largest GEMMs of the update (forward 2450 x 3200 x 1024, input gradient 2450 x 1024 x 3200, weight
# gradient 1024 x 3200 over 2450 rows: 16 GFLOP each).  With G = d1^T feat (3200 x 230) and s = column sums of d1:
#     h1   = relu(feat W01^T + b01)
#     d W1 = W0 G^T + b0 s^T,   d W0 = W1 G,   d b0 = W1 s,   d b1 = channel sums of d1,   d feat = d1 W01
# -- 5 / 7-10 GFLOP forward / backward instead of 17 / 34, exact in real arithmetic (fp32 rounding differs at the 1e-6
# level, as between any two summation orders).  The bias terms ride as a 231st column (W0 | b0), (G | s): the vector
# products b0^T W1 and W1 s alone are one-row GEMMs that cost 90-155 us on the tile engines (measured:
# tools/probe_py/dec_head_compose.py); W1 G is a reduction over 3200 "rows" of W1^T and goes to the row-split weight-gradient
# engine.  From _DEC_COMPOSE_MIN_ROWS rows up (composing costs 1.5 GFLOP whatever the batch: the acting path's single row
# keeps the two layers); REPO_DEC_COMPOSE=0 restores the two-layer form everywhere.
_DEC_COMPOSE_MIN_ROWS = 512
_DEC_PAD = 232   # 230 inputs + the bias column, padded to a multiple of 4


def _dec_compose(rows):
    return rows >= _DEC_COMPOSE_MIN_ROWS and os.environ.get("REPO_DEC_COMPOSE", "1") == "1"


class DecHead:
    """The composed first two decoder layers of one parameter state: w0aug (1024, 232) = [W0 | b0 | 0], w1t (3200, 1024)
    = W1^T, w01aug (3200, 232) = W1^T w0aug = [W01 | W1^T b0 | 0], b01 (3200,)."""
    __slots__ = ("w0aug", "w1t", "w01aug", "b01")


def dec_head_compose(p):
    """Depends on the parameters only: the agents issue it ahead of the scan, off the decoder's chain."""
    w0, b0 = p[0], p[1]
    w1 = p[2].view(p[2].shape[0], -1)                        # (1024, 3200)
    F_ = w0.shape[1]
    dh = DecHead()
    dh.w0aug = torch.zeros(w0.shape[0], _DEC_PAD, dtype=torch.float32, device=w0.device)
    dh.w0aug[:, :F_].copy_(w0)
    dh.w0aug[:, F_].copy_(b0)
    dh.w1t = ops.transpose(w1)                                # (3200, 1024)
    dh.w01aug = ops.gemm(dh.w1t, dh.w0aug)                    # (3200, 232)
    dh.b01 = dh.w01aug[:, F_] + p[3].repeat_interleave(w1.shape[1] // p[3].shape[0])
    return dh


def decoder_trunk_fwd(p, feat, head=None):
    """fc1 + the first three transposed convolutions (models/decoder.py:41-46).
    feat (rows, 230) = [belief|state]; p = [fc1.w, fc1.b, conv1.w, conv1.b, ..., conv4.w, conv4.b].
    Returns (h0, h1, h2, h3): h0 is the DecHead when the first two layers ran composed (see above; `head`: one made
    ahead of time by dec_head_compose)."""
    rows = feat.shape[0]
    pk2, pk3 = ops.conv_up_pack(ops.DEC2, p[4]), ops.conv_up_pack(ops.DEC3, p[6])
    if _dec_compose(rows):
        h0 = head if head is not None else dec_head_compose(p)
        F_ = feat.shape[1]
        h1 = ops.gemm(feat, h0.w01aug[:, :F_], transb=True, bias=h0.b01, epi=ops.EPI_RELU).view(rows, 128, 5, 5)
    else:
        h0 = ops.gemm(feat, p[0], transb=True, bias=p[1])
        w1 = p[2].view(p[2].shape[0], -1)  # (1024, 128*25): 1x1 -> 5x5 transposed conv is a GEMM
        h1 = ops.gemm(h0, w1, bias=p[3], bias_div=25, epi=ops.EPI_RELU).view(rows, 128, 5, 5)
    h2 = ops.conv_up(ops.DEC2, h1, p[4], p[5], epi=ops.EPI_RELU, pack=pk2)
    h3 = ops.conv_up(ops.DEC3, h2, p[6], p[7], epi=ops.EPI_RELU, pack=pk3)
    return h0, h1, h2, h3


def decoder_fwd(p, feat, head=None):
    """VisualObservationModel.forward -> (recon (rows,3,64,64), saved).  Twelve tensors in p = the 128 x 128 stack
    (conv4 32 -> 16 with ReLU, conv5 16 -> 3): recon (rows,3,128,128), saved gains h4."""
    h0, h1, h2, h3 = decoder_trunk_fwd(p, feat, head)
    if len(p) > 10:
        h4 = ops.conv_up(ops.X_DEC4, h3, p[8], p[9], epi=ops.EPI_RELU)
        recon = ops.conv_up(ops.X_DEC5, h4, p[10], p[11], epi=ops.EPI_NONE)
        return recon, (h0, h1, h2, h3, h4)
    # (TIAObservationModel, models/decoder.py:165-175: conv4 has 6 output channels = [recon | mask], layer T_DEC4)
    recon = ops.conv_up(ops.T_DEC4 if p[8].shape[1] == 6 else ops.DEC4, h3, p[8], p[9], epi=ops.EPI_NONE)
    return recon, (h0, h1, h2, h3)


def decoder_fwd_nll(p, feat, target, grad_scale, head=None):
    """Decoder forward fused with the unit-variance pixel NLL (repo.py:46-53).
    Returns (sum 0.5*(recon-target)^2 (1,), saved incl. d loss/d recon * grad_scale)."""
    h0, h1, h2, h3 = decoder_trunk_fwd(p, feat, head)
    if len(p) > 10:  # 128 x 128 stack: the output layer + NLL on the gather engine (ops.conv_up_nll)
        h4 = ops.conv_up(ops.X_DEC4, h3, p[8], p[9], epi=ops.EPI_RELU)
        loss_sum, dpre5, _ = ops.conv_up_nll(ops.X_DEC5, h4, p[10], p[11], target, grad_scale)
        return loss_sum, (h0, h1, h2, h3, h4, dpre5)
    # the output bias gradient (the channel sums of dpre4) comes out of the same kernel
    db4 = torch.empty(3, dtype=torch.float32, device=h3.device)
    loss_sum, dpre4, _, mask3 = ops.decoder_out_nll(h3, p[8], p[9], target, grad_scale, want_mask=True, dbias=db4)
    return loss_sum, (h0, h1, h2, h3, dpre4, mask3, db4)


def decoder_bwd(p, feat, saved, g, dfeat=None, accumulate_dfeat=False, accumulate=False, side=None, deferred=None):
    """Backward from d recon (= saved[4]) to all ten decoder tensors (into g) and, if dfeat is
    given (Dreamer's attached decoder, dreamer.py:262), to the [belief|state] input.
    deferred: a list that receives the weight-gradient launches as closures instead of running them, so
    the caller can issue them beside a later latency-bound kernel (they depend only on tensors kept
    alive by the closures)."""
    rows = feat.shape[0]
    fk = _Fork(side, deferred)
    if len(p) > 10:
        return _decoder_bwd_128(p, feat, saved, g, dfeat, accumulate_dfeat, accumulate, fk)
    h0, h1, h2, h3, d4 = saved[:5]
    mask3 = saved[5] if len(saved) > 5 else None  # quad mask of h3 from the fused output layer (8.8 MB for 282)
    db4 = saved[6] if len(saved) > 6 else None    # and its bias gradient (the channel sums of d4)
    l4 = ops.T_DEC4 if p[8].shape[1] == 6 else ops.DEC4

    def w4():
        ops.conv_wgrad(l4, h3, d4, dw=g[8], db=None, accumulate=accumulate, want_bias=False)
        if db4 is None:
            ops.channel_sum(d4, out=g[9], accumulate=accumulate)
        elif accumulate:
            g[9].add_(db4)
        else:
            g[9].copy_(db4)

    fk.run(w4)
    # The bias gradients of conv3 / conv2 / conv1 (g[7], g[5], g[3]) are the channel sums of d3 / d2 / d1.  conv4's comes
    # out of the fused output layer (its kernel has every d in registers) and conv3's out of its weight-gradient kernel
    # (which stages d3 exactly once, off the critical chain: round 5); conv2's and conv1's stay
    # separate passes on the weight-gradient stream: taking them in the data-gradient kernels' epilogues
    # (ops.conv_down(dbias=), tested) saves 240 us of kernel time per update but puts the shuffles, one more
    # barrier per workgroup and a dependent reduction launch per layer ON the critical chain: 8.65 vs 8.55 ms per
    # update (A/B on one box, round 3).
    if mask3 is not None:
        d3 = ops.conv_down(l4, d4, p[8], None, epi=ops.EPI_MUL_MASK4, aux=mask3)
    else:
        d3 = ops.conv_down(l4, d4, p[8], None, epi=ops.EPI_MUL_DRELU, aux=h3)

    def w3():
        # conv3's bias gradient (the channel sums of d3) rides in the weight-gradient kernel, which stages d3 exactly once
        ops.conv_wgrad(ops.DEC3, h2, d3, dw=g[6], db=None, accumulate=accumulate, want_bias=False, dbig=g[7])

    fk.run(w3)
    _decoder_bwd_tail(p, feat, h0, h1, h2, d3, g, dfeat, accumulate_dfeat, accumulate, fk)


def _decoder_bwd_128(p, feat, saved, g, dfeat, accumulate_dfeat, accumulate, fk):
    """The 128 x 128 stack's two extra layers (conv5 16 -> 3, k2; conv4 32 -> 16, k6), then the shared tail."""
    h0, h1, h2, h3, h4, d5 = saved

    def w5():
        ops.conv_wgrad(ops.X_DEC5, h4, d5, dw=g[10], db=None, accumulate=accumulate, want_bias=False)
        ops.channel_sum(d5, out=g[11], accumulate=accumulate)

    fk.run(w5)
    d4 = ops.conv_down(ops.X_DEC5, d5, p[10], None, epi=ops.EPI_MUL_DRELU, aux=h4)

    def w4():
        ops.conv_wgrad(ops.X_DEC4, h3, d4, dw=g[8], db=None, accumulate=accumulate, want_bias=False)
        ops.channel_sum(d4, out=g[9], accumulate=accumulate)

    fk.run(w4)
    d3 = ops.conv_down(ops.X_DEC4, d4, p[8], None, epi=ops.EPI_MUL_DRELU, aux=h3)

    def w3():
        # conv3's bias gradient (the channel sums of d3) rides in the weight-gradient kernel, which stages d3 exactly once
        ops.conv_wgrad(ops.DEC3, h2, d3, dw=g[6], db=None, accumulate=accumulate, want_bias=False, dbig=g[7])

    fk.run(w3)
    _decoder_bwd_tail(p, feat, h0, h1, h2, d3, g, dfeat, accumulate_dfeat, accumulate, fk)


def _decoder_bwd_tail(p, feat, h0, h1, h2, d3, g, dfeat, accumulate_dfeat, accumulate, fk):
    """From d h3's pre-activation gradient down to fc1 (the same layers at both frame sizes)."""
    rows = feat.shape[0]
    d2 = ops.conv_down(ops.DEC3, d3, p[6], None, epi=ops.EPI_MUL_DRELU, aux=h2)

    def w2():
        ops.conv_wgrad(ops.DEC2, h1, d2, dw=g[4], db=None, accumulate=accumulate, want_bias=False)
        ops.channel_sum(d2, out=g[5], accumulate=accumulate)

    fk.run(w2)
    d1 = ops.conv_down(ops.DEC2, d2, p[4], None, epi=ops.EPI_MUL_DRELU, aux=h1)
    d1f = d1.view(rows, 128 * 25)
    w1 = p[2].view(p[2].shape[0], -1)
    if isinstance(h0, DecHead):
        # the first two layers ran composed (decoder_trunk_fwd): every gradient of the pair through (G | s) = d1^T (feat | 1)
        F_ = feat.shape[1]
        gaug = torch.zeros(d1f.shape[1], _DEC_PAD, dtype=torch.float32, device=d1f.device)     # (3200, 232) = [G | s | 0]
        _, s_ = ops.gemm_wgrad(d1f, feat, dW=gaug[:, :F_])
        gaug[:, F_].copy_(s_)

        def wpair():
            ops.gemm(h0.w0aug, gaug, transb=True, out=g[2].view(w1.shape), accumulate=accumulate)   # d W1 = W0 G^T + b0 s^T
            ops.channel_sum(d1.view(rows, 128, 25), out=g[3], accumulate=accumulate)
            d0aug, _ = ops.gemm_wgrad(h0.w1t, gaug, want_bias=False)                                # (1024, 232) = [W1 G | W1 s | 0]
            if accumulate:
                g[0].add_(d0aug[:, :F_])
                g[1].add_(d0aug[:, F_])
            else:
                g[0].copy_(d0aug[:, :F_])
                g[1].copy_(d0aug[:, F_])

        fk.run(wpair)
        if dfeat is not None:
            # d feat = d1 W01, as an NT product over W01^T (a 3 MB transposing copy: 94 -> ~55 us at 2450 rows)
            w01t = ops.transpose(h0.w01aug)                                                         # (232, 3200)
            ops.gemm(d1f, w01t[:F_], transb=True, out=dfeat, accumulate=accumulate_dfeat)
        fk.join()
        return

    def w1f():
        ops.gemm_wgrad(h0, d1f, dW=g[2].view(w1.shape), db=None, accumulate=accumulate, want_bias=False)
        ops.channel_sum(d1.view(rows, 128, 25), out=g[3], accumulate=accumulate)

    fk.run(w1f)
    dh0 = ops.gemm(d1f, w1, transb=True)
    ops.gemm_wgrad(dh0, feat, dW=g[0], db=g[1], accumulate=accumulate)
    if dfeat is not None:
        ops.gemm(dh0, p[0], out=dfeat, accumulate=accumulate_dfeat)
    fk.join()
