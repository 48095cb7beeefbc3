"""Multi-kernel passes of the FiLM-conditioned conv stacks (multitask agents).

Reference: ConditionalVisualEncoder.forward (/root/reference/algorithms/repo/models/encoder.py:78-88) and
ConditionalVisualObservationModel.forward (models/decoder.py:111-123):

    gammas, betas = film(condition).chunk(2, dim=1)        # one Linear(C -> 2 * sum(channels))
    h_l = relu((1 + gamma_l) * conv_l(h_{l-1}) + beta_l)   # per (frame, channel); the decoder's conv4 is not modulated

Round 5: a modulated layer is the conv kernel with the FiLM + ReLU EPILOGUE (REPO_EPI_FILM_RELU: the layer's (frame, channel)
table [1 + gamma | beta] is read once per output quad; h_l is the only tensor written -- the pre-FiLM y_l is never
materialised, two streaming passes per layer less than conv + repo_film_fwd); backwards the data-gradient kernel of layer
l+1 delivers d h_l already masked by ReLU (its MUL_DRELU epilogue on h_l), repo_film_bwd_h turns it into d y_l and the
(frame, channel) gradients of gamma / beta with y_l recovered from h_l where d h_l != 0, and the conv's weight / data
gradients consume d y_l.  REPO_FILM_FUSED=0 restores the two-kernel form (y_l saved, repo_film_fwd / repo_film_bwd).  Parameter lists are the unconditioned stack's (functional.py) followed by
[film.weight, film.bias]; gradients are written in place into `g` (same order).
"""
import os

import torch

from . import ops
from .algorithms.repo.models.conditional import DEC_CHANNELS, ENC_CHANNELS, film_offsets
from .functional import DecHead, _DEC_PAD, _Fork, _dec_compose, dec_head_compose

_ENC_L = (ops.ENC1, ops.ENC2, ops.ENC3, ops.ENC4)
_ENC_OFF = film_offsets(ENC_CHANNELS)
_DEC_OFF = film_offsets(DEC_CHANNELS)


def _film(p_w, p_b, cond):
    """film(condition): (n, C) @ W^T + b -> (n, 2 * channels)."""
    return ops.gemm(cond, p_w, transb=True, bias=p_b)


def _film_grads(dfilm, cond, g_w, g_b, accumulate):
    ops.gemm_wgrad(dfilm, cond, dW=g_w, db=g_b, accumulate=accumulate)


# ----------------------------------------------------------------------------- encoder
def _fused():
    return os.environ.get("REPO_FILM_FUSED", "1") == "1"


def cond_encoder_fwd(p, obs, cond):
    """p = [conv1.w, conv1.b, ..., conv4.w, conv4.b, film.w, film.b]; obs (n,3,64,64) uint8 | float32 in [-1,1];
    cond (n, C).  Returns (embeds (n, 1024), saved)."""
    film = _film(p[8], p[9], cond)
    if _fused():
        tabs = ops.film_tables(film, ENC_CHANNELS)
        x, hs = obs, []
        for l in range(4):
            x = ops.conv_down(_ENC_L[l], x, p[2 * l], p[2 * l + 1], epi=ops.EPI_FILM_RELU, aux=tabs[l])
            hs.append(x)
        return x.view(x.shape[0], -1), (film, None, hs)
    x, ys, hs = obs, [], []
    for l in range(4):
        y = ops.conv_down(_ENC_L[l], x, p[2 * l], p[2 * l + 1], epi=ops.EPI_NONE)
        x = ops.film_fwd(y, film, *_ENC_OFF[l])
        ys.append(y)
        hs.append(x)
    return x.view(x.shape[0], -1), (film, ys, hs)


def cond_encoder_bwd(p, obs, cond, saved, dembeds, g, accumulate=False, side=None):
    """Gradients of the ten encoder tensors into g."""
    film, ys, hs = saved
    fk = _Fork(side)
    dfilm = torch.empty_like(film)
    packs = [None] + [ops.conv_up_pack(_ENC_L[l], p[2 * l]) for l in (1, 2, 3)]
    dh = ops.relu_mask(dembeds.reshape(hs[3].shape).contiguous(), hs[3])
    dys = []   # every d y_l stays alive until fk.join(): the side stream's weight-gradient kernel of layer l still reads
    #            d y_l while this stream moves on, and a block freed here is handed to this stream's NEXT allocation
    #            (the caching allocator orders reuse on the allocating stream only)
    for l in (3, 2, 1, 0):
        below = hs[l - 1] if l > 0 else obs
        dy = (ops.film_bwd_h(dh, hs[l], film, *_ENC_OFF[l], dfilm,
                             exact=(ops.FILM_CONV_DOWN, _ENC_L[l], below, p[2 * l], p[2 * l + 1])) if ys is None
              else ops.film_bwd(dh, ys[l], film, *_ENC_OFF[l], dfilm))
        dys.append(dy)
        fk.run(lambda l=l, dy=dy, below=below: ops.conv_wgrad(_ENC_L[l], dy, below, dw=g[2 * l], db=g[2 * l + 1],
                                                             accumulate=accumulate))
        if l > 0:
            dh = ops.conv_up(_ENC_L[l], dy, p[2 * l], None, epi=ops.EPI_MUL_DRELU, aux=hs[l - 1], pack=packs[l])
    _film_grads(dfilm, cond, g[8], g[9], accumulate)
    fk.join()
    del dys


# ----------------------------------------------------------------------------- decoder
def _cond_decoder_trunk(p, feat, cond, head=None):
    """feat (rows, D + S) = [belief | state]: the pixel decoder concatenates nothing (models/decoder.py:116), the
    condition enters through FiLM on conv1..conv3 only;
    p = [fc1.w, fc1.b, conv1.w, conv1.b, ..., conv4.w, conv4.b, film.w, film.b]."""
    rows = feat.shape[0]
    film = _film(p[10], p[11], cond)
    pk2, pk3 = ops.conv_up_pack(ops.DEC2, p[4]), ops.conv_up_pack(ops.DEC3, p[6])
    w1 = p[2].view(p[2].shape[0], -1)
    if _fused() and _dec_compose(rows):
        # fc1 and conv1 composed (functional.dec_head_compose: linear in sequence, models/decoder.py:113-116): one bias per
        # output element, one FiLM pair per 25 of them (bias_div = -25)
        tabs = ops.film_tables(film, DEC_CHANNELS)
        h0 = head if head is not None else dec_head_compose(p)
        h1 = ops.gemm(feat, h0.w01aug[:, : feat.shape[1]], transb=True, bias=h0.b01, bias_div=-25, epi=ops.EPI_FILM_RELU,
                      aux=tabs[0].view(rows, -1)).view(rows, 128, 5, 5)
        h2 = ops.conv_up(ops.DEC2, h1, p[4], p[5], epi=ops.EPI_FILM_RELU, aux=tabs[1], pack=pk2)
        h3 = ops.conv_up(ops.DEC3, h2, p[6], p[7], epi=ops.EPI_FILM_RELU, aux=tabs[2], pack=pk3)
        return film, h0, None, (h1, h2, h3)
    h0 = ops.gemm(feat, p[0], transb=True, bias=p[1])
    if _fused() and rows > 8:   # (a handful of rows -- the acting path's reconstruction -- takes the vector path: two kernels)
        tabs = ops.film_tables(film, DEC_CHANNELS)
        h1 = ops.gemm(h0, w1, bias=p[3], bias_div=25, epi=ops.EPI_FILM_RELU, aux=tabs[0].view(rows, -1)).view(rows, 128, 5, 5)
        h2 = ops.conv_up(ops.DEC2, h1, p[4], p[5], epi=ops.EPI_FILM_RELU, aux=tabs[1], pack=pk2)
        h3 = ops.conv_up(ops.DEC3, h2, p[6], p[7], epi=ops.EPI_FILM_RELU, aux=tabs[2], pack=pk3)
        return film, h0, None, (h1, h2, h3)
    y1 = ops.gemm(h0, w1, bias=p[3], bias_div=25, epi=ops.EPI_NONE).view(rows, 128, 5, 5)
    h1 = ops.film_fwd(y1, film, *_DEC_OFF[0])
    y2 = ops.conv_up(ops.DEC2, h1, p[4], p[5], epi=ops.EPI_NONE, pack=pk2)
    h2 = ops.film_fwd(y2, film, *_DEC_OFF[1])
    y3 = ops.conv_up(ops.DEC3, h2, p[6], p[7], epi=ops.EPI_NONE, pack=pk3)
    h3 = ops.film_fwd(y3, film, *_DEC_OFF[2])
    return film, h0, (y1, y2, y3), (h1, h2, h3)


def cond_decoder_fwd(p, feat, cond):
    """Values only: feat (rows, D + S), cond (rows, C) -> (recon (rows,3,64,64), saved)."""
    film, h0, ys, hs = _cond_decoder_trunk(p, feat, cond)
    recon = ops.conv_up(ops.DEC4, hs[2], p[8], p[9], epi=ops.EPI_NONE)
    return recon, (film, h0, ys, hs)


def cond_decoder_fwd_nll(p, feat, cond, target, grad_scale, head=None):
    """Decoder forward fused with the unit-variance pixel NLL (dreamer_mt.py:189-195).
    Returns (sum 0.5*(recon-target)^2 (1,), saved incl. d loss / d recon * grad_scale)."""
    film, h0, ys, hs = _cond_decoder_trunk(p, feat, cond, head)
    db4 = torch.empty(3, dtype=torch.float32, device=feat.device)   # the output bias gradient, out of the same kernel
    loss_sum, dpre4, _, mask3 = ops.decoder_out_nll(hs[2], p[8], p[9], target, grad_scale, want_mask=True, dbias=db4)
    return loss_sum, (film, h0, ys, hs, dpre4, mask3, db4)


def cond_decoder_bwd(p, feat, cond, saved, g, dfeat=None, accumulate_dfeat=False, accumulate=False, side=None):
    """From d recon (saved) to the twelve decoder tensors (into g) and, if dfeat (rows, ld >= D+S) is given (Dreamer's
    attached decoder), to the [belief | state] input (written into its first D+S columns)."""
    film, h0, ys, (h1, h2, h3), d4, mask3, db4 = saved
    rows = feat.shape[0]

    def fbwd(dh, l):   # the FiLM backward of decoder layer l (0 .. 2): from y_l if it was saved, else from h_l
        if ys is None:
            # (a plane whose 1 + gamma is nearly 0 recomputes y from the layer itself: ops.film_bwd_h)
            if l == 0 and composed:   # y1 = feat W01^T + b01: dense over K = 230 with one bias per output element
                exact = (ops.FILM_DENSE, (feat.shape[1], True), feat, w01t[: feat.shape[1]], h0.b01)
            elif l == 0:
                exact = (ops.FILM_DENSE, h0.shape[1], h0, p[2].view(p[2].shape[0], -1), p[3])
            else:
                exact = (ops.FILM_CONV_UP, (ops.DEC2, ops.DEC3)[l - 1], (h1, h2)[l - 1], p[2 + 2 * l], p[3 + 2 * l])
            return ops.film_bwd_h(dh, (h1, h2, h3)[l], film, *_DEC_OFF[l], dfilm, exact=exact)
        return ops.film_bwd(dh, ys[l], film, *_DEC_OFF[l], dfilm)

    composed = isinstance(h0, DecHead)
    w01t = ops.transpose(h0.w01aug) if composed else None   # (232, 3200): W01^T, for the exact FiLM pass and d feat
    fk = _Fork(side)
    dfilm = torch.empty_like(film)

    def w4():
        ops.conv_wgrad(ops.DEC4, h3, d4, dw=g[8], db=None, accumulate=accumulate, want_bias=False)
        if accumulate:
            g[9].add_(db4)
        else:
            g[9].copy_(db4)

    fk.run(w4)
    dh3 = ops.conv_down(ops.DEC4, d4, p[8], None, epi=ops.EPI_MUL_MASK4, aux=mask3)
    dy3 = fbwd(dh3, 2)

    def w3():
        ops.conv_wgrad(ops.DEC3, h2, dy3, dw=g[6], db=None, accumulate=accumulate, want_bias=False, dbig=g[7])

    fk.run(w3)
    dh2 = ops.conv_down(ops.DEC3, dy3, p[6], None, epi=ops.EPI_MUL_DRELU, aux=h2)
    dy2 = fbwd(dh2, 1)

    def w2():
        ops.conv_wgrad(ops.DEC2, h1, dy2, dw=g[4], db=None, accumulate=accumulate, want_bias=False)
        ops.channel_sum(dy2, out=g[5], accumulate=accumulate)

    fk.run(w2)
    dh1 = ops.conv_down(ops.DEC2, dy2, p[4], None, epi=ops.EPI_MUL_DRELU, aux=h1)
    dy1 = fbwd(dh1, 0)
    d1f = dy1.view(rows, 128 * 25)
    w1 = p[2].view(p[2].shape[0], -1)
    if composed:   # functional._decoder_bwd_tail's composed branch: every gradient of the pair through (G | s) = d y1^T (feat | 1)
        F_ = feat.shape[1]
        gaug = torch.zeros(d1f.shape[1], _DEC_PAD, dtype=torch.float32, device=d1f.device)
        _, s_ = ops.gemm_wgrad(d1f, feat, dW=gaug[:, :F_])
        gaug[:, F_].copy_(s_)

        def wpair():
            ops.gemm(h0.w0aug, gaug, transb=True, out=g[2].view(w1.shape), accumulate=accumulate)
            ops.channel_sum(dy1.view(rows, 128, 25), out=g[3], accumulate=accumulate)
            d0aug, _ = ops.gemm_wgrad(h0.w1t, gaug, want_bias=False)
            if accumulate:
                g[0].add_(d0aug[:, :F_])
                g[1].add_(d0aug[:, F_])
            else:
                g[0].copy_(d0aug[:, :F_])
                g[1].copy_(d0aug[:, F_])

        fk.run(wpair)
        if dfeat is not None:
            ops.gemm(d1f, w01t[:F_], transb=True, out=dfeat, accumulate=accumulate_dfeat)
        _film_grads(dfilm, cond, g[10], g[11], accumulate)
        fk.join()
        return

    def w1f():
        ops.gemm_wgrad(h0, d1f, dW=g[2].view(w1.shape), db=None, accumulate=accumulate, want_bias=False)
        ops.channel_sum(dy1.view(rows, 128, 25), out=g[3], accumulate=accumulate)

    fk.run(w1f)
    dh0 = ops.gemm(d1f, w1, transb=True)
    ops.gemm_wgrad(dh0, feat, dW=g[0], db=g[1], accumulate=accumulate)
    if dfeat is not None:
        ops.gemm(dh0, p[0], out=dfeat, accumulate=accumulate_dfeat)
    _film_grads(dfilm, cond, g[10], g[11], accumulate)
    fk.join()
