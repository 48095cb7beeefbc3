#!/usr/bin/env python3
"""Every kernel of the update ALONE on an idle GPU, at the update's shapes (B=50, L=50, H=15, A=6):
us per call (HIP events on the launch stream), useful TFLOP/s or GB/s and the fraction of the bound
(fp32 MFMA 157.3 TFLOP/s; HBM 8 TB/s).  Output = profiles/rNN_layers_isolated.txt.

    python tools/layers_isolated.py [substring ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import fixtures as fx
from repo_amd import functional as Fn
from repo_amd import ops

dev = torch.device("cuda")
IMAGE = int(os.environ.get("ISO_IMAGE", "64"))   # 128: the six layers of the build-defined 128 x 128 stack at B=32
L, B, H, A, D, S, E = 50, (50 if IMAGE == 64 else 32), 15, (6 if IMAGE == 64 else 7), 200, 30, 1024
T, Hm = L - 1, H - 1
N = T * B          # 2450 frames / start states
NI = Hm * N        # 34300 imagined rows
PEAK_TF, PEAK_GB = 157.3, 8000.0
g = torch.Generator(device="cuda").manual_seed(0)


def r(*s, scale=1.0):
    return torch.randn(*s, device=dev, generator=g) * scale


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream()
    e0.record(s)
    for _ in range(iters):
        fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


ROWS = []


def case(name, flop=None, bytes_=None):
    def deco(make):
        ROWS.append((name, flop, bytes_, make))
        return make
    return deco


LAYERS = ["enc1", "enc2", "enc3", "enc4", "dec2", "dec3", "dec4", "enc1@128", "enc2@128", "enc3@128", "enc4@128", "dec4@128",
          "dec5@128", "dec4@tia"]
for li, nm in enumerate(LAYERS):
    if ("@128" in nm) != (IMAGE == 128) and nm != "dec3" or ("@tia" in nm and IMAGE == 128):
        continue
    (cb, hb, _), (cs, hs, _) = ops.conv_shapes(li)
    ks = ops.CONV_GEO[li][3]
    fl = 2.0 * N * cs * hs * hs * cb * ks * ks

    def mk(li=li, cb=cb, hb=hb, cs=cs, hs=hs, ks=ks):
        u8 = li in (ops.ENC1, ops.X_ENC1)
        big = (torch.randint(0, 256, (N, cb, hb, hb), device=dev, dtype=torch.uint8) if u8
               else r(N, cb, hb, hb).relu_())
        small = r(N, cs, hs, hs).relu_()
        w = r(cs, cb, ks, ks, scale=0.05)
        bias_s, bias_b = r(cs), r(cb)
        return big, small, w, bias_s, bias_b

    if nm.startswith("enc"):
        role = {"down": "forward", "up": "data-gradient", "wgrad": "weight-gradient"}
    else:
        role = {"up": "forward", "down": "data-gradient", "wgrad": "weight-gradient"}

    def down(li=li, mk=mk):
        big, small, w, bs, bb = mk()
        out = torch.empty_like(small)
        if LAYERS[li].startswith("enc"):
            # as the update calls it: enc2 and enc3 also write the channel-quad mask of their ReLU
            wm = LAYERS[li] in ("enc2", "enc3") and os.environ.get("ISO_MASK", "1") == "1"
            return lambda: ops.conv_down(li, big, w, bs, epi=ops.EPI_RELU, out=out, want_cmask=wm)
        aux = small.clone()
        if LAYERS[li] == "dec4" and os.environ.get("ISO_MASK", "1") == "1":   # as the update calls it: the ReLU operand is the quad mask of the fused output layer
            mask = torch.randint(0, 16, (small.numel() // 4,), device=dev, dtype=torch.uint8)
            return lambda: ops.conv_down(li, big.float(), w, None, epi=ops.EPI_MUL_MASK4, aux=mask, out=out)
        return lambda: ops.conv_down(li, big.float(), w, None, epi=ops.EPI_MUL_DRELU, aux=aux, out=out)

    def up(li=li, mk=mk):
        big, small, w, bs, bb = mk()
        bigf = big.float() if big.dtype != torch.float32 else big
        out = torch.empty_like(bigf)
        if LAYERS[li].startswith("dec"):
            return lambda: ops.conv_up(li, small, w, bb, epi=ops.EPI_RELU, out=out)
        if LAYERS[li] in ("enc3", "enc4") and os.environ.get("ISO_MASK", "1") == "1":
            # as the update calls it: the ReLU operand is the channel-quad mask the forward wrote (1/16 of the bytes)
            cmask = torch.randint(0, 16, (bigf.numel() // 4,), device=dev, dtype=torch.uint8)
            return lambda: ops.conv_up(li, small, w, None, epi=ops.EPI_MUL_CMASK, aux=cmask, out=out)
        return lambda: ops.conv_up(li, small, w, None, epi=ops.EPI_MUL_DRELU, aux=bigf, out=out)

    def wgrad(li=li, mk=mk):
        big, small, w, bs, bb = mk()
        dw, db = torch.empty_like(w), torch.empty_like(bs)
        return lambda: ops.conv_wgrad(li, small, big, dw=dw, db=db)

    skip_up = nm.startswith("enc1")   # the frames need no gradient
    for kind, fn in (("down", down), ("up", up), ("wgrad", wgrad)):
        if kind == "up" and skip_up:
            continue
        if nm == "dec4" and kind == "up":
            continue         # the forward of dec4 is the fused NLL kernel below
        case(f"conv {nm} {kind:5s} ({role[kind]})", flop=fl)(fn)


@case("dec4 forward + pixel NLL (u8 target)", flop=2.0 * N * 32 * 30 * 30 * 3 * 36,
      bytes_=N * (32 * 900 * 4 + 3 * 4096 * (1 + 4)))
def _nll():
    h3 = r(N, 32, 30, 30).relu_()
    w, b = r(32, 3, 6, 6, scale=0.05), r(3)
    tgt = torch.randint(0, 256, (N, 3, 64, 64), device=dev, dtype=torch.uint8)
    return lambda: ops.decoder_out_nll(h3, w, b, tgt, 1e-3, want_mask=os.environ.get("ISO_MASK", "1") == "1")


@case("dec5@128 forward + pixel NLL (u8 target, gather engine)", flop=2.0 * N * 16 * 64 * 64 * 3 * 4,
      bytes_=N * (16 * 4096 * 4 + 3 * 16384 * (1 + 4)))
def _nll128():
    h4 = r(N, 16, 64, 64).relu_()
    w, b = r(16, 3, 2, 2, scale=0.05), r(3)
    tgt = torch.randint(0, 256, (N, 3, 128, 128), device=dev, dtype=torch.uint8)
    return lambda: ops.conv_up_nll(ops.X_DEC5, h4, w, b, tgt, 1e-3)


def gemm(M, Nn, K, tb=True, epi=ops.EPI_NONE):
    Am = r(M, K)
    Bm = r(Nn, K) if tb else r(K, Nn)
    bias = r(Nn)
    out = torch.empty(M, Nn, device=dev)
    return lambda: ops.gemm(Am, Bm, False, tb, bias=bias, out=out, epi=epi)


for (M, Nn, K, tb, what) in [
    (N, 1024, 230, True, "decoder fc1 fwd (two-layer form: small batches / REPO_DEC_COMPOSE=0)"),
    (N, 3200, 1024, False, "decoder conv1 (1x1->5x5) fwd (two-layer form)"),
    (N, 1024, 3200, True, "decoder conv1 dgrad (two-layer form)"),
    (N, 200, 1224, True, "posterior embed (hoisted) fwd"),
    (N, 1024, 9216, True, "encoder fc@128 fwd"),
    (N, 9216, 1024, False, "encoder fc@128 dgrad"),
    (NI, 200, 230, True, "head layer 1 fwd (34300 rows)"),
    (NI, 200, 200, True, "head layer 2/3 fwd (34300 rows)"),
    (NI, 200, 200, False, "head layer dgrad (34300 rows)"),
    (NI, 230, 200, False, "head layer 1 dgrad (34300 rows)"),
]:
    case(f"gemm {M}x{Nn}x{K} {'nt' if tb else 'nn'}  {what}", flop=2.0 * M * Nn * K)(
        lambda M=M, Nn=Nn, K=K, tb=tb: gemm(M, Nn, K, tb, ops.EPI_ELU if "head" in what else ops.EPI_NONE))


@case("gemm_wgrad 34300 rows 200x200 (head weight gradient)", flop=2.0 * NI * 200 * 200)
def _wg():
    dY, X = r(NI, 200), r(NI, 200)
    dW, db = torch.empty(200, 200, device=dev), torch.empty(200, device=dev)
    return lambda: ops.gemm_wgrad(dY, X, dW=dW, db=db)


@case("gemm_wgrad encoder fc@128 weight gradient (1024 x 9216)", flop=2.0 * N * 9216 * 1024)
def _wgfc():
    dY, X = r(N, 1024), r(N, 9216)
    dW, db = torch.empty(1024, 9216, device=dev), torch.empty(1024, device=dev)
    return lambda: ops.gemm_wgrad(dY, X, dW=dW, db=db)


@case("gemm_wgrad 2450 rows 1024x3200 (decoder conv1 weight gradient, two-layer form)", flop=2.0 * N * 3200 * 1024)
def _wg2():
    dY, X = r(N, 1024), r(N, 3200)
    dW = torch.empty(1024, 3200, device=dev)
    return lambda: ops.gemm_wgrad(dY, X, dW=dW, db=None, want_bias=False)


# the decoder's first two layers composed (repo_amd/functional.py, dec_head_compose): what the update runs since round 6
@case("dec head: W01aug = W1^T [W0|b0]  3200x232x1024 nn (+ the 1024x3200 transposing copy)", flop=2.0 * 3200 * 232 * 1024)
def _dh_compose():
    w1, w0aug = r(1024, 3200), r(1024, 232)
    return lambda: ops.gemm(ops.transpose(w1), w0aug)


@case("dec head: h1 = relu(feat W01^T + b01)  2450x3200x230 nt", flop=2.0 * N * 3200 * 230)
def _dh_fwd():
    feat, w01aug, b01 = r(N, 230), r(3200, 232), r(3200)
    return lambda: ops.gemm(feat, w01aug[:, :230], transb=True, bias=b01, epi=ops.EPI_RELU)


@case("dec head: (G|s) = d1^T feat  3200x230 over 2450 rows", flop=2.0 * N * 3200 * 230)
def _dh_g():
    d1, feat, gaug = r(N, 3200), r(N, 230), torch.zeros(3200, 232, device=dev)
    return lambda: ops.gemm_wgrad(d1, feat, dW=gaug[:, :230])


@case("dec head: d W1 = [W0|b0] (G|s)^T  1024x3200x232 nt", flop=2.0 * 1024 * 3200 * 232)
def _dh_dw1():
    w0aug, gaug, out = r(1024, 232), r(3200, 232), torch.empty(1024, 3200, device=dev)
    return lambda: ops.gemm(w0aug, gaug, transb=True, out=out)


@case("dec head: [d W0|d b0] = W1 (G|s)  1024x232 over the 3200 rows of W1^T", flop=2.0 * 1024 * 3200 * 232)
def _dh_dw0():
    w1t, gaug = r(3200, 1024), r(3200, 232)
    return lambda: ops.gemm_wgrad(w1t, gaug, want_bias=False)


P = fx.make_params(A, 7)
rp = [torch.tensor(v).cuda() for v in P["transition_model"].values()]
ap = [torch.tensor(v).cuda() for v in P["actor_model"].values()]
vp = [torch.tensor(v).cuda() for v in P["value_model"].values()]
MLP3_MAC = 230 * 200 + 200 * 200 * 2 + 200
MLP4_MAC = 230 * 200 + 200 * 200 * 3 + 200 * 12


@case("mlp_fwd value/reward head, 34300 rows (3 ELU layers + out)", flop=2.0 * NI * MLP3_MAC)
def _mf():
    x = r(NI, 230)
    return lambda: ops.mlp_fwd(vp, x)


@case("mlp_bwd value head input-gradient only, 34300 rows", flop=2.0 * NI * MLP3_MAC)
def _mb():
    x = r(NI, 230)
    out, hid = ops.mlp_fwd(vp, x)
    dout, dx = r(NI, 1), torch.empty(NI, 230, device=dev)
    return lambda: ops.mlp_bwd(vp, x, hid, dout, dparams=None, dx=dx)


@case("mlp_bwd value head weight-gradients only, 31850 rows", flop=2.0 * (NI - N) * MLP3_MAC)
def _mw():
    x = r(NI - N, 230)
    out, hid = ops.mlp_fwd(vp, x)
    dout = r(NI - N, 1)
    gp = [torch.zeros_like(v) for v in vp]
    return lambda: ops.mlp_bwd(vp, x, hid, dout, dparams=gp, dx=None)


@case("mlp_bwd actor trunk weight-gradients, 36750 rows (4 ELU layers + head)", flop=2.0 * (NI + N) * MLP4_MAC * 2)
def _ma():
    x = r(NI + N, 230)
    out, hid = ops.mlp_fwd(ap, x)
    dout = r(NI + N, 12)
    gp = [torch.zeros_like(v) for v in ap]
    return lambda: ops.mlp_bwd(ap, x, hid, dout, dparams=gp, dx=None)


SCAN_MAC = 351e3 + 205e3  # per row-step


@case("observe scan fwd (T=49,B=50; packs + hoisted embed GEMM + persistent scan, prior head in the scan)", flop=2.0 * N * SCAN_MAC)
def _of():
    act, non, emb = r(T, B, A), torch.ones(T, B, device=dev), r(T, B, E).relu_()
    e1, e2, b0, s0 = r(T, B, S), r(T, B, S), r(B, D, scale=0.3), r(B, S)
    return lambda: ops.rssm_observe_fwd(rp, b0, s0, act, non, emb, e1, e2)


@case("observe scan fwd as the update calls it (prior head hoisted: 2 GEMMs + sample on a side stream, joined)",
      flop=2.0 * N * SCAN_MAC)
def _ofh():
    act, non, emb = r(T, B, A), torch.ones(T, B, device=dev), r(T, B, E).relu_()
    e1, e2, b0, s0 = r(T, B, S), r(T, B, S), r(B, D, scale=0.3), r(B, S)
    side = torch.cuda.Stream()

    def run():
        ops.rssm_observe_fwd(rp, b0, s0, act, non, emb, e1, e2, prior_stream=side)
        torch.cuda.current_stream().wait_stream(side)
    return run


@case("observe scan bwd (reverse scan + 8 deferred weight-gradient GEMMs + d-embed GEMM)", flop=4.0 * N * SCAN_MAC)
def _ob():
    act, non, emb = r(T, B, A), torch.ones(T, B, device=dev), r(T, B, E).relu_()
    e1, e2, b0, s0 = r(T, B, S), r(T, B, S), r(B, D, scale=0.3), r(B, S)
    sv = ops.rssm_observe_fwd(rp, b0, s0, act, non, emb, e1, e2)
    dp = [torch.zeros_like(v) for v in rp]
    dfeat = r(T, B, D + S, scale=0.1)
    dq = [r(T, B, S, scale=0.1) for _ in range(4)]
    dem = torch.empty(T, B, E, device=dev)
    return lambda: ops.rssm_observe_bwd(rp, sv, dp, dfeat=dfeat, dpm=dq[0], dps=dq[1], dqm=dq[2], dqs=dq[3], dembeds=dem)


IMG_MAC = 467.6e3


@case("imagine fwd (14 steps x 2450 rows, rollout + actor)", flop=2.0 * NI * IMG_MAC)
def _if():
    b0, s0 = r(N, D, scale=0.3), r(N, S)
    ea, ep = r(Hm, N, A), r(Hm, N, S)
    return lambda: ops.rssm_imagine_fwd(rp, ap, b0, s0, ea, ep)


@case("imagine bwd (reverse rollout, frozen weights)", flop=2.0 * NI * (IMG_MAC - MLP4_MAC))
def _ib():
    b0, s0 = r(N, D, scale=0.3), r(N, S)
    ea, ep = r(Hm, N, A), r(Hm, N, S)
    sv = ops.rssm_imagine_fwd(rp, ap, b0, s0, ea, ep)
    dfeat = r(Hm, N, D + S, scale=0.01)
    return lambda: ops.rssm_imagine_bwd(rp, sv, dfeat)


@case("tanh-Normal entropy, 100 samples x 34300 rows x 6 (+grads)", bytes_=100 * NI * A * 4 + 4 * NI * A * 4)
def _ent():
    mean, std, eps = r(NI, A), r(NI, A).abs() + 0.1, r(100, NI, A)
    return lambda: ops.tanh_normal_entropy(mean, std, eps, gscale=1e-5)


@case("clip + Adam, model group (5.17 M floats)", bytes_=5170420 * 4 * 7)
def _adam():
    n = 5170420
    p_, g_, m_, v_ = r(n), r(n), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    sq = torch.ones(1, device=dev)
    return lambda: (ops.grad_sqnorm(g_, out=sq), ops.clip_adam(p_, g_, m_, v_, sq, 100.0, 3e-4, 1))


@case("channel_sum d recon (2450 x 3 x 4096)", bytes_=N * 3 * 4096 * 4)
def _cs():
    x, out = r(N, 3, 64, 64), torch.empty(3, device=dev)
    return lambda: ops.channel_sum(x, out=out)


@case("channel_sum d h3 (2450 x 32 x 900)", bytes_=N * 32 * 900 * 4)
def _cs2():
    x, out = r(N, 32, 30, 30), torch.empty(32, device=dev)
    return lambda: ops.channel_sum(x, out=out)


@case("torch.randn noise of one update (5 tensors, 21.3 M floats)", bytes_=(2 * T * B * S + Hm * N * (A + S) + 100 * NI * A) * 4)
def _noise():
    shapes = [(T, B, S), (T, B, S), (Hm, N, A), (Hm, N, S), (100, NI, A)]
    return lambda: [torch.randn(*s, device=dev) for s in shapes]


def main():
    pats = sys.argv[1:]
    print(f"# isolated kernels, B={B} L={L} H={H} A={A}: {torch.cuda.get_device_properties(0).gcnArchName}, torch {torch.__version__}")
    print(f"# {'kernel / call':86s} {'us':>9s} {'TFLOP/s':>9s} {'GB/s':>8s} {'frac':>6s}")
    total = 0.0
    for name, flop, nbytes, make in ROWS:
        if pats and not any(p in name for p in pats):
            continue
        fn = make()
        us = timeit(fn)
        total += us
        tf = flop / us / 1e6 if flop else None
        gb = nbytes / us / 1e3 if nbytes else None
        frac = (tf / PEAK_TF) if tf is not None and (gb is None or tf / PEAK_TF > gb / PEAK_GB) else gb / PEAK_GB
        print(f"{name:88s} {us:9.1f} {tf if tf is not None else float('nan'):9.2f} {gb if gb is not None else float('nan'):8.0f} {frac:6.3f}",
              flush=True)
        del fn
        torch.cuda.empty_cache()
    print(f"# sum of the rows above: {total:.0f} us")


if __name__ == "__main__":
    main()
