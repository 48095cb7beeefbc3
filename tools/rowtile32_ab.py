#!/usr/bin/env python3
"""A/B of the 32-row bf16x6 row-tile engines (csrc/rowtile32.h) against the 16-row fp32-MFMA engines (rowtile.h) on
the update's shapes: every saved tensor compared, both timed.    python tools/rowtile32_ab.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import fixtures as fx
from repo_amd import ops
from repo_amd._lib import lib

dev = torch.device("cuda")
Hm, A, D, S = 14, 6, 200, 30
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2450
g = torch.Generator(device="cuda").manual_seed(0)


def r(*s, scale=1.0):
    return torch.randn(*s, device=dev, generator=g) * scale


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


P = fx.make_params(A, 7)
rp = [torch.tensor(v).cuda() for v in P["transition_model"].values()]
ap = [torch.tensor(v).cuda() for v in P["actor_model"].values()]
b0, s0 = r(N, D, scale=0.3), r(N, S)
ea, ep = r(Hm, N, A), r(Hm, N, S)


def run(flag):
    lib().repo_debug_rowtile32(flag)
    sv = ops.rssm_imagine_fwd(rp, ap, b0, s0, ea, ep)
    torch.cuda.synchronize()
    return sv


a, b = run(0), run(1)
for name in ("featx", "prior_mean", "prior_std", "a_hidden", "a_raw", "a_mean", "a_std", "xsa", "e", "gates", "hp"):
    x, y = getattr(a, name), getattr(b, name)
    d = (x - y).abs().max().item()
    print(f"{name:12s} max|16-row - 32-row| = {d:.3e}   (max |x| = {x.abs().max().item():.3e})  finite: {bool(torch.isfinite(y).all())}")
for flag in (0, 1):
    lib().repo_debug_rowtile32(flag)
    print(f"imagine fwd, rowtile32={flag}: {timeit(lambda: ops.rssm_imagine_fwd(rp, ap, b0, s0, ea, ep)):.1f} us")

# ---- reverse rollout
dfeat = r(Hm, N, D + S, scale=0.01)
dpm, dps = r(Hm, N, S, scale=0.01), r(Hm, N, S, scale=0.01)
res = []
for flag in (0, 1):
    lib().repo_debug_rowtile32(flag)
    sv = ops.rssm_imagine_fwd(rp, ap, b0, s0, ea, ep)
    d_araw, dfeat0 = ops.rssm_imagine_bwd(rp, sv, dfeat, dpm, dps, want_dfeat0=True)
    torch.cuda.synchronize()
    res.append((d_araw.clone(), dfeat0.clone()))
    print(f"imagine bwd, rowtile32={flag}: {timeit(lambda: ops.rssm_imagine_bwd(rp, sv, dfeat, dpm, dps, want_dfeat0=True)):.1f} us")
for name, x, y in (("d_araw", res[0][0], res[1][0]), ("dfeat0", res[0][1], res[1][1])):
    print(f"{name:8s} max|16-row - 32-row| = {(x - y).abs().max().item():.3e}  (max |x| = {x.abs().max().item():.3e})  finite: {bool(torch.isfinite(y).all())}")
