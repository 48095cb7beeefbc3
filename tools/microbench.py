#!/usr/bin/env python3
"""Micro-benchmarks of individual kernels at the update's shapes (B=50, L=50, H=15).
Usage: python tools/microbench.py [name ...]   (HIP-event timing, us per call)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repo_amd import ops

dev = torch.device("cuda")
N = 2450


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def conv_case(layer, kind):
    (cb, hb, _), (cs, hs, _) = ops.conv_shapes(layer)
    ks = ops.CONV_GEO[layer][3]
    big = torch.randn(N, cb, hb, hb, device=dev)
    small = torch.randn(N, cs, hs, hs, device=dev)
    w = torch.randn(cs, cb, ks, ks, device=dev) * 0.05
    flop = 2.0 * N * cs * hs * hs * cb * ks * ks
    if kind == "down":
        out = torch.empty_like(small)
        return (lambda: ops.conv_down(layer, big, w, None, epi=ops.EPI_RELU, out=out)), flop
    if kind == "up":
        out = torch.empty_like(big)
        return (lambda: ops.conv_up(layer, small, w, None, epi=ops.EPI_RELU, out=out)), flop
    dw = torch.empty_like(w)
    db = torch.empty(cs, device=dev)
    return (lambda: ops.conv_wgrad(layer, small, big, dw=dw, db=db)), flop


def gemm_case(M, Nn, K, ta=False, tb=True):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    Bm = torch.randn((Nn, K) if tb else (K, Nn), device=dev)
    out = torch.empty(M, Nn, device=dev)
    return (lambda: ops.gemm(A, Bm, ta, tb, out=out)), 2.0 * M * Nn * K


CASES = {}
for l, nm in enumerate(["enc1", "enc2", "enc3", "enc4", "dec2", "dec3", "dec4"]):
    for kind in ("down", "up", "wgrad"):
        CASES[f"{nm}_{kind}"] = (lambda l=l, kind=kind: conv_case(l, kind))
CASES["gemm_2450x200x200_nt"] = lambda: gemm_case(2450, 200, 200)
CASES["gemm_2450x600x200_nt"] = lambda: gemm_case(2450, 600, 200)
CASES["gemm_2450x200x600_nn"] = lambda: gemm_case(2450, 200, 600, tb=False)
CASES["gemm_34300x200x200_nt"] = lambda: gemm_case(34300, 200, 200)
CASES["gemm_34300x200x230_nt"] = lambda: gemm_case(34300, 200, 230)
CASES["gemm_2450x3200x1024_nn"] = lambda: gemm_case(2450, 3200, 1024, tb=False)
CASES["gemm_2450x1024x3200_nt"] = lambda: gemm_case(2450, 1024, 3200)


def main():
    names = sys.argv[1:] or list(CASES)
    for n in names:
        fn, flop = CASES[n]()
        us = timeit(fn)
        print(f"{n:28s} {us:9.1f} us  {flop / us / 1e6:7.2f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
