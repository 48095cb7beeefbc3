#!/usr/bin/env python3
"""observe scan forward alone (hoisted embed GEMM + packs + scan + prior head), row scan (rssm.hip) vs the column-split
weight-stationary scan (scan_cs.hip), at the batch sizes of the strong-scaling shards and of one GPU: us per call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import fixtures as fx
from repo_amd import ops

T, A, D, S, E = 49, 6, 200, 30, 1024
p = [torch.tensor(v).cuda() for v in fx.make_params(A, 7)["transition_model"].values()]
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, iters=20, warm=3):
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


print("# observe scan (T=49), us per call: forward row scan | column-split;  reverse (scan + deferred weight-gradient GEMMs) row | column-split")
for B in (6, 7, 13, 16, 25, 32, 50):
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    act, non, emb = r(T, B, A), torch.ones(T, B, device="cuda"), r(T, B, E).relu_()
    b0, s0 = r(B, D) * 0.3, r(B, S)
    row, rev = [], []
    ups = dict(dfeat=r(T, B, D + S), dpm=r(T, B, S), dps=r(T, B, S), dqm=r(T, B, S), dqs=r(T, B, S))
    gp = [torch.zeros_like(t) for t in p]
    dembeds = torch.empty(T, B, E, device="cuda")
    for mode in ("0", "1"):
        os.environ["REPO_SCAN_CS"] = mode
        row.append(timeit(lambda: ops.rssm_observe_fwd(p, b0, s0, act, non, emb, None, None, 0.1, noise=(1, 0))))
        sv = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, None, None, 0.1, noise=(1, 0))
        rev.append(timeit(lambda: ops.rssm_observe_bwd(p, sv, gp, dembeds=dembeds, **ups)))
    print(f"  B={B:3d}  fwd {row[0]:8.1f} {row[1]:8.1f}   bwd {rev[0]:8.1f} {rev[1]:8.1f}", flush=True)
