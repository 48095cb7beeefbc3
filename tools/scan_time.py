"""Isolated duration of the observe scan kernels at T=49, B=50."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import fixtures as fx
from repo_amd import ops
from tools.microbench import timeit
T, B, A, D, S, E = 49, 50, 6, 200, 30, 1024
p = [torch.tensor(v).cuda() for v in fx.make_params(A, 7)["transition_model"].values()]
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
act, non, emb = r(T, B, A), torch.ones(T, B, device="cuda"), r(T, B, E).relu()
e1, e2, b0, s0 = r(T, B, S), r(T, B, S), r(B, D) * 0.3, r(B, S)
sv = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, e1, e2)
print(f"observe fwd (incl. packs + hoisted GEMM): {timeit(lambda: ops.rssm_observe_fwd(p, b0, s0, act, non, emb, e1, e2), iters=10):.0f} us")
dp = [torch.zeros_like(v) for v in p]
dfeat = r(T, B, D + S) * 0.1
dq = [r(T, B, S) * 0.1 for _ in range(4)]
dem = torch.empty(T, B, E, device="cuda")
f = lambda: ops.rssm_observe_bwd(p, sv, dp, dfeat=dfeat, dpm=dq[0], dps=dq[1], dqm=dq[2], dqs=dq[3], dembeds=dem)
print(f"observe bwd (incl. packs + 8 wgrad GEMMs + dembeds GEMM): {timeit(f, iters=10):.0f} us")
