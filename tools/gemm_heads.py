import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repo_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda")
for (M, N, K) in ((34300, 200, 200), (34300, 200, 230), (34300, 230, 200), (2450, 200, 230), (2450, 600, 200)):
    for tb in (False, True):
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
        out = torch.empty(M, N, device=dev)
        us = timeit(lambda: ops.gemm(A, B, False, tb, out=out, epi=ops.EPI_ELU), iters=20)
        ref = (A @ (B.t() if tb else B))
        ref = torch.nn.functional.elu(ref)
        err = (out - ref).abs().max().item()
        print(f"tile={os.environ.get('REPO_GEMM_TILE','auto')} M={M} N={N} K={K} {'nt' if tb else 'nn'}: {us:9.1f} us {2*M*N*K/us/1e6:7.1f} TF  maxerr {err:.2e}", flush=True)
