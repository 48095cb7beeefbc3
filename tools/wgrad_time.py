import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repo_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda")
for (M, N, K) in ((34300, 200, 200), (34300, 200, 230), (2450, 200, 230), (2450, 600, 200)):
    dY = torch.randn(M, N, device=dev); X = torch.randn(M, K, device=dev)
    us = timeit(lambda: ops.gemm_wgrad(dY, X), iters=20)
    print(f"wt={os.environ.get('REPO_WGRAD_TILE','0')} wgrad M={M} N={N} K={K}: {us:8.1f} us {2*M*N*K/us/1e6:6.1f} TF", flush=True)
