import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repo_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda")
for M in (2450, 34300):
    for K in (16, 32, 64, 128, 256, 512, 1024):
        A = torch.randn(M, K, device=dev); B = torch.randn(200, K, device=dev); out = torch.empty(M, 200, device=dev)
        us = timeit(lambda: ops.gemm(A, B, False, True, out=out), iters=20)
        print(f"M={M} K={K:5d}: {us:8.1f} us", flush=True)
# host overhead of one python->C call: launch an empty-ish gemm (M=1)
A = torch.randn(1, 16, device=dev); B = torch.randn(1, 16, device=dev); out = torch.empty(1, 1, device=dev)
print("tiny gemm", timeit(lambda: ops.gemm(A, B, False, True, out=out), iters=50), "us")
