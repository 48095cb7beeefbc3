import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repo_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda")
n = 2450
h3 = torch.randn(n, 32, 30, 30, device=dev); w = torch.randn(32, 3, 6, 6, device=dev) * 0.05; b = torch.randn(3, device=dev)
tgt = torch.randint(0, 255, (n, 3, 64, 64), device=dev, dtype=torch.uint8)
us = timeit(lambda: ops.decoder_out_nll(h3, w, b, tgt, 1e-3), iters=20)
print(f"decoder_out_nll u8 target: {us:.1f} us")
