import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repo_amd import ops
torch.manual_seed(0)
for M in (1, 3, 28, 31, 112, 700, 34300):
    for N in (200, 7, 1):
        A = torch.randn(M, 1, device="cuda"); B = torch.randn(1, N, device="cuda")
        h = torch.randn(M, N, device="cuda")
        want = (A @ B) * torch.where(h > 0, torch.ones_like(h), h + 1)
        got = ops.gemm(A, B, epi=ops.EPI_MUL_DELU, aux=h)
        e = (got - want).abs().max().item()
        fin = torch.isfinite(got).all().item()
        # column view of a wider matrix (lda != 1)
        Aw = torch.randn(M, 5, device="cuda"); Av = Aw[:, 2:3]
        got2 = ops.gemm(Av, B)
        e2 = (got2 - Av @ B).abs().max().item()
        print(M, N, f"err {e:.2e} finite {fin}  view err {e2:.2e}", flush=True)
