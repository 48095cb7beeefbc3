"""Decoder conv1 (1x1 -> 5x5 transposed conv = a 1024 -> 3200 dense layer) as GEMMs: the forms the update uses (NN forward, TN
weight gradient) against the NT form of bgemm on pre-transposed operands.  Round-5 probe."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from repo_amd import ops

dev = torch.device("cuda")
M, K, N = 2450, 1024, 3200
g = torch.Generator(device="cuda").manual_seed(0)
h0 = torch.randn(M, K, device=dev, generator=g)
w1 = torch.randn(K, N, device=dev, generator=g) * 0.05
d1 = torch.randn(M, N, device=dev, generator=g)


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(it)]; e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


out = torch.empty(M, N, device=dev)
print("fwd NN (update)           %7.1f us" % t(lambda: ops.gemm(h0, w1, out=out)))
w1T = w1.t().contiguous()
print("fwd NT on w1^T            %7.1f us" % t(lambda: ops.gemm(h0, w1T, transb=True, out=out)))
print("   transpose w1 (torch)   %7.1f us" % t(lambda: w1.t().contiguous()))
ref = h0.double() @ w1.double()
print("   err NN %.2e  NT %.2e" % (float((ops.gemm(h0, w1).double() - ref).abs().max() / ref.abs().max()),
                                    float((ops.gemm(h0, w1T, transb=True).double() - ref).abs().max() / ref.abs().max())))
dW = torch.empty(K, N, device=dev)
print("wgrad TN (update)         %7.1f us" % t(lambda: ops.gemm_wgrad(h0, d1, dW=dW.view(K, N), db=None, want_bias=False)))
LD = 2452
h0T = torch.zeros(K, LD, device=dev); h0T[:, :M] = h0.t()
d1T = torch.zeros(N, LD, device=dev); d1T[:, :M] = d1.t()
print("wgrad NT on transposes    %7.1f us" % t(lambda: ops.gemm(h0T[:, :M], d1T[:, :M], transb=True, out=dW)))
print("   transposes (torch copy)%7.1f us" % t(lambda: (h0T[:, :M].copy_(h0.t()), d1T[:, :M].copy_(d1.t()))))
