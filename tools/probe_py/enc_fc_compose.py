"""The 128 x 128 stack's encoder `fc` (9216 -> 1024, no activation) feeding the posterior's embedding columns (1024 -> 200):
the GEMMs of the two-layer form against those of a composed layer Wc = W_e W_fc, alone at 1568 rows (B=32, L=50)."""
import sys

import torch

sys.path.insert(0, ".")
from repo_amd import ops  # noqa: E402

torch.manual_seed(0)
rows, dev = 1568, "cuda"
flat = torch.randn(rows, 9216, device=dev)
wfc, bfc = torch.randn(1024, 9216, device=dev) * 0.01, torch.randn(1024, device=dev)
wbq = torch.randn(200, 1224, device=dev) * 0.03
we = wbq[:, 200:]                       # (200, 1024), ld 1224
emb = torch.randn(rows, 1024, device=dev)
dhq = torch.randn(rows, 200, device=dev)
demb = torch.randn(rows, 1024, device=dev)
wc = torch.randn(200, 9216, device=dev)
G = torch.randn(200, 9216, device=dev)


def t(name, fn, flop):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 10
    print(f"{name:72s} {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)


print("# two-layer form")
t("fc fwd        rows x 1024 x 9216 nt", lambda: ops.gemm(flat, wfc, transb=True, bias=bfc), 2 * rows * 1024 * 9216)
t("eemb          rows x 200 x 1024 nt", lambda: ops.gemm(emb, we, transb=True), 2 * rows * 200 * 1024)
t("d emb         rows x 1024 x 200 nn", lambda: ops.gemm(dhq, we), 2 * rows * 200 * 1024)
t("d W_e         200 x 1024 over rows", lambda: ops.gemm_wgrad(dhq, emb, want_bias=False), 2 * rows * 200 * 1024)
t("fc dgrad      rows x 9216 x 1024 nn", lambda: ops.gemm(demb, wfc), 2 * rows * 1024 * 9216)
t("fc wgrad      1024 x 9216 over rows (+db)", lambda: ops.gemm_wgrad(demb, flat), 2 * rows * 1024 * 9216)
print("# composed form")
t("Wc = W_e W_fc 200 x 9216 x 1024 nn", lambda: ops.gemm(we, wfc), 2 * 200 * 9216 * 1024)
t("eemb          rows x 200 x 9216 nt", lambda: ops.gemm(flat, wc, transb=True), 2 * rows * 200 * 9216)
t("d flat        rows x 9216 x 200 nn", lambda: ops.gemm(dhq, wc), 2 * rows * 200 * 9216)
t("G             200 x 9216 over rows (+db)", lambda: ops.gemm_wgrad(dhq, flat), 2 * rows * 200 * 9216)
t("d W_fc        1024 x 9216 x 200 tn (W_e^T G)", lambda: ops.gemm(we, G, transa=True), 2 * 1024 * 9216 * 200)
t("d W_e         200 x 1024 x 9216 nt (G W_fc^T)", lambda: ops.gemm(G, wfc, transb=True), 2 * 200 * 1024 * 9216)
