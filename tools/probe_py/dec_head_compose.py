"""The decoder's first two layers as one composed Linear (repo_amd/functional.py, _dec_head_compose): every GEMM of the
two forms alone on an idle GPU at the update's shapes (rows = 2450)."""
import sys

import torch

sys.path.insert(0, ".")
from repo_amd import ops  # noqa: E402

torch.manual_seed(0)
rows = 2450
dev = "cuda"
feat = torch.randn(rows, 230, device=dev)
w0, b0 = torch.randn(1024, 230, device=dev) * 0.05, torch.randn(1024, device=dev)
w1, b1 = torch.randn(1024, 3200, device=dev) * 0.03, torch.randn(128, device=dev)
d1 = torch.randn(rows, 3200, device=dev)
h0 = torch.randn(rows, 1024, device=dev)
G = torch.randn(3200, 230, device=dev)
s = torch.randn(3200, device=dev)
w01 = torch.randn(3200, 230, device=dev)
b01 = torch.randn(3200, device=dev)
gw1, gw0, gb0 = torch.zeros(1024, 3200, device=dev), torch.zeros(1024, 230, device=dev), torch.zeros(1024, 1, device=dev)


def t(name, fn, flop):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"{name:70s} {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)


print("# two-layer form")
t("fc1 fwd            2450 x 1024 x 230  nt", lambda: ops.gemm(feat, w0, transb=True, bias=b0), 2 * rows * 1024 * 230)
t("conv1 fwd          2450 x 3200 x 1024 nn (+relu)", lambda: ops.gemm(h0, w1, bias=b1, bias_div=25, epi=ops.EPI_RELU), 2 * rows * 3200 * 1024)
t("conv1 dgrad        2450 x 1024 x 3200 nt", lambda: ops.gemm(d1, w1, transb=True), 2 * rows * 3200 * 1024)
t("conv1 wgrad        1024 x 3200 over 2450 rows", lambda: ops.gemm_wgrad(h0, d1, dW=gw1, db=None, want_bias=False), 2 * rows * 3200 * 1024)
t("fc1 wgrad          1024 x 230 over 2450 rows (+db)", lambda: ops.gemm_wgrad(h0, feat, dW=gw0, db=gb0.view(-1)), 2 * rows * 1024 * 230)
print("# composed form (bias terms as a 231st column; W1 G on the row-split weight-gradient engine)")
w0aug = torch.zeros(1024, 232, device=dev)
w0aug[:, :230].copy_(w0)
w0aug[:, 230].copy_(b0)
w1t = ops.transpose(w1)
w01aug = ops.gemm(w1t, w0aug)
gaug = torch.zeros(3200, 232, device=dev)
t("W1^T              transpose 1024 x 3200", lambda: ops.transpose(w1), 0)
t("W01aug = W1^T [W0|b0]   3200 x 232 x 1024 nn", lambda: ops.gemm(w1t, w0aug), 2 * 3200 * 232 * 1024)
t("h1 = relu(feat W01^T + b01)  2450 x 3200 x 230 nt", lambda: ops.gemm(feat, w01aug[:, :230], transb=True, bias=b01, epi=ops.EPI_RELU), 2 * rows * 3200 * 230)
t("G, s = d1^T feat   3200 x 230 over 2450 rows (+db)", lambda: ops.gemm_wgrad(d1, feat, dW=gaug[:, :230]), 2 * rows * 3200 * 230)
t("d W1 = [W0|b0] [G|s]^T  1024 x 3200 x 232 nt", lambda: ops.gemm(w0aug, gaug, transb=True, out=gw1), 2 * 1024 * 3200 * 232)
t("[d W0|d b0] = W1 [G|s]  1024 x 232 over 3200 rows", lambda: ops.gemm_wgrad(w1t, gaug, want_bias=False), 2 * 1024 * 3200 * 232)
t("d feat = d1 W01    2450 x 230 x 3200  nn (Dreamer)", lambda: ops.gemm(d1, w01aug[:, :230]), 2 * rows * 3200 * 230)
print("# with the feature rows at a pitch of 232 ([feat | 1 | 0]): both K = 232 products on aligned operands")
featp = torch.cat([feat, torch.ones(rows, 1, device=dev), torch.zeros(rows, 1, device=dev)], 1).contiguous()
w01b = torch.randn(3200, 232, device=dev)
t("[feat | 1 | 0]     torch.cat copy 2450 x 232", lambda: torch.cat([feat, torch.ones(rows, 1, device=dev), torch.zeros(rows, 1, device=dev)], 1), 0)
t("h1 = relu([feat|1] [W01|b01]^T)  2450 x 3200 x 232 nt", lambda: ops.gemm(featp, w01b, transb=True, epi=ops.EPI_RELU), 2 * rows * 3200 * 232)
t("(G|s) = d1^T [feat|1]  3200 x 232 over 2450 rows (no db)", lambda: ops.gemm_wgrad(d1, featp, want_bias=False), 2 * rows * 3200 * 232)
