"""Run the imagination rollout (fwd+bwd) at the update's size a few times (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import fixtures as fx
from repo_amd import ops
Hm, N, A, D, S = 14, 2450, 6, 200, 30
P = fx.make_params(A, 7)
rp = [torch.tensor(v).cuda() for v in P["transition_model"].values()]
ap = [torch.tensor(v).cuda() for v in P["actor_model"].values()]
b0 = torch.randn(N, D, device="cuda") * 0.3; s0 = torch.randn(N, S, device="cuda")
ea = torch.randn(Hm, N, A, device="cuda"); ep = torch.randn(Hm, N, S, device="cuda")
dfeat = torch.randn(Hm, N, D + S, device="cuda") * 0.01
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    sv = ops.rssm_imagine_fwd(rp, ap, b0, s0, ea, ep)
    ops.rssm_imagine_bwd(rp, sv, dfeat)
torch.cuda.synchronize()
print("done")
