"""The two recurrences alone at the update's size (for rocprofv3 --pmc): observe scan fwd+bwd, rollout fwd+bwd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import fixtures as fx
from repo_amd import ops
T, B, A, D, S, E, Hm = 49, 50, 6, 200, 30, 1024, 14
N = T * B
P = fx.make_params(A, 7)
rp = [torch.tensor(v).cuda() for v in P["transition_model"].values()]
ap = [torch.tensor(v).cuda() for v in P["actor_model"].values()]
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
act, non, emb = r(T, B, A), torch.ones(T, B, device="cuda"), r(T, B, E).relu()
b0, s0 = r(B, D) * 0.3, r(B, S)
dp = [torch.zeros_like(v) for v in rp]
dfeat, dq, dem = r(T, B, D + S) * 0.1, [r(T, B, S) * 0.1 for _ in range(4)], torch.empty(T, B, E, device="cuda")
ib, is_ = r(N, D) * 0.3, r(N, S)
difeat = r(Hm, N, D + S) * 0.01
side = torch.cuda.Stream()
for i in range(4):
    sv = ops.rssm_observe_fwd(rp, b0, s0, act, non, emb, None, None, noise=(1, 10 * i), prior_stream=side)  # as the update calls it
    torch.cuda.current_stream().wait_stream(side)
    ops.rssm_observe_bwd(rp, sv, dp, dfeat=dfeat, dpm=dq[0], dps=dq[1], dqm=dq[2], dqs=dq[3], dembeds=dem)
    si = ops.rssm_imagine_fwd(rp, ap, ib, is_, None, None, noise=(2, 10 * i), horizon=Hm)
    ops.rssm_imagine_bwd(rp, si, difeat)
torch.cuda.synchronize()
