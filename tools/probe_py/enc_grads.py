"""encoder forward + backward at nimg images: writes embeds and the eight gradients to a file (compare two library variants)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from repo_amd import functional as Fn
nimg, out = int(sys.argv[1]), sys.argv[2]
torch.manual_seed(1)
obs = (torch.rand(nimg, 3, 64, 64).cuda() - 0.5)
shapes = [(32, 3, 4, 4), (32,), (64, 32, 4, 4), (64,), (128, 64, 4, 4), (128,), (256, 128, 4, 4), (256,)]
p = [(torch.randn(*s) * (0.1 if len(s) > 1 else 0.05)).cuda() for s in shapes]
emb, saved = Fn.encoder_fwd(p, obs)
d = torch.randn_like(emb)
g = [torch.zeros_like(x) for x in p]
Fn.encoder_bwd(p, obs, saved, d, g)
torch.cuda.synchronize()
torch.save([emb.cpu()] + [x.cpu() for x in g] + [None if s is None else s.cpu() for s in saved], out)
