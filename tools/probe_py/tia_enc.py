"""TIA test agent's encoder forward + backward on the test's own frames: dump for a cross-library comparison."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_tia_gpu as T
from repo_amd import functional as Fn
over = dict(tia_obs_coef=0.7, tia_adv_coef=1.3, tia_reward_train_steps=2, free_nats=0.1)
agent, cfg = T.make_tia(9, 5, 5, 6, **over)
batch, host = T.dev_batch(9, 5, 6, 60, u8=(sys.argv[2] == "u8"))
obs = batch[0]
frames = obs[1:].reshape(40, *obs.shape[2:])
pe, ge = agent._pg(agent.encoder)
print([ (tuple(x.shape), x.data_ptr() % 16) for x in pe ])
emb, saved = Fn.encoder_fwd(pe, frames)
torch.manual_seed(3)
d = torch.randn_like(emb)
g = [torch.zeros_like(x) for x in pe]
Fn.encoder_bwd(pe, frames, saved, d, g)
torch.cuda.synchronize()
torch.save([emb.cpu()] + [x.cpu() for x in g] + [None if s is None else s.cpu() for s in saved], sys.argv[1])
