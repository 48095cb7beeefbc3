"""Instrumented copy of tests/test_host_gpu.py::test_train_and_eval_loops_on_fake_env: prints the magnitude of every latent
the acting path returns and of every reconstruction, with poisoned torch.empty (see tests/conftest.py)."""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.conftest import _poisoned  # noqa: E402
from tests.test_host_gpu import DumpLogger, FakeDMC  # noqa: E402
from tests.test_update_gpu import make_agent  # noqa: E402

if os.environ.get("POISON", "1") == "1":
    torch.empty, torch.empty_like = _poisoned(torch.empty), _poisoned(torch.empty_like)

A, hor = 6, 9
agent, cfg = make_agent("repo", 6, 3, 4, A)
cfg.replay_size, cfg.prefill, cfg.num_steps = 400, 20, 25
cfg.train_every, cfg.train_steps, cfg.eval_every, cfg.checkpoint_every, cfg.log_every = 10, 2, 20, 25, 5
cfg.action_noise, cfg.save_buffer = 0.3, True
agent.buffer = type(agent.buffer)(cfg.replay_size, (3, 64, 64), (A,), obs_type=np.uint8)
agent.buffer.enable_device_mirror(agent.device)
agent.env, agent.eval_env = FakeDMC(A, hor, 1), FakeDMC(A, hor, 2)
agent.logger = DumpLogger()
agent.logger.dir = tempfile.mkdtemp()

orig_act = agent.update_latent_and_select_action
n = [0]


def act(b, s, a, o, explore=False):
    out = orig_act(b, s, a, o, explore)
    torch.cuda.synchronize()
    mx = [float(t.float().abs().max()) if torch.isfinite(t).all() else float("nan") for t in out]
    print(f"act {n[0]:3d} explore={explore} step={agent.step} in(b,s,a)max=({float(b.abs().max()):.3g},{float(s.abs().max()):.3g},"
          f"{float(a.abs().max()):.3g}) out max={mx}", flush=True)
    n[0] += 1
    return out


agent.update_latent_and_select_action = act
orig_rec = agent._reconstruct


def rec(b, s_):
    out = orig_rec(b, s_)
    torch.cuda.synchronize()
    print(f"   recon max={float(out.abs().max()):.4g} finite={bool(torch.isfinite(out).all())} "
          f"belief max={float(b.abs().max()):.3g} state max={float(s_.abs().max()):.3g}", flush=True)
    # the same reconstruction again, eagerly and from clones of the inputs
    out2 = orig_rec(b.clone(), s_.clone())
    print(f"   again max={float(out2.abs().max()):.4g} equal={bool(torch.equal(out, out2))}", flush=True)
    return out


agent._reconstruct = rec
agent.train()
print("nonfinite logged:", agent.logger.nonfinite)
