"""Latency of the acting path (update_latent_and_select_action, one env step, B=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from repo_amd.algorithms.repo.repo import RePo
agent = RePo(bench.config("repo"), bench.Env(), bench.Env(), bench.NullLogger())
b, s, a = agent.init_latent_and_action()
obs = torch.rand(1, 3, 64, 64, device="cuda") * 2 - 1
for explore in (True, False):
    for _ in range(20):
        b, s, a = agent.update_latent_and_select_action(b, s, a, obs, explore)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        b, s, a = agent.update_latent_and_select_action(b, s, a, obs, explore)
        a_host = a.cpu()
    dt = (time.perf_counter() - t0) / 200
    print(f"explore={explore}: {dt*1e6:.0f} us per env step (incl. action D2H)")
