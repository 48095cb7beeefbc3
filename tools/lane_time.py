"""Wall time of each lane alone (world-model chain / actor-critic chain) vs the pipelined update."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def timeit(fn, K=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K

def main():
    algo = sys.argv[1] if len(sys.argv) > 1 else "repo"
    wmonly = algo == "wmonly"
    if wmonly:
        algo = "repo"
    from repo_amd.algorithms.repo.repo import RePo
    from repo_amd.algorithms.repo.dreamer import Dreamer
    agent = (RePo if algo == "repo" else Dreamer)(bench.config(algo), bench.Env(), bench.Env(), bench.NullLogger())
    obs, act, rew, done = (torch.from_numpy(x).cuda() for x in bench.synthetic_batch(1234))
    nonterm = 1.0 - done.float()
    for _ in range(3): agent.update((obs, act, rew, done))
    out = {}
    def wm():
        out["bp"] = agent.train_dynamics(obs, act, rew, nonterm)
    print(f"world-model lane alone : {timeit(wm, 6 if wmonly else 20):.2f} ms")
    if wmonly:
        return
    b, p = out["bp"]
    def ac():
        agent.train_actor_critic(b.flatten(0, 1), p.flatten(0, 1))
    print(f"actor-critic lane alone: {timeit(ac):.2f} ms")
    print(f"joined update          : {timeit(lambda: agent.update((obs, act, rew, done))):.2f} ms")
    def pipe():
        agent.update((obs, act, rew, done), join=False)
    t = timeit(pipe); agent.synchronize()
    print(f"pipelined update       : {t:.2f} ms")
main()
