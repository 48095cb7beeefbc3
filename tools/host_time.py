"""Host enqueue time per update vs GPU time (is the Python launch path the bottleneck?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    from repo_amd.algorithms.repo.repo import RePo
    torch.manual_seed(0)
    agent = RePo(bench.config("repo"), bench.Env(), bench.Env(), bench.NullLogger())
    host = bench.synthetic_batch(1234)
    batch = tuple(torch.from_numpy(x).cuda() for x in host)
    for _ in range(5):
        agent.update(batch, join=False)
    agent.synchronize(); torch.cuda.synchronize()
    K = 12
    base = torch.cuda.Event(enable_timing=True); base.record(); torch.cuda.synchronize()
    h0 = time.perf_counter()
    marks = []
    orig_td, orig_ac = agent.train_dynamics, agent.train_actor_critic
    def td(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        t_in = time.perf_counter()
        r = orig_td(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        marks.append(["wm", t_in - h0, time.perf_counter() - h0, e0, e1])
        return r
    def ac(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        t_in = time.perf_counter()
        r = orig_ac(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        marks.append(["ac", t_in - h0, time.perf_counter() - h0, e0, e1])
        return r
    agent.train_dynamics, agent.train_actor_critic = td, ac
    for _ in range(K):
        agent.update(batch, join=False)
    agent.synchronize(); torch.cuda.synchronize()
    for kind, hin, hout, e0, e1 in marks:
        print(f"{kind}: host enqueue [{1e3*hin:8.2f}, {1e3*hout:8.2f}] ms   gpu [{base.elapsed_time(e0):8.2f}, {base.elapsed_time(e1):8.2f}] ms")

main()
