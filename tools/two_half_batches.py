"""Probe: would pipelining two half-batches through the update hide the forward scan?  Upper bound = two B=25 agents
enqueued alternately on their own stream sets against one B=50 agent.  Measured (round 3): 11.2 ms per PAIR vs 8.08 ms
(one B=25 update alone: 5.59 ms) -- no overlap at all between the two stream sets (HIP multiplexes streams onto 4 hardware
queues), so micro-batching was not built."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from repo_amd.algorithms.repo import RePo
from repo_amd.common.utils import set_gpu_mode
set_gpu_mode(True)
def mk(B):
    torch.manual_seed(0)
    a = RePo(bench.config("repo", B), bench.Env(6), bench.Env(6), bench.NullLogger())
    b = tuple(torch.from_numpy(x).cuda() for x in bench.synthetic_batch(1234, B, 6))
    return a, b
def run(agents, n):
    for _ in range(5):
        for a, b in agents: a.update(b, join=False)
    for a, _ in agents: a.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for a, b in agents: a.update(b, join=False)
    for a, _ in agents: a.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("one agent  B=50: %.3f ms per update" % run([mk(50)], 40))
print("two agents B=25 each, interleaved enqueue: %.3f ms per PAIR of updates" % run([mk(25), mk(25)], 40))
print("one agent  B=25: %.3f ms per update" % run([mk(25)], 40))
