import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from repo_amd.algorithms.repo import RePo
from repo_amd.common.utils import set_gpu_mode
set_gpu_mode(True)
B = 7
torch.manual_seed(0)
agent = RePo(bench.config("repo", B), bench.Env(6), bench.Env(6), bench.NullLogger())
batch = tuple(torch.from_numpy(x).cuda() for x in bench.synthetic_batch(1234, B, 6))
def run(n):
    t0 = time.perf_counter()
    for _ in range(n): agent.update(batch, join=False)
    t1 = time.perf_counter()
    agent.synchronize(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3
run(5)
print("with per-update log flush: enqueue %.2f total %.2f ms" % run(40))
orig = agent._flush_log
agent._flush_log = lambda defer=False: setattr(agent, "_log_pending", None)
run(5)
print("without the flush (no host wait on the previous update): enqueue %.2f total %.2f ms" % run(40))
