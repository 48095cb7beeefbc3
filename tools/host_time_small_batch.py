"""Host enqueue time per update against the wall time at B = 7 (a strong-scaling shard) and B = 50: is the Python launch
path or the GPU the bound?  (tools/host_time2.py repeats it without the per-update log flush.)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import bench
from repo_amd.algorithms.repo import RePo
from repo_amd.common.utils import set_gpu_mode
set_gpu_mode(True)
for B in (7, 50):
    torch.manual_seed(0)
    cfg = bench.config("repo", B)
    agent = RePo(cfg, bench.Env(6), bench.Env(6), bench.NullLogger())
    batch = tuple(torch.from_numpy(x).cuda() for x in bench.synthetic_batch(1234, B, 6))
    for _ in range(5): agent.update(batch, join=False)
    agent.synchronize(); torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter(); c0 = time.process_time()
    for _ in range(n): agent.update(batch, join=False)
    t1 = time.perf_counter(); c1 = time.process_time()
    agent.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: host enqueue {(t1-t0)/n*1e3:.2f} ms/update (cpu {(c1-c0)/n*1e3:.2f}), total {(t2-t0)/n*1e3:.2f} ms/update")
