"""Long pipelined run: memory must stay flat, losses finite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
import bench
from repo_amd.algorithms.repo.repo import RePo
agent = RePo(bench.config("repo"), bench.Env(), bench.Env(), bench.NullLogger())
batch = tuple(torch.from_numpy(x).cuda() for x in bench.synthetic_batch(1234))
t0 = time.perf_counter()
for i in range(400):
    agent.update(batch, join=False)
    if i % 100 == 99:
        s = agent.last_scalars
        print(i + 1, f"{torch.cuda.memory_allocated()/2**20:.0f} MiB alloc {torch.cuda.memory_reserved()/2**20:.0f} MiB reserved",
              {k: round(v, 4) for k, v in list(s.items())[:4]}, flush=True)
        assert all(math.isfinite(v) for v in s.values())
agent.synchronize(); torch.cuda.synchronize()
print("ms/update", 1e3 * (time.perf_counter() - t0) / 400)
