"""End-to-end train_agent() loop (host sampling -> pinned -> H2D -> pipelined updates) vs resident-batch bench."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from repo_amd.algorithms.repo.repo import RePo
cfg = bench.config("repo")
agent = RePo(cfg, bench.Env(), bench.Env(), bench.NullLogger())
N = 20000
buf = type(agent.buffer)(N, (3, 64, 64), (6,), obs_type=np.uint8)
rs = np.random.RandomState(0)
buf.observations[:] = rs.randint(0, 256, size=buf.observations.shape, dtype=np.uint8)
buf.actions[:] = rs.uniform(-1, 1, buf.actions.shape)
buf.rewards[:] = rs.uniform(0, 1, buf.rewards.shape)
buf.dones[:] = 0
buf.pos, buf.full = 0, True
if os.environ.get('REPO_MIRROR', '1') == '1':
    buf.enable_device_mirror(agent.device)
buf.invalidate_mirror()
agent.buffer = buf
# host gather alone
t0 = time.perf_counter()
for _ in range(5):
    h = buf.prefetch(cfg.batch_size, cfg.chunk_size, agent.device); buf.release(h, cfg.batch_size, cfg.chunk_size, agent.device)
torch.cuda.synchronize()
print(f"prefetch (host gather + H2D enqueue): {(time.perf_counter()-t0)/5*1e3:.2f} ms per batch")
cfg.train_steps = 10
agent.train_agent(); torch.cuda.synchronize()
cfg.train_steps = 40
t0 = time.perf_counter()
agent.train_agent(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"train_agent: {dt/40*1e3:.2f} ms per update ({40/dt:.1f} updates/s) incl. sampling + H2D")
