"""GPU tests of the rows either side of the update: the pinned-memory replay
sampler, the acting path, checkpoint round trips, Dreamer (config 5) at full size, and the
data-parallel hooks driven on one GPU."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import repo_oracle as ro
from tests.test_update_gpu import Env, Logger, dev_batch, dev_noise, make_agent
from tests.util import log

pytestmark = pytest.mark.gpu


def test_replay_buffer_pinned_prefetch_matches_host_sample():
    from repo_amd.common.buffers import SequenceReplayBuffer

    rs = np.random.RandomState(0)
    buf = SequenceReplayBuffer(300, (3, 64, 64), (6,), obs_type=np.uint8)
    for i in range(450):  # wraps: exercises the head rotation of sample()
        buf.push(rs.randint(0, 256, (3, 64, 64)).astype(np.uint8), rs.uniform(-1, 1, 6), float(i), i % 50 == 49)
    dev = torch.device("cuda")
    B, L = 5, 12
    np.random.seed(3)
    want = buf.sample(B, L)
    np.random.seed(3)
    h = buf.prefetch(B, L, dev)
    got = buf.acquire(h, B, L, dev)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert tuple(g.shape) == w.shape
        assert np.array_equal(g.cpu().numpy(), w)
    assert got[0].dtype == torch.uint8
    # sequences never straddle the write head: rewards are consecutive integers along time
    r = got[2].cpu().numpy()[:, :, 0]
    assert np.all(np.diff(r, axis=0) == 1)
    # double buffering: a second prefetch while the first is in use lands in the other slot
    buf.release(h, B, L, dev)
    np.random.seed(4)
    h2 = buf.prefetch(B, L, dev)
    assert h2 != h
    got2 = buf.acquire(h2, B, L, dev)
    np.random.seed(4)
    want2 = buf.sample(B, L)
    torch.cuda.synchronize()
    assert np.array_equal(got2[0].cpu().numpy(), want2[0])
    assert np.array_equal(got[0].cpu().numpy(), want[0])  # first slot untouched


def test_device_mirror_batches_equal_host_sampling():
    """enable_device_mirror: batches gathered on the GPU from the device copy of the ring equal the
    reference's host-side sample() for the same np.random state -- after the initial upload, after
    incremental pushes (partial flush), across the ring's wrap-around and after load()."""
    from repo_amd.common.buffers import SequenceReplayBuffer

    dev = torch.device("cuda", 0)
    B, L, cap = 5, 7, 60
    buf = SequenceReplayBuffer(cap, (3, 64, 64), (6,), obs_type=np.uint8)
    buf.enable_device_mirror(dev)
    rs = np.random.RandomState(11)

    def push(n):
        for _ in range(n):
            buf.push(rs.randint(0, 256, (3, 64, 64)).astype(np.uint8), rs.uniform(-1, 1, 6), rs.uniform(), rs.uniform() < 0.1)

    def check(seed):
        np.random.seed(seed)
        want = buf.sample(B, L)
        np.random.seed(seed)
        h = buf.prefetch(B, L, dev)
        got = buf.acquire(h, B, L, dev)
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert tuple(g.shape) == w.shape
            assert np.array_equal(g.cpu().numpy(), w.astype(g.cpu().numpy().dtype))
        buf.release(h, B, L, dev)

    push(25)
    check(1)          # first use: full upload of what is stored
    push(9)
    check(2)          # incremental top-up
    check(3)          # nothing new
    push(40)          # wraps the ring (74 > 60): two-piece top-up
    check(4)
    push(70)          # more than a full ring since the last batch: full re-upload
    check(5)
    buf.observations[3] = 7
    buf.invalidate_mirror()
    check(6)


def test_train_agent_runs_from_buffer():
    agent, cfg = make_agent("repo", 8, 4, 5, 6)
    cfg.train_steps = 3
    rs = np.random.RandomState(1)
    agent.buffer = type(agent.buffer)(200, (3, 64, 64), (6,), obs_type=np.uint8)
    for i in range(120):
        agent.buffer.push(rs.randint(0, 256, (3, 64, 64)).astype(np.uint8), rs.uniform(-1, 1, 6), rs.uniform(), i % 40 == 39)
    agent.train_agent()
    torch.cuda.synchronize()
    assert agent.model_optimizer.step_count == 3 and agent.beta_optimizer.step_count == 3
    assert all(math.isfinite(v) for v in agent.last_scalars.values())


def test_acting_path_matches_oracle():
    """update_latent_and_select_action (dreamer.py:175-196): one filtering step + rsample / mode."""
    A = 6
    agent, cfg = make_agent("repo", 8, 4, 5, A)
    o = ro.OracleAgent(cfg, A, seed=7)
    rs = np.random.RandomState(5)
    obs_u8 = rs.randint(0, 256, (1, 3, 64, 64)).astype(np.uint8)
    obs = torch.from_numpy(fx.preprocess_u8(obs_u8))
    belief = torch.from_numpy(rs.standard_normal((1, 200)).astype(np.float32) * 0.3)
    state = torch.from_numpy(rs.standard_normal((1, 30)).astype(np.float32))
    action = torch.from_numpy(rs.uniform(-1, 1, (1, A)).astype(np.float32))
    e1 = torch.from_numpy(rs.standard_normal((1, 1, 30)).astype(np.float32))
    e2 = torch.from_numpy(rs.standard_normal((1, 1, 30)).astype(np.float32))
    ea = torch.from_numpy(rs.standard_normal((1, A)).astype(np.float32))
    es = torch.from_numpy(rs.standard_normal((100, 1, A)).astype(np.float32))
    with torch.no_grad():
        emb = ro.encoder_fwd(o.p["encoder"], obs)
        outs = ro.observe(o.p["transition_model"], belief, state, action[None], emb[None], torch.ones(1, 1, 1), e1, e2)
        ob, os_ = outs[0][0], outs[4][0]
        mean, std = ro.actor_fwd(o.p["actor_model"], ob, os_)
        want_explore = torch.tanh(mean + std * ea)
        ys = torch.tanh(mean + std * es)
        lp = ro.tanh_normal_log_prob(ys, mean, std)
        want_mode = ys[lp.argmax(0), torch.arange(1)]
    with torch.no_grad():
        emb_g = agent.encoder(obs.cuda())
        outs_g = agent.transition_model.observe(belief.cuda(), state.cuda(), action.cuda()[None], emb_g[None],
                                                noise=(e1.cuda(), e2.cuda()))
        b_g, s_g = outs_g[0][0], outs_g[4][0]
        a_explore = agent.actor_model.get_action(b_g, s_g, det=False, eps=ea.cuda())
        a_mode = agent.actor_model.get_action(b_g, s_g, det=True, eps=es.cuda())
    np.testing.assert_allclose(b_g.cpu().numpy(), ob.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(s_g.cpu().numpy(), os_.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(a_explore.cpu().numpy(), want_explore.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(a_mode.cpu().numpy(), want_mode.numpy(), rtol=1e-4, atol=1e-5)
    # full method (draws its own noise): shapes, range, determinism of the belief path
    b2, s2, a2 = agent.update_latent_and_select_action(belief.cuda(), state.cuda(), action.cuda(), obs.cuda(), explore=True)
    assert b2.shape == (1, 200) and s2.shape == (1, 30) and a2.shape == (1, A)
    assert float(a2.abs().max()) <= 1.0
    np.testing.assert_allclose(b2.cpu().numpy(), ob.numpy(), rtol=1e-4, atol=1e-5)


def test_checkpoint_roundtrip_and_reference_layout(tmp_path):
    agent, cfg = make_agent("repo", 8, 4, 5, 6)
    batch, _ = dev_batch(8, 4, 6, 3)
    agent.noise_source, _ = dev_noise(8, 4, 5, 6, 9)
    agent.update(batch)
    agent.logger.dir = str(tmp_path)
    agent.step = 123
    agent.save_checkpoint()
    ck = torch.load(os.path.join(str(tmp_path), "models.pt"), map_location="cpu", weights_only=False)
    # key layout of the reference's get_param_dict (dreamer.py:501-520, repo.py:114-118)
    assert set(ck) == {"step", "encoder", "transition_model", "obs_model", "reward_model", "actor_model", "value_model",
                       "model_optimizer", "actor_optimizer", "value_optimizer", "log_beta", "beta_optimizer"}
    shapes = fx.param_shapes(6)
    for mod in fx.MODULES:
        assert list(ck[mod].keys()) == list(shapes[mod].keys())
        for k, shp in shapes[mod].items():
            assert tuple(ck[mod][k].shape) == tuple(shp)
    # the optimiser state loads into a real torch.optim.Adam over same-shaped parameters
    ps = [torch.nn.Parameter(torch.zeros(tuple(s))) for m in fx.MODEL_MODULES for s in shapes[m].values()]
    torch.optim.Adam(ps, lr=1.0).load_state_dict(ck["model_optimizer"])
    # round trip: a fresh agent restored from the checkpoint continues bit-identically
    agent2, _ = make_agent("repo", 8, 4, 5, 6, seed=8)
    agent2.logger.dir = str(tmp_path)
    agent2.load_checkpoint()
    assert agent2.step == 123
    batch2, _ = dev_batch(8, 4, 6, 4)
    nz, _ = dev_noise(8, 4, 5, 6, 10)
    agent.noise_source = nz
    agent2.noise_source = nz
    agent.update(batch2)
    s1 = dict(agent.last_scalars)
    agent2.update(batch2)
    s2 = dict(agent2.last_scalars)
    assert s1 == s2, (s1, s2)


def test_dreamer_full_size_matches_oracle_scalars():
    """BASELINE config 5 (algo=dreamer, B=50 L=50 H=15): one update against the CPU oracle."""
    L, B, H, A = 50, 50, 15, 6
    agent, cfg = make_agent("dreamer", L, B, H, A)
    batch, host = dev_batch(L, B, A, 1234)
    agent.noise_source, nz = dev_noise(L, B, H, A, 77)
    agent.update(batch)
    got = dict(agent.last_scalars)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    want = ro.OracleAgent(cfg, A, seed=7).update(*host, nz)[2]
    for k, w in want.items():
        r = abs(got[k] - w) / (abs(w) + 1e-12)
        log(f"[dreamer C5 full] {k}: got {got[k]:.7g} oracle {w:.7g} rel {r:.2e}")
        assert r < 1e-3, (k, got[k], w)


def test_repo_config0_b16_matches_oracle_scalars_two_updates():
    """BASELINE config 0 shapes (B=16, L=50, H=15, full RePo): two consecutive updates against the CPU
    oracle -- every logged loss, the KL and the dual variable within 1e-3 relative (north_star)."""
    L, B, H, A = 50, 16, 15, 6
    agent, cfg = make_agent("repo", L, B, H, A)
    oracle = ro.OracleAgent(cfg, A, seed=7)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    for u in range(2):
        batch, host = dev_batch(L, B, A, 4321 + u, u8=(u == 0))
        agent.noise_source, nz = dev_noise(L, B, H, A, 88 + u)
        agent.update(batch)
        got = dict(agent.last_scalars)
        want = oracle.update(*host, nz)[2]
        for k, w in want.items():
            r = abs(got[k] - w) / (abs(w) + 1e-12)
            log(f"[repo C0 B=16 update {u}] {k}: got {got[k]:.7g} oracle {w:.7g} rel {r:.2e}")
            assert r < 1e-3, (u, k, got[k], w)


class ThreadDP:
    """Stands in for repo_amd.parallel.DataParallel with world_size 2 on ONE GPU: the two 'ranks'
    run in two host threads; all_reduce meets at a barrier and sums the two tensors."""

    def __init__(self, rank, shared):
        self.rank, self.world_size, self.sh = rank, 2, shared

    def global_count(self, n):
        return n * self.world_size

    def all_reduce(self, t):
        sh = self.sh
        sh["slot"][self.rank] = t
        torch.cuda.current_stream().synchronize()  # the peer reads t from ITS stream
        sh["barrier"].wait()
        total = sh["slot"][0] + sh["slot"][1]
        torch.cuda.current_stream().synchronize()  # before the peer overwrites its slot
        sh["barrier"].wait()
        t.copy_(total)
        return t

    def all_reduce_prefix(self, buf, n):
        self.all_reduce(buf[:n])
        return buf


def test_data_parallel_two_shards_equal_full_batch():
    """Two row shards with sum-all-reduced gradients reproduce the full-batch update (8e)."""
    import threading

    L, B, H, A = 8, 6, 5, 6
    batch, _ = dev_batch(L, B, A, 21)
    nz, _ = dev_noise(L, B, H, A, 22)
    full, _ = make_agent("repo", L, B, H, A)
    full.noise_source = nz
    full.update(batch)
    s_full = dict(full.last_scalars)
    T, N = L - 1, (L - 1) * B

    def shard_noise(lo, hi):
        nb = hi - lo
        rows = torch.arange(T * B).view(T, B)[:, lo:hi].reshape(-1).cuda()  # imagined rows t*B+b of this shard
        return {
            "obs_prior": nz["obs_prior"][:, lo:hi].contiguous(),
            "obs_post": nz["obs_post"][:, lo:hi].contiguous(),
            "img_act": nz["img_act"][:, rows].contiguous(),
            "img_prior": nz["img_prior"][:, rows].contiguous(),
            "entropy": nz["entropy"].view(100, H - 1, N, A)[:, :, rows].reshape(100, (H - 1) * T * nb, A).contiguous(),
        }

    halves = [(0, 3), (3, 6)]
    shared = {"slot": [None, None], "barrier": threading.Barrier(2)}
    agents, scal, errs = [None, None], [None, None], []
    for r in range(2):
        agents[r], _ = make_agent("repo", L, 3, H, A)
        agents[r].dp = ThreadDP(r, shared)
        agents[r].noise_source = shard_noise(*halves[r])

    def run(r):
        try:
            torch.cuda.set_device(0)
            lo, hi = halves[r]
            agents[r].update(tuple(x[:, lo:hi].contiguous() for x in batch))
            scal[r] = dict(agents[r].last_scalars)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            shared["barrier"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errs, errs
    torch.cuda.synchronize()
    for k, w in s_full.items():
        assert abs(scal[0][k] - w) <= 2e-4 * abs(w) + 1e-7, (k, scal[0][k], w)
    assert scal[0] == scal[1]
    for r in range(2):
        e = ((agents[r].model_optimizer.flat - full.model_optimizer.flat).abs().max()).item()
        ea = ((agents[r].actor_optimizer.flat - full.actor_optimizer.flat).abs().max()).item()
        log(f"[dp 2 shards] rank {r}: max |param diff| vs full batch after one update: model {e:.2e} actor {ea:.2e}")
        assert e < 2e-5 and ea < 2e-5
    assert torch.equal(agents[0].model_optimizer.flat, agents[1].model_optimizer.flat)  # replicas stay identical
    assert abs(float(agents[0].log_beta) - float(full.log_beta)) < 1e-6


def test_pipelined_updates_bitwise_equal_sequential():
    """update(join=False) overlaps WM(k+1) with AC(k) on two streams; the parameters after K such
    updates must be BIT-identical to K joined (back-to-back) updates."""
    L, B, H, A, K = 10, 6, 6, 6, 4
    agents = [make_agent("repo", L, B, H, A)[0] for _ in range(2)]
    batches = [dev_batch(L, B, A, 600 + u, u8=(u % 2 == 0))[0] for u in range(K)]
    noises = [dev_noise(L, B, H, A, 700 + u)[0] for u in range(K)]
    for mode, ag in enumerate(agents):
        for u in range(K):
            ag.noise_source = noises[u]
            ag.update(batch=batches[u], join=(mode == 0))
        ag.synchronize()
    torch.cuda.synchronize()
    a, b = agents
    for name in ("model_optimizer", "actor_optimizer", "value_optimizer"):
        pa, pb = getattr(a, name).flat, getattr(b, name).flat
        assert torch.equal(pa, pb), (name, (pa - pb).abs().max().item())
    assert a.last_scalars == b.last_scalars
    assert float(a.log_beta) == float(b.log_beta)
