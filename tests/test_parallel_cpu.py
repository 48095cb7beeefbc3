"""world_size-2 gloo tests (CPU) of the data-parallel exchange in repo_amd/parallel.py: uneven row shards +
SUM all-reduces of the flat gradients scaled by local/global rows -- the model gradient as two asynchronous
buckets, actor + critic as one -- reproduce the full-batch update of all three optimisers and the dual variable;
attach() broadcasts; global_count() is consistent across ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from repo_amd.parallel import DataParallel, shard_rows  # noqa: E402


def test_shard_rows_uneven():
    sizes = [shard_rows(50, 8, r) for r in range(8)]
    assert [b - a for a, b in sizes] == [7, 7, 6, 6, 6, 6, 6, 6]
    assert sizes[0][0] == 0 and sizes[-1][1] == 50
    assert all(sizes[i][1] == sizes[i + 1][0] for i in range(7))
    assert [shard_rows(3, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FlatOpt:
    """The three buffers DataParallel.attach() broadcasts for each FlatAdam (host tensors here)."""

    def __init__(self, seed, n):
        g = torch.Generator().manual_seed(seed)
        self.flat, self.exp_avg, self.exp_avg_sq = (torch.randn(n, generator=g) for _ in range(3))


class _FakeAgent:
    """Stands in for repo_amd's agent in attach(): same attribute names, CPU tensors."""

    def __init__(self, rank):
        self.device = torch.device("cpu")
        self.model_optimizer, self.actor_optimizer, self.value_optimizer = (_FlatOpt(10 * rank + i, 13 + i) for i in range(3))
        self.log_beta = torch.tensor(float(rank + 1))
        self.beta_optimizer = type("B", (), {})()
        self.beta_optimizer.exp_avg = torch.full((1,), float(rank + 2))
        self.beta_optimizer.exp_avg_sq = torch.full((1,), float(rank + 3))
        self.dp = None


def _shard_noise(noise, L, B, H, A, lo, hi):
    T, N = L - 1, (L - 1) * B
    rows = np.arange(T * B).reshape(T, B)[:, lo:hi].reshape(-1)  # imagination starts t*B+b of this shard
    return {
        "obs_prior": noise["obs_prior"][:, lo:hi], "obs_post": noise["obs_post"][:, lo:hi],
        "img_act": noise["img_act"][:, rows], "img_prior": noise["img_prior"][:, rows],
        "entropy": noise["entropy"].reshape(100, H - 1, N, A)[:, :, rows].reshape(100, (H - 1) * len(rows), A),
    }


def _apply_step(params, grads, flat, opt, max_norm):
    """Global-norm clip + Adam on the exchanged flat gradient.  Returns the pre-clip global norm."""
    from oracle.repo_oracle import clip_grad_norm

    off = 0
    for p_, g in zip(params, grads):
        p_.grad = flat[off:off + g.numel()].view_as(g).clone()
        off += g.numel()
    total = clip_grad_norm(params, max_norm)
    opt.step()
    return float(total)


def _exchange_buckets(dp, flat, buckets, drop):
    """The product's exchange of one flat gradient buffer (repo_amd/algorithms/repo/dreamer.py:
    _model_bucket_begin / _model_step): one asynchronous SUM all-reduce per bucket, begun in the order the
    backward finishes the buckets, all joined before the clip.  `buckets` = [(name, lo, hi)], `drop` names one
    to leave out.  With nothing dropped the result must equal ONE all-reduce of the whole buffer, bit for bit."""
    single = flat.clone()
    works = [dp.all_reduce_begin(flat[lo:hi]) for name, lo, hi in buckets if name != drop]
    dp.all_reduce_end(works)
    if drop is None or drop not in [b[0] for b in buckets]:
        dp.all_reduce(single)
        assert torch.equal(single, flat), "bucketed exchange differs from the single-bucket exchange"
    return flat


def _dp_oracle_update(agent, dp, shard, noise, nb, B, drop=None):
    """One full RePo update of a row shard with the product's exchange points: the model gradient as two
    buckets (decoder + reward head first, then encoder + RSSM), the KL sum for the dual step, actor and critic
    gradients as ONE bucket, and the prefix all-reduce of the logged sums.  `drop` names an exchange to leave out."""
    c = agent.c
    w = nb / B
    t = {k: torch.as_tensor(np.ascontiguousarray(v)) for k, v in noise.items()}
    obs, act, rew, done = (torch.as_tensor(np.ascontiguousarray(x)) for x in shard)
    beliefs, post, scal = agent.train_dynamics(obs, act, rew, 1 - done, t["obs_prior"], t["obs_post"], apply=False)
    grads = {}
    mg = agent.last["model_grads"]
    flat = torch.cat([g.reshape(-1) for g in mg]) * w
    # the product cuts between the RSSM's and the decoder's parameters; the oracle keeps the same order
    # (encoder, RSSM, decoder, reward head), so the cut is the size of the first two groups
    n_head = sum(t_.numel() for mod in ("encoder", "transition_model") for t_ in agent.p[mod].values())
    assert 0 < n_head < flat.numel()
    _exchange_buckets(dp, flat, [("model_tail", n_head, flat.numel()), ("model_head", 0, n_head)], drop)
    norms = [_apply_step(agent.model_params, mg, flat, agent.model_opt, c.grad_clip_norm)]
    grads["model"] = flat.numpy().copy()
    kl = torch.tensor([scal["train/kl_div"] * nb], dtype=torch.float64)
    if drop != "kl":
        dp.all_reduce(kl)
    kl_div = float(kl.item()) / B
    agent.log_beta.grad = torch.tensor(-(kl_div - c.target_kl), dtype=torch.float32)
    agent.beta_opt.step()
    ac = agent.train_actor_critic(beliefs.flatten(0, 1), post.flatten(0, 1), t["img_act"], t["img_prior"], t["entropy"],
                                  apply=False)
    ag, vg = agent.last["actor_grads"], agent.last["value_grads"]
    flat = torch.cat([g.reshape(-1) for g in ag + vg]) * w   # Dreamer._ac_grad: [actor | critic]
    _exchange_buckets(dp, flat, [("actor_critic", 0, flat.numel())], drop)
    na = sum(g.numel() for g in ag)
    norms.append(_apply_step(agent.actor_params, ag, flat[:na], agent.actor_opt, c.grad_clip_norm))
    norms.append(_apply_step(agent.value_params, vg, flat[na:], agent.value_opt, c.grad_clip_norm))
    grads["actor"], grads["value"] = flat[:na].numpy().copy(), flat[na:].numpy().copy()
    # logged scalars: per-rank partial sums in the prefix, already-global norms behind it
    keys = ["train/obs_loss", "train/reward_loss", "train/kl_div"]
    sums = [scal[k] * nb for k in keys] + [ac[k] * nb for k in sorted(ac)]
    buf = torch.tensor(sums + norms, dtype=torch.float64)
    if drop != "scalars":
        dp.all_reduce_prefix(buf, len(sums))
    means = (buf[:len(sums)] / B).tolist()
    out = dict(zip(keys + sorted(ac), means))
    out["norms"] = buf[len(sums):].tolist()
    out["log_beta"] = float(agent.log_beta.detach())
    out["grads"] = grads
    out["kl_for_dual"] = kl_div   # the dual step's input (Adam's first step only shows its sign in log_beta)
    return out


def _flat(ps):
    return torch.cat([p_.detach().reshape(-1) for p_ in ps])


def _worker(rank, world, port, L, B, H, A, drop, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import fixtures as fx
    from oracle.repo_oracle import OracleAgent

    dp = DataParallel()
    # attach(): rank 0's parameters / optimiser state / dual variable win on every rank
    fake = _FakeAgent(rank)
    dp.attach(fake)
    ref = _FakeAgent(0)
    for name in ("model_optimizer", "actor_optimizer", "value_optimizer"):
        for buf in ("flat", "exp_avg", "exp_avg_sq"):
            assert torch.equal(getattr(getattr(fake, name), buf), getattr(getattr(ref, name), buf)), (name, buf)
    assert float(fake.log_beta) == 1.0 and float(fake.beta_optimizer.exp_avg) == 2.0 and fake.dp is dp
    assert float(fake.beta_optimizer.exp_avg_sq) == 3.0

    obs, act, rew, done = fx.make_batch(L, B, A, seed=21)
    noise = fx.make_noise(L, B, H, A, seed=22)
    lo, hi = shard_rows(B, world, rank)
    nb = hi - lo
    T = L - 1
    # uneven shards (3 and 2 rows): the first call gathers, later calls (any multiple) are communication-free
    assert dp.global_count(T * nb) == T * B
    assert dp.shard_counts == [T * (b - a) for a, b in (shard_rows(B, world, r) for r in range(world))]
    assert dp.global_count(nb) == B and dp.global_count(T * nb) == T * B
    assert abs(dp.max_float(float(rank)) - (world - 1)) < 1e-12

    cfg = fx.default_config(algo="repo", batch_size=nb, chunk_size=L, horizon=H)
    agent = OracleAgent(cfg, A, seed=7)
    shard = (fx.preprocess_u8(obs[:, lo:hi]), act[:, lo:hi], rew[:, lo:hi], done[:, lo:hi])
    res = _dp_oracle_update(agent, dp, shard, _shard_noise(noise, L, B, H, A, lo, hi), nb, B, drop)
    res["model"], res["actor"], res["value"] = (_flat(x).numpy() for x in (agent.model_params, agent.actor_params,
                                                                             agent.value_params))
    out.put((rank, res))
    dp.barrier()
    dist.destroy_process_group()


def _run_world(L, B, H, A, world, drop=None):
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, L, B, H, A, drop, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get() for _ in range(world))
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return got


def _full_batch(L, B, H, A):
    from oracle import fixtures as fx
    from oracle.repo_oracle import OracleAgent

    obs, act, rew, done = fx.make_batch(L, B, A, seed=21)
    noise = fx.make_noise(L, B, H, A, seed=22)
    cfg = fx.default_config(algo="repo", batch_size=B, chunk_size=L, horizon=H)
    agent = OracleAgent(cfg, A, seed=7)
    scal = agent.update(obs, act, rew, done, noise)[2]
    scal["log_beta"] = float(agent.log_beta.detach())
    scal["norms"] = [agent.last["model_total_norm"], agent.last["actor_total_norm"], agent.last["value_total_norm"]]
    scal["model"], scal["actor"], scal["value"] = (_flat(x).numpy() for x in (agent.model_params, agent.actor_params,
                                                                              agent.value_params))
    scal["grads"] = {k: torch.cat([g.reshape(-1) for g in agent.last[k + "_grads"]]).numpy() for k in ("model", "actor", "value")}
    return scal


SCALARS = ["train/obs_loss", "train/reward_loss", "train/kl_div", "train/actor_loss", "train/value_loss",
           "train/action_entropy", "train/latent_entropy"]


def _mismatches(got, want):
    bad = []
    for k in SCALARS:
        if abs(got[k] - want[k]) > 1e-5 * abs(want[k]) + 1e-7:
            bad.append(k)
    for i, name in enumerate(("model", "actor", "value")):
        if abs(got["norms"][i] - want["norms"][i]) > 1e-4 * want["norms"][i]:
            bad.append("norm/" + name)
        # the exchanged (pre-clip) gradient itself, normwise
        g, w = got["grads"][name].astype(np.float64), want["grads"][name].astype(np.float64)
        if np.linalg.norm(g - w) > 1e-5 * np.linalg.norm(w):
            bad.append("params/" + name)
            continue
        # parameters after the step: Adam's first step is lr*sign(g)-like, so a gradient entry within rounding of
        # zero may land on the other side (|diff| up to 2 lr); everything else must agree to fp32 rounding
        d = np.abs(got[name] - want[name])
        if d.max() > 7e-4 or np.mean(d > 2e-6) > 2e-3:
            bad.append("params/" + name)
    if abs(got["log_beta"] - want["log_beta"]) > 1e-7 or abs(got["kl_for_dual"] - want["train/kl_div"]) > 1e-5 * abs(
            want["train/kl_div"]):
        bad.append("log_beta")
    return bad


@pytest.mark.timeout(900)
def test_dp_uneven_shards_reproduce_full_batch_update():
    """world_size 2 over gloo, B=5 split 3/2: a whole RePo update (model, dual, actor, critic) through
    repo_amd.parallel.DataParallel equals the full-batch update, and the replicas stay bit-identical."""
    L, B, H, A, world = 5, 5, 3, 6, 2
    want = _full_batch(L, B, H, A)
    got = _run_world(L, B, H, A, world)
    for r in range(world):
        assert _mismatches(got[r], want) == [], (r, _mismatches(got[r], want))
    for name in ("model", "actor", "value"):
        assert np.array_equal(got[0][name], got[1][name]), name
    assert got[0]["log_beta"] == got[1]["log_beta"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("drop,expect", [("model_tail", "params/model"), ("model_head", "params/model"),
                                         ("actor_critic", "params/actor"), ("actor_critic", "params/value"),
                                         ("kl", "log_beta"), ("scalars", "train/obs_loss")])
def test_dp_dropping_any_exchange_is_detected(drop, expect):
    """The comparison above is not vacuous: leaving out any one of the exchanges (either model bucket, the
    actor+critic bucket, the KL sum, the logged sums) breaks it."""
    L, B, H, A, world = 5, 5, 3, 6, 2
    want = _full_batch(L, B, H, A)
    got = _run_world(L, B, H, A, world, drop=drop)
    assert expect in _mismatches(got[0], want), (drop, _mismatches(got[0], want))


def _fault_worker(rank, world, port, faulty_rank, out):
    """The status protocol of a faulted update (Dreamer._take_status / _raise_update_fault) over gloo: a scan timeout
    on ONE rank must make EVERY rank skip its steps and raise in the same update, and the collectives behind it must
    still line up (nobody is left waiting in the next gradient all-reduce)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from repo_amd import ops
    from repo_amd._lib import RepoHipError

    dp = DataParallel()
    cpu = torch.device("cpu")
    events = []
    for update, fault in enumerate((None, faulty_rank, None)):
        # ... the scans of this update ran; the column-split engine ORed a bit into THIS rank's sticky word
        if fault == rank:
            ops.scan_status(cpu).fill_(2)
        st = ops.take_scan_status(cpu, dp)                 # the update's own copy, MAX-reduced; sticky word cleared
        assert int(ops.scan_status(cpu).item()) == 0
        grad = torch.full((5,), float(rank + 1))
        dp.all_reduce(grad)                                # the gradient exchange every rank still takes part in
        stepped = int(st.item()) == 0                      # what repo_clip_adam's skip_if_nonzero decides on the device
        try:
            ops.raise_scan_status(int(st.item()), "the update's optimiser steps were skipped")
            events.append((update, "ok", stepped, float(grad[0])))
        except RepoHipError as e:
            events.append((update, "raised", stepped, "reverse" in str(e)))
    dp.barrier()
    out.put((rank, events))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("faulty_rank", [0, 1])
def test_dp_rank_local_scan_fault_raises_on_every_rank_in_the_same_update(faulty_rank):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_fault_worker, args=(r, world, port, faulty_rank, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get() for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0          # neither rank hung or died
    want = [(0, "ok", True, 3.0), (1, "raised", False, True), (2, "ok", True, 3.0)]
    assert got[0] == want and got[1] == want, got
