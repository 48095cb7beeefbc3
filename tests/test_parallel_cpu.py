"""world_size-2 gloo test (CPU) of the data-parallel exchange: row shards + SUM all-reduce of
flat gradients scaled by local/global rows reproduce the full-batch gradient and the dual step."""
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


def _worker(rank, world, port, L, B, H, A, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import fixtures as fx
    from oracle.repo_oracle import OracleAgent

    dp = DataParallel()
    obs, act, rew, done = fx.make_batch(L, B, A, seed=21)
    noise = fx.make_noise(L, B, H, A, seed=22)
    a, b = shard_rows(B, world, rank)
    nb = b - a
    assert dp.global_count(nb) == B and dp.global_count(nb) == B  # second call is cached
    cfg = fx.default_config(algo="repo", batch_size=nb, chunk_size=L, horizon=H)
    agent = OracleAgent(cfg, A, seed=7)
    o = torch.from_numpy(fx.preprocess_u8(obs[:, a:b]))
    agent.train_dynamics(o, torch.from_numpy(act[:, a:b]), torch.from_numpy(rew[:, a:b]),
                         1 - torch.from_numpy(done[:, a:b]), torch.from_numpy(noise["obs_prior"][:, a:b]),
                         torch.from_numpy(noise["obs_post"][:, a:b]), apply=False)
    # local mean-loss gradients -> sum-of-sums: scale by local/global rows, then SUM all-reduce
    flat = torch.cat([g.reshape(-1) for g in agent.last["model_grads"]]) * (nb / B)
    dp.all_reduce(flat)
    kl = torch.tensor([agent.last_kl_div * nb]) if hasattr(agent, "last_kl_div") else torch.zeros(1)
    dp.all_reduce(kl)
    if rank == 0:
        out.put((flat.numpy(), float(kl.item())))
    dp.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp_gradients_match_full_batch():
    L, B, H, A, world = 5, 3, 3, 6, 2
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, L, B, H, A, out)) for r in range(world)]
    for p in procs:
        p.start()
    flat, _ = out.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    from oracle import fixtures as fx
    from oracle.repo_oracle import OracleAgent

    obs, act, rew, done = fx.make_batch(L, B, A, seed=21)
    noise = fx.make_noise(L, B, H, A, seed=22)
    cfg = fx.default_config(algo="repo", batch_size=B, chunk_size=L, horizon=H)
    agent = OracleAgent(cfg, A, seed=7)
    agent.train_dynamics(torch.from_numpy(fx.preprocess_u8(obs)), torch.from_numpy(act), torch.from_numpy(rew),
                         1 - torch.from_numpy(done), torch.from_numpy(noise["obs_prior"]),
                         torch.from_numpy(noise["obs_post"]), apply=False)
    want = torch.cat([g.reshape(-1) for g in agent.last["model_grads"]]).numpy()
    err = np.linalg.norm(flat - want) / np.linalg.norm(want)
    assert err < 1e-5, err
