#!/usr/bin/env python3
"""Benchmark of the hot path: RePo world-model + imagination updates per second.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): algo=repo, B=50 sequences per GPU,
L=50, H=15, A=6, 64x64x3 uint8 frames, synthetic replay batch from RandomState(1234) already
resident in HBM, parameters at torch default init under torch.manual_seed(0), fp32 arithmetic.
One step = one train_dynamics + one train_actor_critic (all four optimiser steps included,
noise generated inside the timed region).  N > 1 (launched by torch.distributed.run, one rank
per GPU): data parallel over batch rows, weak scaling (B=50 per GPU, global batch 50*N), RCCL
all-reduce of the flat gradient buffers.

Rank 0 prints ONE JSON line.  `value` = B=50-equivalent updates per second of the whole job
(N * K / t).  `roofline` is for the dominant kernel (the fp32-MFMA implicit-GEMM engine on its
largest launch, the decoder's 64->32 transposed convolution): algorithmic FLOPs of that launch
divided by its average duration measured here with HIP events on the launch stream.
`cpu_baseline` is the CPU oracle (PyTorch fp32 restatement of the reference) timed on this
box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, L, H, A = 50, 50, 15, 6
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
FLOP_PER_UPDATE = 740.4e9      # SURVEY.md 8d, autograd-counted on the reference at B=50 L=50 H=15


class Space:
    def __init__(self, shape):
        self.shape = shape


class Env:
    observation_space = Space((3, 64, 64))
    action_space = Space((A,))


class NullLogger:
    dir = "/tmp"

    def record(self, k, v, exclude=None):
        pass

    def dump(self, step=None):
        pass


def config(algo="repo"):
    from types import SimpleNamespace

    # defaults of experiments/train_repo.py:8-76 (hot-path keys)
    return SimpleNamespace(
        algo=algo, pixel_obs=True, embedding_size=1024, hidden_size=200, belief_size=200, state_size=30,
        dense_activation_function="elu", cnn_activation_function="relu", batch_size=B, chunk_size=L, horizon=H,
        gamma=0.99, gae_lambda=0.95, action_noise=0.0, action_ent_coef=3e-4, latent_ent_coef=0.0, free_nats=3,
        model_lr=3e-4, actor_lr=8e-5, value_lr=8e-5, grad_clip_norm=100.0, target_kl=3.0, beta_lr=1e-4,
        init_beta=1e-5, prior_train_steps=5, disag_model=False, inv_dynamics=False, disag_coef=0.0,
        replay_size=8, train_steps=1, prefill=0, load_checkpoint=False, load_offline=False, save_buffer=False,
    )


def synthetic_batch(seed=1234):
    rs = np.random.RandomState(seed)
    obs = rs.randint(0, 256, (L, B, 3, 64, 64)).astype(np.uint8)
    actions = rs.uniform(-1, 1, (L, B, A)).astype(np.float32)
    rewards = rs.uniform(0, 1, (L, B, 1)).astype(np.float32)
    dones = (rs.uniform(size=(L, B, 1)) < 1 / 500).astype(np.float32)
    return obs, actions, rewards, dones


def dominant_kernel_roofline(iters=20):
    """Decoder conv3 (64x13x13 -> 32x30x30, k6 s2) as launched inside the update: the largest
    single launch (61.05 GFLOP; dconv_up_kernel<GDec3>, all four parity classes in one launch), timed
    with HIP events on the launch stream.  `traffic` is that launch's HBM bytes from the rocprofv3 PMC
    passes committed in profiles/r01_pmc_direct_conv.txt (FETCH_SIZE x2 for the gfx950 wide-read
    under-report + WRITE_SIZE, per launch of the same nimg=2450 shape)."""
    from repo_amd import ops

    nimg = (L - 1) * B
    dev = torch.device("cuda")
    small = torch.randn(nimg, 64, 13, 13, device=dev).relu_()
    w = torch.randn(64, 32, 6, 6, device=dev) * 0.05
    bias = torch.randn(32, device=dev)
    out = torch.empty(nimg, 32, 30, 30, device=dev)
    for _ in range(3):
        ops.conv_up(ops.DEC3, small, w, bias, epi=ops.EPI_RELU, out=out)
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        ops.conv_up(ops.DEC3, small, w, bias, epi=ops.EPI_RELU, out=out)
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flop = 2.0 * nimg * 169 * 64 * 32 * 36  # every (input pixel, cin, cout, tap) MAC once
    achieved = flop / (ms * 1e-3) / 1e12
    return {
        "bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": 4.37e8,
        "kernel": "dconv_up_kernel<Geo<32,64,30,6>, UTile<128,2>> (decoder conv3 forward)",
        "ms_per_launch": round(ms, 4), "flop_per_launch": flop, "mfma_pipe_busy_pmc": 0.795,
    }


def _cpu_baseline_worker(q, nb, threads):
    """Child process: the CPU oracle on the first `nb` sequences of the synthetic batch."""
    import time as _t

    import torch as _torch

    _torch.set_num_threads(threads)
    from oracle import fixtures as fx
    from oracle.repo_oracle import OracleAgent

    obs, act, rew, done = synthetic_batch(1234)
    batch = (obs[:, :nb], act[:, :nb], rew[:, :nb], done[:, :nb])
    cfg = fx.default_config(algo="repo", batch_size=nb, chunk_size=L, horizon=H)
    agent = OracleAgent(cfg, A, seed=7)
    noise = fx.make_noise(L, nb, H, A, seed=1)
    agent.update(*batch, noise)  # warm-up (thread pools, oneDNN primitive caches)
    t0 = _t.perf_counter()
    agent.update(*batch, noise)
    q.put(_t.perf_counter() - t0)


def cpu_baseline(nb=10, threads=None, timeout_s=150.0):
    """The CPU oracle (PyTorch fp32 restatement of the reference update) on a BOUNDED sample:
    the first `nb` of the 50 sequences of the same synthetic batch, full L and H, 1 warm-up +
    1 timed update, in a child process that is killed after `timeout_s`.  Reported in
    B=50-equivalent updates/s (= nb/50 / seconds); every term of the update is linear in B."""
    import multiprocessing as mp

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = threads or max(1, min(avail, 32))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_baseline_worker, args=(q, nb, threads))
    p.start()
    p.join(timeout_s)
    if p.is_alive():
        p.kill()
        p.join()
        return {"value": None, "unit": "updates/s", "cores": threads, "kind": "port",
                "sample": f"timed out after {timeout_s:.0f} s on {nb}/{B} sequences"}
    try:
        dt = q.get(timeout=5)
    except Exception:
        return {"value": None, "unit": "updates/s", "cores": threads, "kind": "port",
                "sample": f"oracle child exited with code {p.exitcode} before reporting"}
    return {
        "value": round((nb / B) / dt, 5), "unit": "updates/s", "cores": threads, "kind": "port",
        "sample": f"1 timed update after 1 warm-up on {nb} of the {B} sequences (L={L}, H={H}), {dt:.2f} s, "
                  f"scaled by {nb}/{B}; PyTorch {torch.__version__} CPU, {threads} threads of {avail} available cores",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--algo", default="repo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--join", action="store_true", help="join the two update lanes after every update (no overlap)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: ONE global batch of 50 sequences sharded over the ranks (7,7,6,...) "
                         "instead of 50 per GPU; not the contract's default")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs a HIP device"

    from repo_amd.algorithms.repo import Dreamer, RePo
    from repo_amd.common.utils import set_gpu_mode

    set_gpu_mode(True, local_rank)
    dev = torch.device("cuda", local_rank)
    dp = None
    # REPO_FORCE_DP=1: take the RCCL path with a single rank too (exercises process-group init and the
    # collectives on the update's lane streams on a 1-GPU box)
    if world > 1 or os.environ.get("REPO_FORCE_DP") == "1":
        import torch.distributed as dist

        from repo_amd.parallel import DataParallel

        dist.init_process_group("nccl", device_id=dev)
        dp = DataParallel(dist.group.WORLD)

    torch.manual_seed(0)
    agent = (RePo if args.algo == "repo" else Dreamer)(config(args.algo), Env(), Env(), NullLogger())
    if dp is not None:
        dp.attach(agent)
    if args.strong:
        from repo_amd.parallel import shard_rows

        lo, hi = shard_rows(B, world, rank)
        host = tuple(np.ascontiguousarray(x[:, lo:hi]) for x in synthetic_batch(1234))
    else:
        host = synthetic_batch(1234 + rank)  # each rank holds its own B=50 shard of the global batch
    batch = tuple(torch.from_numpy(x).to(dev) for x in host)

    # same call pattern as Dreamer.train_agent()'s loop: update(join=False) lets the world-model
    # half of update k+1 overlap the actor-critic half of update k; the timed region is closed by
    # joining both lanes + a device synchronize, so every one of the K updates is complete
    for _ in range(args.warmup):
        agent.update(batch, join=args.join)
    agent.synchronize()
    if dp is not None:
        dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agent.update(batch, join=args.join)
    agent.synchronize()
    torch.cuda.synchronize()
    if dp is not None:
        dp.barrier()
    dt = time.perf_counter() - t0
    if dp is not None:
        dt = dp.max_float(dt)

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = (1 if args.strong else world) * args.steps / dt
        line = {
            "metric": "world-model+imagine updates/sec (B=50,L=50,64x64x3)",
            "value": round(value, 3),
            "unit": "updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"algo={args.algo} dmc_distracted-walker-walk shapes: B=50/GPU L=50 H=15 A=6 64x64x3 uint8, "
                            "one update = train_dynamics + train_actor_critic incl. 4 optimiser steps",
                "global_batch": B if args.strong else B * world, "per_gpu_batch": (B / world) if args.strong else B,
                "seq_len": L, "horizon": H,
                "parallelism": f"dp{world}", "sequences_per_s": round(value * B, 2),
                "algorithmic_tflops": round(FLOP_PER_UPDATE * value / 1e12, 2),
                "frac_of_fp32_mfma_peak_all_gpus": round(FLOP_PER_UPDATE * value / 1e12 / (FP32_MFMA_PEAK_TFLOPS * world), 4),
            },
            "last_scalars": {k: round(float(v), 6) for k, v in agent.last_scalars.items()},
        }
        line["roofline"] = dominant_kernel_roofline()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dp is not None:
        dp.barrier()
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
