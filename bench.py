#!/usr/bin/env python3
"""Benchmark of the hot path: RePo world-model + imagination updates per second.

    python bench.py --gpus N --steps K --warmup W [--config c2|c4|c5|c4x128|tia|mt]

Workload (default `--config c2` = BASELINE.json configs[1]): algo=repo, B=50
sequences per GPU, L=50, H=15, A=6, 64x64x3 uint8 frames, parameters at torch default init under
torch.manual_seed(0), fp32 arithmetic.  One step = ONE iteration of the reference's train_agent() loop body
(dreamer.py:385-401): draw a fresh batch from the replay ring, train_dynamics, train_actor_critic -- all
four optimiser steps and the noise generation inside the timed region.  The synthetic replay ring
(RandomState(1234+rank)) is mirrored in HBM before the timed region starts (repo_amd/common/buffers.py:
the sampler draws indices on the host exactly like the reference, the 30.7 MB batch is gathered on the
device), so `value` is the rate with inputs resident in HBM; the same K steps on ONE resident batch are
timed afterwards and reported as `resident_batch_ms` (what round 1's line measured).

N > 1: data parallel over batch rows, RCCL all-reduce of the flat gradient buffers (model gradient in two
buckets overlapped with the encoder backward, actor + critic in one).  `value` is WEAK scaling (B=50 per GPU,
global batch 50*N); the same line carries a `strong` object: ONE global batch of 50 sequences dealt 7,7,6,...
over the ranks (BASELINE's "updates/sec (B=50 ...) at 1/2/4/8"), each rank sampling its shard from its ring,
timed the same way (barrier + synchronize on both sides, max over ranks), plus the per-rank time spent inside
the all-reduces (`allreduce_ms`, HIP events around the collectives of the weak run).  `python bench.py --gpus N`
starts its own ranks (one child process per GPU via torch.distributed.run, before this process touches the
GPU); when an external launcher already set RANK/WORLD_SIZE it runs as that rank.  Rank 0 prints ONE JSON
line; `n_gpus` is the number of ranks the process group saw.

`roofline` is for the dominant kernel = the FIRST ROW BY TOTAL TIME of the committed rocprofv3 --kernel-trace --stats
summary of this same command (profiles/dominant_kernel_rocprof.json, written from profiles/rNN_bench_kernel_stats_pipelined.csv
by tools/layers_in_update.py): algorithmic FLOPs of one launch / its average duration measured live with HIP events on its
launch stream INSIDE the timed updates; `peak` is the peak of the pipe the kernel EXECUTES on (157.3 TFLOP/s for the
fp32-MFMA kernels; 2516.6 / 6 = 419.4 "fp32-equivalent" TFLOP/s for the bf16x6 kernels, which form every fp32 product
from six bf16 products), `frac_of_fp32_mfma_peak` the same launch against the fp32 peak, `kernel_ms_rocprof` /
`frac_at_rocprof_duration` the trace's own average for that row, `kernel_time_sum_ms` the trace's kernel time per update.
`roofline.largest_launch` is the same object for the update's largest single product (decoder conv3 forward, 61 GFLOP).
`traffic` comes from the rocprofv3 --pmc summaries committed under profiles/ (named in `traffic_source`), never from this run.
`cpu_baseline` is the CPU oracle (PyTorch fp32 restatement of the reference) timed on this box's host
cores on the full 50-sequence batch, 2 warm-up + 3 timed updates (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC between the ranks' processes (RCCL); read when the HIP runtime starts, so set before the first GPU call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2516.6  # "Peak BF16/FP16 MFMA ~2.5 PF dense" = 16 x the fp32 MFMA rate (1024 vs 64 FLOP/clk/SIMD)
L, H = 50, 15
# per-config: (algo, B, A, label, frame size); FLOPs scale with rows (SURVEY.md 8d scaling law)
CONFIGS = {
    "c2": ("repo", 50, 6, "dmc_distracted-walker-walk shapes (BASELINE configs[1])", 64),
    "c4": ("repo", 32, 7, "maniskill-PushCubeMatterport shapes at the reference's 64x64 frames, A=7 (BASELINE configs[3] pin)", 64),
    "c5": ("dreamer", 50, 6, "algo=dreamer on dmc_distracted-walker-walk shapes (BASELINE configs[4])", 64),
    # BASELINE configs[3] at its own 128x128 frames: the reference cannot run it (its encoder flatten and decoder are
    # 64x64 only), the conv stack is build-defined and its parity is pinned by the oracle only
    # f4 widening: the reference's TIA (algorithms/repo/tia.py) on configs[1]'s shapes: two filters, three decoders
    "tia": ("tia", 50, 6, "algo=tia (tia.py: distractor filter, masked pair of decoders, distractor-only decoder) on "
            "dmc_distracted-walker-walk shapes", 64),
    # f4 widening: the reference's MultitaskRePo (repo_mt.py; 3 tasks like every multitask environment of the reference:
    # environments/__init__.py:121-147) on configs[1]'s shapes.  FLOPs: the c2 figure (FiLM is elementwise, the task
    # one-hot adds 3 K columns to five dense layers)
    "mt": ("repo_multitask", 50, 6, "algo=repo_multitask (repo_mt.py: FiLM-conditioned conv stacks, task-conditioned RSSM / "
           "heads / rollout, per-task beta; 3 tasks) on dmc_distracted-walker-walk shapes", 64),
    "c4x128": ("repo", 32, 7, "maniskill-PushCubeMatterport shapes at 128x128 frames through the BUILD-DEFINED 128x128 "
               "conv stack (no reference model exists for it), A=7", 128),
}
FLOP_PER_UPDATE_B50 = 740.4e9  # SURVEY.md 8d, autograd-counted on the reference at B=50 L=50 H=15 A=6
# 128x128 stack: extra conv/fc FLOPs per decoded/encoded frame over the 64x64 stack, forward + data gradient + weight
# gradient (2*Cout*Hout^2*Cin*k^2 per layer; encoder 29.4 -> 179.2 MFLOP incl. the 9216x1024 fc, decoder 48.4 -> 77.0
# MFLOP forward; the first encoder layer has no data gradient)
EXTRA_FLOP_PER_FRAME_128 = 3 * (179.2e6 - 29.4e6 + 77.0e6 - 48.4e6) - (12.19e6 - 2.95e6)


# TIA: two more decoder passes (48.4 MFLOP forward per frame each, x3 with both gradients) and one more observe scan
# (0.556 MMAC per row-step forward, x3) per frame over Dreamer's update
EXTRA_FLOP_PER_FRAME_TIA = 2 * 3 * 48.4e6 + 3 * 2 * 0.556e6


def flop_per_update(B, image=64, algo="repo"):
    """Algorithmic FLOPs of one update of B sequences (L=50, H=15)."""
    return (FLOP_PER_UPDATE_B50 * B / 50.0 + (EXTRA_FLOP_PER_FRAME_128 * (L - 1) * B if image == 128 else 0.0)
            + (EXTRA_FLOP_PER_FRAME_TIA * (L - 1) * B if algo == "tia" else 0.0))
N_TASKS = 3                    # --config mt
RING_FRAMES = 6000             # synthetic replay ring per rank (72 MB of frames; 120 windows of 50)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "dominant_kernel_pmc.json")


class Space:
    def __init__(self, shape):
        self.shape = shape


class Env:
    def __init__(self, A=6, image=64, num_tasks=None):
        self.observation_space = Space((3, image, image))
        self.action_space = Space((A,))
        if num_tasks:
            self.num_tasks = num_tasks


class NullLogger:
    dir = "/tmp"

    def record(self, k, v, exclude=None):
        pass

    def dump(self, step=None):
        pass


def config(algo="repo", B=50):
    from types import SimpleNamespace

    # defaults of experiments/train_repo.py:8-76 (hot-path keys)
    return SimpleNamespace(
        algo=algo, pixel_obs=True, embedding_size=1024, hidden_size=200, belief_size=200, state_size=30,
        dense_activation_function="elu", cnn_activation_function="relu", batch_size=B, chunk_size=L, horizon=H,
        gamma=0.99, gae_lambda=0.95, action_noise=0.0, action_ent_coef=3e-4, latent_ent_coef=0.0, free_nats=3,
        model_lr=3e-4, actor_lr=8e-5, value_lr=8e-5, grad_clip_norm=100.0, target_kl=3.0, beta_lr=1e-4,
        init_beta=1e-5, prior_train_steps=5, disag_model=False, inv_dynamics=False, disag_coef=0.0,
        tia_obs_coef=1.0, tia_adv_coef=1.0, tia_reward_train_steps=1, share_repr=False,
        replay_size=8, train_steps=1, prefill=0, load_checkpoint=False, load_offline=False, save_buffer=False,
    )


def synthetic_batch(seed=1234, B=50, A=6, image=64, num_tasks=0):
    rs = np.random.RandomState(seed)
    obs = rs.randint(0, 256, (L, B, 3, image, image)).astype(np.uint8)
    actions = rs.uniform(-1, 1, (L, B, A)).astype(np.float32)
    rewards = rs.uniform(0, 1, (L, B, 1)).astype(np.float32)
    dones = (rs.uniform(size=(L, B, 1)) < 1 / 500).astype(np.float32)
    if num_tasks:   # multitask batches lead with the task one-hots (one task per sequence)
        tasks = np.eye(num_tasks, dtype=np.float32)[np.broadcast_to(rs.randint(0, num_tasks, B)[None], (L, B))]
        return np.ascontiguousarray(tasks), obs, actions, rewards, dones
    return obs, actions, rewards, dones


def synthetic_ring(buffer_cls, seed, A, device, image=64, num_tasks=0):
    """A full replay ring of RING_FRAMES synthetic transitions (same distributions as synthetic_batch:
    uniform u8 frames, uniform actions / rewards, episode ends with probability 1/500), mirrored in HBM."""
    rs = np.random.RandomState(seed)
    if num_tasks:
        ring = buffer_cls(RING_FRAMES, num_tasks, (3, image, image), (A,), obs_type=np.uint8)
        ring.tasks[:] = np.eye(num_tasks, dtype=np.float32)[np.repeat(rs.randint(0, num_tasks, RING_FRAMES // 500 + 1), 500)[:RING_FRAMES]]
    else:
        ring = buffer_cls(RING_FRAMES, (3, image, image), (A,), obs_type=np.uint8)
    ring.observations[:] = rs.randint(0, 256, size=ring.observations.shape, dtype=np.uint8)
    ring.actions[:] = rs.uniform(-1, 1, ring.actions.shape)
    ring.rewards[:] = rs.uniform(0, 1, ring.rewards.shape)
    ring.dones[:] = rs.uniform(size=ring.dones.shape) < 1 / 500
    ring.pos, ring.full = 0, True
    ring.enable_device_mirror(device)
    ring.invalidate_mirror()
    return ring


# (CB, CS, HB, KS) of the conv layers (repo_amd.ops.CONV_GEO; repeated here so that the table below needs no import)
_CONV_GEO = {(3, 32, 64, 4): 0, (32, 64, 31, 4): 1, (64, 128, 14, 4): 2, (128, 256, 6, 4): 3, (64, 128, 13, 5): 4,
             (32, 64, 30, 6): 5, (3, 32, 64, 6): 6}
_CONV_NAMES = ["encoder conv1", "encoder conv2", "encoder conv3", "encoder conv4", "decoder conv2", "decoder conv3",
               "decoder conv4"]
_SCAN_MAC, _IMG_MAC, _MLP4_MAC = 556e3 - 204.8e3, 467.6e3, 230 * 200 + 3 * 200 * 200   # per row and step


def kernel_spec(name, nimg, horizon_rows):
    """What bench.py needs to time one kernel of the update live and to price it: the repo_amd.ops entry point whose call
    launches it (+ the conv layer id to filter on), algorithmic FLOPs of one launch, the matrix pipe it executes on.
    `name` is a rocprofv3 kernel name (any prefix / template spelling)."""
    import re

    spec = _kernel_spec(name, nimg, horizon_rows)
    if spec is not None:
        spec["kernel_name"] = name
    return spec


def _kernel_spec(name, nimg, horizon_rows):
    import re

    m = re.search(r"Geo<(\d+), ?(\d+), ?(\d+), ?(\d+)>", name)
    geo = tuple(int(x) for x in m.groups()) if m else None
    layer = _CONV_GEO.get(geo)
    if "tconv_up_kernel" in name:   # encoder conv2's data gradient in gather form (csrc/tconv_up.h): no Geo<> in its name
        return {"op": "conv_up", "layer": 1, "pipe": "bf16x6", "flop": 2.0 * nimg * 64 * 14 * 14 * 32 * 16,
                "label": "tconv_up_kernel (encoder conv2 data gradient, gather form)"}
    if "tconv_down_kernel" in name:   # staging + multiplying waves (csrc/tconv_down.h): TcdGeoT<KS, WB, WS, ..>, no Geo<>
        t = re.search(r"TcdGeoT<(\d+), ?(\d+), ?(\d+)", name)
        ks, _, ws = (int(x) for x in t.groups()) if t else (6, 30, 13)
        lay, what = (1, "encoder conv2 forward") if ks == 4 else (5, "decoder conv3 data gradient")
        return {"op": "conv_down", "layer": lay, "pipe": "bf16x6", "flop": 2.0 * nimg * 64 * ws * ws * 32 * ks * ks,
                "label": f"tconv_down_kernel ({what}, staging + multiplying waves)"}
    conv = {"buconv_scatter_kernel": ("conv_up", "bf16x6"), "uconv_scatter_kernel": ("conv_up", "fp32"),
            "bconv_down_kernel": ("conv_down", "bf16x6"), "dconv_down_kernel": ("conv_down", "fp32"),
            "tconv_wgrad_kernel": ("conv_wgrad", "bf16x6"), "bconv_wgrad_kernel": ("conv_wgrad", "bf16x6"),
            "dconv_wgrad_kernel": ("conv_wgrad", "fp32")}
    for k, (op, pipe) in conv.items():
        if k in name and layer is not None:
            cb, cs, hb, ks = geo
            hs = (hb - ks) // 2 + 1
            enc = layer < 4
            role = {"conv_up": "data gradient" if enc else "forward", "conv_down": "forward" if enc else "data gradient",
                    "conv_wgrad": "weight gradient"}[op]
            note = " (+ its slab-reduce launch inside the bracket)" if op == "conv_wgrad" else ""
            return {"op": op, "layer": layer, "pipe": pipe, "flop": 2.0 * nimg * cs * hs * hs * cb * ks * ks,
                    "label": f"{k}<Geo<{cb},{cs},{hb},{ks}>> ({_CONV_NAMES[layer]} {role}){note}"}
    other = {"bdec4_nll_kernel": ("decoder_out_nll", "bf16x6", 2.0 * nimg * 32 * 900 * 3 * 36, "decoder conv4 forward + pixel NLL"),
             "dconv_dec4_nll_kernel": ("decoder_out_nll", "fp32", 2.0 * nimg * 32 * 900 * 3 * 36, "decoder conv4 forward + pixel NLL"),
             "imagine32_fwd_kernel": ("rssm_imagine_fwd", "bf16x6", 2.0 * horizon_rows * _IMG_MAC, "imagination rollout forward"),
             "imagine32_bwd_kernel": ("rssm_imagine_bwd", "bf16x6", 2.0 * horizon_rows * (_IMG_MAC - _MLP4_MAC), "imagination rollout reverse"),
             "observe_cs_fwd_kernel": ("rssm_observe_fwd", "fp32", 2.0 * nimg * _SCAN_MAC, "observe scan forward (latency chain; incl. its hoisted GEMM launches)"),
             "observe_cs_bwd_kernel": ("rssm_observe_bwd", "fp32", 4.0 * nimg * _SCAN_MAC, "observe scan reverse (latency chain; incl. its deferred GEMM launches)")}
    for k, (op, pipe, flop, what) in other.items():
        if k in name:
            return {"op": op, "layer": None, "pipe": pipe, "flop": flop, "label": f"{k} ({what})"}
    return None


PIPE_PEAK = {"fp32": FP32_MFMA_PEAK_TFLOPS, "bf16x6": BF16_MFMA_PEAK_TFLOPS / 6.0}
PIPE_NOTE = {"fp32": "fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)",
             "bf16x6": "bf16 MFMA, every fp32 product as six exact bf16 products: peak = dense bf16 peak / 6, in fp32-equivalent FLOPs"}


class LaunchTimer:
    """HIP-event pairs around every call of one repo_amd.ops entry point (optionally one conv layer of it), recorded on
    the launch stream while the timed updates run; keeps the arguments of the first call for the isolated replay."""

    def __init__(self, spec):
        self.spec, self.pairs, self.first = spec, [], None

    def __enter__(self):
        from repo_amd import functional as Fn
        from repo_amd import ops

        self._ops, self._orig = ops, getattr(ops, self.spec["op"])
        timer = self

        def timed(*a, **k):
            if timer.spec["layer"] is not None and a[0] != timer.spec["layer"]:
                return timer._orig(*a, **k)
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            out = timer._orig(*a, **k)
            e1.record(s)
            timer.pairs.append((e0, e1))
            if timer.first is None:
                timer.first = (a, k)
            return out

        setattr(ops, self.spec["op"], timed)
        assert Fn.ops is ops
        return self

    def __exit__(self, *exc):
        setattr(self._ops, self.spec["op"], self._orig)
        return False

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / max(len(self.pairs), 1)

    def isolated_ms(self, iters=20):
        """The same call (the first timed one's arguments) alone on an idle GPU."""
        if self.first is None:
            return None
        a, k = self.first
        torch.cuda.synchronize()
        for _ in range(3):
            self._orig(*a, **k)
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            self._orig(*a, **k)
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / iters


class AllReduceTimer:
    """HIP events around every gradient all-reduce of this rank during the timed updates: the blocking ones are
    bracketed on the calling stream; a bucket begun asynchronously is timed from its begin to the point the
    calling stream has waited for it (so the figure is the exchange's span, part of which overlaps the encoder
    backward)."""

    def __init__(self, dp):
        self.dp, self.pairs = dp, []
        self._orig = (dp.all_reduce, dp.all_reduce_begin, dp.all_reduce_end)
        self._orig_small = (dp.all_reduce_status, dp.all_reduce_prefix)
        t = self

        def small(which, label):
            # the 4-byte MAX all-reduce of the update's status word and the logged scalars' prefix sum: latency-only
            # collectives, one each per update, in line on the update's streams
            def call(*a, **k):
                s = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                out = t._orig_small[which](*a, **k)
                e1.record(s)
                t.pairs.append((label, e0, e1))
                return out
            return call

        dp.all_reduce_status, dp.all_reduce_prefix = small(0, "status_word_max_4B"), small(1, "logged_scalars_prefix")

        def all_reduce(x):
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            out = t._orig[0](x)
            e1.record(s)
            t.pairs.append((x.numel() * 4, e0, e1))
            return out

        def begin(x, stream=None):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            w = t._orig[1](x, stream=stream)
            return (w, x.numel() * 4, e0)

        def end(works):
            t._orig[2]([w for w, _, _ in works])
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            for _, nbytes, e0 in works:
                t.pairs.append((nbytes, e0, e1))

        dp.all_reduce, dp.all_reduce_begin, dp.all_reduce_end = all_reduce, begin, end

    def stop(self):
        self.dp.all_reduce, self.dp.all_reduce_begin, self.dp.all_reduce_end = self._orig
        self.dp.all_reduce_status, self.dp.all_reduce_prefix = self._orig_small

    def summary(self, steps):
        torch.cuda.synchronize()
        by = {}
        for nbytes, e0, e1 in self.pairs:
            by.setdefault(nbytes, []).append(e0.elapsed_time(e1))
        return {"rank": 0, "per_update_by_bucket_bytes": {str(k): round(sum(v) / steps, 4) for k, v in sorted(by.items(), key=lambda kv: str(kv[0]))},
                "note": "rank 0; span from issue to the consumer stream's join, summed per update; gradient buckets by "
                        "their bytes, the two latency-only collectives (status word MAX, logged-scalar sums) by name"}


def under_profiler():
    """rocprofv3 preloads its tool library into the profiled process: the isolated re-runs of the timed kernels are
    skipped then, so that their rows in the kernel trace hold the launches INSIDE the updates only."""
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "") + os.environ.get("HSA_TOOLS_LIB", "")
    return "rocprof" in pre or "ROCPROFILER_REGISTER_FORCE_LOAD" in os.environ or os.environ.get("REPO_BENCH_NO_ISOLATED") == "1"


ROCPROF_SUMMARY = os.path.join(ROOT, "profiles", "dominant_kernel_rocprof.json")
LARGEST_LAUNCH = "buconv_scatter_kernel<Geo<32, 64, 30, 6>, BSConf<Geo<32, 64, 30, 6>, 1, 4> >"   # decoder conv3 forward: the update's largest single product (full template spelling: the committed counters are matched on it)


def trace_summary():
    """profiles/dominant_kernel_rocprof.json: the committed kernel trace's top row by total time, the largest launch's
    row and the kernel time per update (tools/layers_in_update.py --json)."""
    if os.path.exists(ROCPROF_SUMMARY):
        return json.load(open(ROCPROF_SUMMARY))
    return {}


def roofline_specs(nimg, horizon_rows):
    """(top-by-time spec, largest-launch spec): which kernels this run times live."""
    tr = trace_summary()
    top_name = (tr.get("top_by_time") or {}).get("name") or LARGEST_LAUNCH
    top = kernel_spec(top_name, nimg, horizon_rows) or kernel_spec(LARGEST_LAUNCH, nimg, horizon_rows)
    big = kernel_spec(LARGEST_LAUNCH, nimg, horizon_rows)
    return top, big


def _roofline_obj(timer, row, nimg):
    spec = timer.spec
    flop, pipe = spec["flop"], spec["pipe"]
    peak = PIPE_PEAK[pipe]
    ms = timer.mean_ms()
    achieved = flop / (ms * 1e-3) / 1e12 if ms else 0.0
    out = {
        "bound": "mfma", "kernel": spec["label"], "pipe": PIPE_NOTE[pipe],
        "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
        "traffic": None, "ms_per_launch": round(ms, 4), "launches_timed": len(timer.pairs), "flop_per_launch": flop,
        "timing": "HIP events on the launch stream inside the timed updates (other streams of the update run beside "
                  "it); achieved = flop_per_launch / ms_per_launch (ALGORITHMIC fp32 FLOPs), frac = achieved / peak",
    }
    if not under_profiler():
        iso = timer.isolated_ms()
        if iso:
            out["isolated_ms_per_launch"] = round(iso, 4)
            out["isolated_frac"] = round(flop / (iso * 1e-3) / 1e12 / peak, 4)
    if row and row.get("nimg") == nimg:
        out["kernel_ms_rocprof"] = row.get("avg_ms_in_update")
        if row.get("avg_ms_in_update"):
            out["frac_at_rocprof_duration"] = round(flop / (row["avg_ms_in_update"] * 1e-3) / 1e12 / peak, 4)
        out["share_of_kernel_time"] = row.get("share_of_kernel_time")
    if os.path.exists(PMC_SUMMARY):
        pmc = json.load(open(PMC_SUMMARY)).get("kernels", {})
        # the counters of THIS instantiation only: the full template spelling must agree (a name prefix or the Geo<>
        # tuple alone would also match another tile / wave configuration of the same kernel)
        want = _norm_kernel(spec.get("kernel_name", ""))
        key = next((k for k in pmc if _norm_kernel(k) == want), None)
        if key and pmc[key].get("nimg") == nimg:
            out["traffic"] = pmc[key].get("traffic_bytes_per_launch")
            out["traffic_source"] = "profiles/dominant_kernel_pmc.json <- " + str(pmc[key].get("source"))
            out["mfma_pipe_busy_pmc"] = pmc[key].get("mfma_pipe_busy")
    return out


def _norm_kernel(name):
    """A rocprofv3 kernel name without return type, namespace, argument list and blanks."""
    import re

    name = re.sub(r"^void\s+", "", name.strip())
    name = re.sub(r"\(.*$", "", name)          # the demangled argument list, if any
    return name.replace("repo::", "").replace(" ", "")


def roofline(top_timer, big_timer, nimg):
    """`roofline` of the JSON line: the trace's top kernel by total time, measured live; `largest_launch` nested."""
    tr = trace_summary()
    out = _roofline_obj(top_timer, tr.get("top_by_time"), nimg)
    out["selected_by"] = ("first row by total time of " + str(tr.get("csv")) if tr.get("top_by_time")
                          else "no committed trace summary: the largest launch")
    if tr.get("nimg") == nimg or (tr.get("top_by_time") or {}).get("nimg") == nimg:
        out["kernel_time_sum_ms"] = tr.get("kernel_time_sum_ms_per_update")
        out["kernel_time_source"] = tr.get("source")
    if big_timer is not top_timer:
        out["largest_launch"] = _roofline_obj(big_timer, tr.get("largest_launch"), nimg)
    return out


def _cpu_baseline_worker(q, threads, warm, timed, B, A, algo, image=64):
    """Child process: the CPU oracle on the full synthetic batch."""
    import time as _t

    import torch as _torch

    _torch.set_num_threads(threads)
    from oracle import fixtures as fx
    from oracle.repo_oracle import OracleAgent

    batch = synthetic_batch(1234, B, A, image, N_TASKS if algo.endswith("multitask") else 0)
    cfg = fx.default_config(algo=algo, batch_size=B, chunk_size=L, horizon=H, share_repr=False)
    if algo == "tia":
        from oracle.repo_oracle import OracleTIA

        agent = OracleTIA(cfg, A, seed=7)
    elif algo.endswith("multitask"):
        from oracle.repo_oracle import OracleMultitask

        agent = OracleMultitask(cfg, A, N_TASKS, seed=7)
    else:
        agent = OracleAgent(cfg, A, seed=7, image=image)
    noise = fx.make_noise(L, B, H, A, seed=1, tia=algo == "tia")
    for _ in range(warm):
        agent.update(*batch, noise)  # thread pools, oneDNN primitive caches
    t0 = _t.perf_counter()
    for _ in range(timed):
        agent.update(*batch, noise)
    q.put((_t.perf_counter() - t0) / timed)


def cpu_baseline(threads=None, warm=2, timed=3, timeout_s=240.0, B=50, A=6, algo="repo", image=64):
    """the CPU oracle (PyTorch fp32 restatement of the reference update, validated against the
    reference's goldens) on the SAME full workload -- B=50, L=50, H=15 -- 2 warm-up + 3 timed updates on
    this box's host cores, in a child process that is killed after `timeout_s` (any --config: its B, A, algo)."""
    import multiprocessing as mp

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = threads or max(1, min(avail, 32))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_baseline_worker, args=(q, threads, warm, timed, B, A, algo, image))
    p.start()
    p.join(timeout_s)
    base = {"value": None, "unit": "updates/s", "cores": threads, "kind": "port"}
    if p.is_alive():
        p.kill()
        p.join()
        return {**base, "sample": f"timed out after {timeout_s:.0f} s ({warm}+{timed} full-batch updates)"}
    try:
        dt = q.get(timeout=5)
    except Exception:
        return {**base, "sample": f"oracle child exited with code {p.exitcode} before reporting"}
    return {
        **base, "value": round(1.0 / dt, 5),
        "sample": f"{timed} timed updates after {warm} warm-up on the full batch (algo={algo}, B={B}, L={L}, H={H}, A={A}, {image}x{image} frames), "
                  f"{dt:.2f} s per update; PyTorch {torch.__version__} CPU, {threads} threads of {avail} available cores",
    }


# ----------------------------------------------------------------------------- self-launch (N > 1)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """Start one child process per GPU with torch.distributed.run and hand back its exit code.  This process
    has not touched the GPU (no HIP call has been made), and it never replaces itself: the ranks are children."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(world, rank):
    """--rendezvous-only: the launcher / process-group plumbing without the workload (gloo when there is no
    GPU), so the self-launch path is testable on a CPU-only box."""
    import torch.distributed as dist

    dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    t = torch.ones(1, device="cuda" if torch.cuda.is_available() else "cpu")
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"rendezvous_ranks": int(t.item()), "world_size": dist.get_world_size()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--algo", default=None, help="override the config's algorithm (repo | dreamer | tia | repo_multitask | dreamer_multitask)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--join", action="store_true", help="join the two update lanes after every update (no overlap)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: ONE global batch of B sequences sharded over the ranks (7,7,6,...) "
                         "instead of B per GPU; resident batch only; not the contract's default")
    ap.add_argument("--batch", type=int, default=None,
                    help="override the config's batch size (e.g. 7 = one rank's shard of the strong-scaling job at 8 "
                         "GPUs, to time its compute alone); not the contract's default, the line says so")
    ap.add_argument("--rendezvous-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.rendezvous_only:
        return rendezvous_only(world, rank)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"

    from repo_amd.algorithms.repo import TIA, Dreamer, MultitaskDreamer, MultitaskRePo, RePo
    from repo_amd.common.buffers import MultitaskSequenceReplayBuffer, SequenceReplayBuffer
    from repo_amd.common.utils import set_gpu_mode

    algo, B, A, label, image = CONFIGS[args.config]
    algo = args.algo or algo
    if args.batch:
        B, label = args.batch, label + f" with the batch overridden to {args.batch} (--batch)"
    set_gpu_mode(True, local_rank)
    dev = torch.device("cuda", local_rank)
    dp = None
    # REPO_FORCE_DP=1: take the RCCL path with a single rank too (exercises process-group init and the
    # collectives on the update's lane streams on a 1-GPU box)
    if world > 1 or os.environ.get("REPO_FORCE_DP") == "1":
        import torch.distributed as dist

        from repo_amd.parallel import DataParallel

        if "RANK" not in os.environ:  # REPO_FORCE_DP=1 without a launcher: a one-rank group of our own
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                              MASTER_PORT=str(_free_port()))
        dist.init_process_group("nccl", device_id=dev)
        dp = DataParallel(dist.group.WORLD)

    torch.manual_seed(0)
    cfg = config(algo, B)
    ntasks = N_TASKS if algo.endswith("multitask") else 0
    agent = {"repo": RePo, "dreamer": Dreamer, "tia": TIA, "repo_multitask": MultitaskRePo,
             "dreamer_multitask": MultitaskDreamer}[algo](cfg, Env(A, image, ntasks), Env(A, image, ntasks), NullLogger())
    if dp is not None:
        dp.attach(agent)
    if args.strong:
        from repo_amd.parallel import shard_rows

        lo, hi = shard_rows(B, world, rank)
        host = tuple(np.ascontiguousarray(x[:, lo:hi]) for x in synthetic_batch(1234, B, A, image, ntasks))
        cfg.batch_size = hi - lo
    else:
        host = synthetic_batch(1234 + rank, B, A, image, ntasks)  # each rank holds its own B-sequence shard of the global batch
    resident = tuple(torch.from_numpy(x).to(dev) for x in host)
    Bl = cfg.batch_size

    def run_resident(k):
        # same call pattern as Dreamer.train_agent()'s loop: update(join=False) lets the world-model half of
        # update k+1 overlap the actor-critic half of update k
        for _ in range(k):
            agent.update(resident, join=args.join)
        agent.synchronize()

    np.random.seed(4321 + rank)  # the sampler's RNG (np.random.choice, like the reference)
    agent.buffer = synthetic_ring(MultitaskSequenceReplayBuffer if ntasks else SequenceReplayBuffer, 1234 + rank, A, dev,
                                  image, ntasks)

    def run_from_ring(k):
        # train_agent()'s loop body K times: fresh indices on the host, device gather, pipelined update
        ring = agent.buffer
        h = ring.prefetch(Bl, L, dev)
        for i in range(k):
            batch = ring.acquire(h, Bl, L, dev)
            cur = h
            agent.update(batch, join=args.join)
            with torch.cuda.stream(agent._wm_stream):
                ring.release(cur, Bl, L, dev)
            if i + 1 < k:
                h = ring.prefetch(Bl, L, dev)
        agent.synchronize()

    def timed(fn, k):
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(k)
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
        dt = time.perf_counter() - t0
        return dp.max_float(dt) if dp is not None else dt

    main_loop = run_resident if args.strong else run_from_ring
    main_loop(args.warmup)
    from repo_amd import ops

    ar = AllReduceTimer(dp) if dp is not None else None
    top_spec, big_spec = roofline_specs((L - 1) * Bl, (H - 1) * (L - 1) * Bl)
    timer = LaunchTimer(top_spec)
    big_timer = timer if big_spec["label"] == top_spec["label"] else LaunchTimer(big_spec)
    import contextlib

    with contextlib.ExitStack() as stack:
        stack.enter_context(timer)
        if big_timer is not timer:
            stack.enter_context(big_timer)
        dt = timed(main_loop, args.steps)
    if ar is not None:
        ar.stop()
    run_resident(2)
    dt_res = timed(run_resident, args.steps)

    # north_star's literal input path ("sequence sampler ... pinned-memory hipMemcpyAsync'd to device"): the same loop
    # with the HBM mirror off -- the batch is gathered on the host by repo_host_gather_rows (a few threads, into a
    # page-locked slot) and copied on a side stream while the previous update computes
    dt_pin = None
    if world == 1 and not args.strong:
        ring = agent.buffer
        mirror, ring._mirror = ring._mirror, None
        ring.__dict__.pop("_stage", None)
        run_from_ring(3)
        dt_pin = timed(run_from_ring, args.steps)
        ring._mirror = mirror
        ring.__dict__.pop("_stage", None)

    # N > 1: the same line also carries STRONG scaling -- one global batch of B sequences dealt 7,7,6,... over
    # the ranks, every rank sampling its shard from its own ring (a collective-free re-shard: reset_counts()
    # on all ranks, then the first global_count() of the next update gathers the new shard sizes)
    strong = None
    # REPO_BENCH_FORCE_STRONG=1 (with REPO_FORCE_DP=1): walk the re-shard path with one rank (the whole batch is its shard)
    if dp is not None and (world > 1 or os.environ.get("REPO_BENCH_FORCE_STRONG") == "1") and not args.strong:
        from repo_amd.parallel import shard_rows

        lo, hi = shard_rows(B, world, rank)
        if hi > lo:
            Bl_weak = Bl
            Bl = cfg.batch_size = hi - lo
            agent.synchronize()
            dp.reset_counts()
            run_from_ring(max(3, args.warmup // 2))
            dt_s = timed(run_from_ring, args.steps)
            strong = {"value": round(args.steps / dt_s, 3), "unit": "updates/s", "ms_per_step": round(dt_s / args.steps * 1e3, 3),
                      "global_batch": B, "shards": [b - a for a, b in (shard_rows(B, world, r) for r in range(world))],
                      "algorithmic_tflops": round(flop_per_update(B, image, algo) * args.steps / dt_s / 1e12, 2)}
            Bl = cfg.batch_size = Bl_weak
            dp.reset_counts()
        else:
            strong = {"value": None, "note": f"B={B} < {world} ranks: an empty shard"}

    if rank == 0:
        ms = dt / args.steps * 1e3
        nranks = dp.world_size if dp is not None else 1
        value = (1 if args.strong else nranks) * args.steps / dt
        # FLOPs of one update of the GLOBAL batch this line's `value` counts: B sequences (--strong: one global
        # batch sharded over the ranks) or B per rank (weak)
        flop = flop_per_update(B, image, algo)
        line = {
            "metric": "world-model+imagine updates/sec (B=50,L=50,64x64x3)" if not args.config.startswith("c4") else
                      f"world-model+imagine updates/sec (B=32,L=50,{image}x{image}x3,A=7)",
            "value": round(value, 3),
            "unit": "updates/s",
            "n_gpus": nranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "fp32 inputs, outputs and accumulation, the reference's precision; the MFMA-bound conv / dense "
                          "kernels (csrc/bgemm.h, bconv.h, buconv.h, bwgrad.h) and the imagination rollout (csrc/imagine32.hip) form each fp32 product EXACTLY-split as six bf16 x bf16 "
                          "partial products on the bf16 matrix pipe -- measured error at or below the fp32-MFMA kernels' "
                          "(tests/test_ops_gpu.py::test_bf16x6_*, test_bgemm_*); nothing is rounded to bf16",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: algo={algo} {label}: B={B}/GPU L={L} H={H} A={A} {image}x{image}x3 uint8; one step = "
                            "sample a fresh batch from the HBM-mirrored replay ring + train_dynamics + "
                            "train_actor_critic incl. 4 optimiser steps" + (" (resident batch: --strong)" if args.strong else ""),
                "global_batch": B if args.strong else B * nranks, "per_gpu_batch": Bl,
                "seq_len": L, "horizon": H, "ring_frames": RING_FRAMES,
                "parallelism": f"dp{nranks}", "sequences_per_s": round(value * B, 2),
                "algorithmic_tflops": round(flop * value / 1e12, 2),
                "frac_of_fp32_mfma_peak_all_gpus": round(flop * value / 1e12 / FP32_MFMA_PEAK_TFLOPS / nranks, 4),
            },
            "resident_batch_ms": round(dt_res / args.steps * 1e3, 3),
            "pinned_path_ms": round(dt_pin / args.steps * 1e3, 3) if dt_pin is not None else None,
            "last_scalars": {k: round(float(v), 6) for k, v in agent.last_scalars.items()},
        }
        line["roofline"] = roofline(timer, big_timer, (L - 1) * Bl)
        if strong is not None:
            line["strong"] = strong
        if ar is not None:
            line["allreduce_ms"] = ar.summary(args.steps)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B=B, A=A, algo=algo, image=image)
        print(json.dumps(line), flush=True)
    if dp is not None:
        dp.barrier()
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
