#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE on CPU in the build container.

Run here only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/gen_golden.py

Outputs small .npz fixtures next to this script.  A fixture holds only numbers
(scalars the reference logs, latent slices, gradient norms, parameter checksums);
parameters, replay batches and noise are regenerated from RandomState seeds by
oracle/fixtures.py and are not stored.

The reference draws its noise from the global torch generator
(torch.randn_like in models/rssm.py:49,61-63 and Normal.rsample ->
torch.distributions.normal._standard_normal for the actor and the 100-sample
entropy, models/utils.py:161).  Both entry points are patched below to serve the
pre-drawn RandomState noise in the reference's own draw order, and every served
shape is checked against the order documented in SURVEY.md section 8c.
"""
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
# --out DIR: write the fixtures somewhere else (tests/test_golden_recipe.py regenerates into a temp dir and compares)
OUT = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else HERE

from oracle import fixtures as fx  # noqa: E402


def import_reference():
    for name in ("wandb", "wandb.data_types"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import warnings

    warnings.filterwarnings("ignore")
    import common.utils as cu

    cu.set_gpu_mode(False)
    from algorithms.repo import TIA, Dreamer, RePo

    return Dreamer, RePo, TIA


class FakeSpace:
    def __init__(self, shape):
        self.shape = shape


class FakeEnv:
    def __init__(self, A):
        self.observation_space = FakeSpace((3, 64, 64))
        self.action_space = FakeSpace((A,))


class RecLogger:
    dir = "/tmp"

    def __init__(self):
        self.kv = OrderedDict()

    def record(self, k, v, exclude=None):
        self.kv[k] = v


class NoiseFeeder:
    """Serves pre-drawn noise through the two RNG entry points the reference uses."""

    def __init__(self):
        self.queue = []
        self.served = []

    def load(self, noise, T, H):
        q = []
        for t in range(T):
            q.append(noise["obs_prior"][t])
            q.append(noise["obs_post"][t])
        if "d_obs_prior" in noise:  # TIA: the distractor filter's scan follows the task scan (tia.py:88-121)
            for t in range(T):
                q.append(noise["d_obs_prior"][t])
                q.append(noise["d_obs_post"][t])
        for t in range(H - 1):
            q.append(noise["img_act"][t])
            q.append(noise["img_prior"][t])
        q.append(noise["entropy"])
        self.queue = q

    def pop(self, shape):
        a = self.queue.pop(0)
        assert tuple(a.shape) == tuple(shape), (a.shape, tuple(shape))
        self.served.append(tuple(shape))
        return torch.from_numpy(a.copy())


def install_patches(feeder, record):
    import torch.distributions.normal as tdn
    import torch.nn as nn

    def randn_like(x, **kw):
        return feeder.pop(x.shape)

    def std_normal(shape, dtype, device):
        return feeder.pop(shape)

    torch.randn_like = randn_like
    tdn._standard_normal = std_normal

    orig_clip = nn.utils.clip_grad_norm_

    def clip(params, max_norm, *a, **k):
        params = list(params) if not isinstance(params, torch.Tensor) else [params]
        record["clip_calls"].append(
            [None if p.grad is None else p.grad.detach().clone() for p in params]
        )
        tn = orig_clip(params, max_norm, *a, **k)
        record["total_norms"].append(float(tn))
        return tn

    nn.utils.clip_grad_norm_ = clip


def load_params(algo, params):
    for mod in fx.MODULES:
        m = getattr(algo, mod)
        sd = m.state_dict()
        assert list(sd.keys()) == list(params[mod].keys()), (mod, list(sd.keys()))
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params[mod].items()})


def module_norms(algo, grads, which):
    """L2 norm of pre-clip gradients per module for one clip_grad_norm_ call."""
    if which == "model":
        out, i = {}, 0
        for mod in fx.MODEL_MODULES:
            n = len(list(getattr(algo, mod).parameters()))
            sq = sum(float((g.double() ** 2).sum()) for g in grads[i : i + n] if g is not None)
            out[mod] = np.sqrt(sq)
            i += n
        return out
    sq = sum(float((g.double() ** 2).sum()) for g in grads if g is not None)
    return {which: np.sqrt(sq)}


def run_case(Algo, algo_name, L, B, H, A, n_updates, full_latents, feeder, record, out_path):
    cfg = fx.default_config(algo=algo_name, batch_size=B, chunk_size=L, horizon=H)
    logger = RecLogger()
    algo = Algo(cfg, FakeEnv(A), FakeEnv(A), logger)
    load_params(algo, fx.make_params(A, seed=7))
    T, N = L - 1, (L - 1) * B

    g = OrderedDict()
    g["meta"] = np.array([L, B, H, A, n_updates], dtype=np.int64)
    scalar_keys = None
    for u in range(n_updates):
        obs_u8, actions, rewards, dones = fx.make_batch(L, B, A, seed=11 + u)
        noise = fx.make_noise(L, B, H, A, seed=101 + u)
        feeder.load(noise, T, H)
        record["clip_calls"].clear()
        record["total_norms"].clear()
        logger.kv.clear()

        obs = torch.from_numpy(fx.preprocess_u8(obs_u8))
        acts = torch.from_numpy(actions)
        rews = torch.from_numpy(rewards)
        nonterms = torch.from_numpy(1 - dones)
        beliefs, post = algo.train_dynamics(obs, acts, rews, nonterms)
        algo.train_actor_critic(beliefs.flatten(0, 1), post.flatten(0, 1))
        assert not feeder.queue, "noise left over: draw order differs from SURVEY 8c"

        keys = sorted(logger.kv.keys())
        if scalar_keys is None:
            scalar_keys = keys
        assert keys == scalar_keys
        g[f"u{u}/scalars"] = np.array([logger.kv[k] for k in keys], dtype=np.float64)
        if hasattr(algo, "log_beta"):
            g[f"u{u}/log_beta"] = np.array(algo.log_beta.item(), dtype=np.float64)
        g[f"u{u}/total_norms"] = np.array(record["total_norms"], dtype=np.float64)
        mn = module_norms(algo, record["clip_calls"][0], "model")
        mn.update(module_norms(algo, record["clip_calls"][1], "actor_model"))
        mn.update(module_norms(algo, record["clip_calls"][2], "value_model"))
        g[f"u{u}/module_grad_norms"] = np.array([mn[m] for m in fx.MODULES], dtype=np.float64)
        if full_latents:
            g[f"u{u}/beliefs"] = beliefs.numpy().copy()
            g[f"u{u}/posterior_states"] = post.numpy().copy()
        else:
            g[f"u{u}/beliefs"] = beliefs.numpy()[::7, ::3, :8].copy()
            g[f"u{u}/posterior_states"] = post.numpy()[::7, ::3, :8].copy()
        print(
            f"  [{os.path.basename(out_path)}] update {u}: "
            + " ".join(f"{k.split('/')[-1]}={logger.kv[k]:.6g}" for k in keys),
            flush=True,
        )
    g["scalar_keys"] = np.array(scalar_keys)
    # per-tensor checksums after the last update
    sums, abssums, names = [], [], []
    for mod in fx.MODULES:
        for k, v in getattr(algo, mod).state_dict().items():
            names.append(f"{mod}.{k}")
            sums.append(float(v.double().sum()))
            abssums.append(float(v.double().abs().sum()))
    g["param_names"] = np.array(names)
    g["param_sums"] = np.array(sums, dtype=np.float64)
    g["param_abssums"] = np.array(abssums, dtype=np.float64)
    np.savez_compressed(out_path, **g)
    print(f"wrote {out_path} ({os.path.getsize(out_path)} bytes)")


def run_mt_case(algo_name, L, B, H, A, C, n_updates, feeder, record, out_path, **over):
    """MultitaskDreamer / MultitaskRePo (dreamer_mt.py, repo_mt.py; share_repr=False): tasks (L, B, C) lead every
    batch.  Same stored quantities as run_case; log_beta is the per-task VECTOR."""
    from algorithms.repo import MultitaskDreamer, MultitaskRePo

    Algo = MultitaskRePo if algo_name == "repo_multitask" else MultitaskDreamer
    cfg = fx.default_config(algo=algo_name, batch_size=B, chunk_size=L, horizon=H, share_repr=False, **over)
    logger = RecLogger()
    env = FakeEnv(A)
    env.num_tasks = C
    algo = Algo(cfg, env, env, logger)
    load_params(algo, fx.make_params(A, seed=7, cond=C))
    T = L - 1
    g = OrderedDict()
    g["meta"] = np.array([L, B, H, A, n_updates, C], dtype=np.int64)
    g["cfg"] = np.array([cfg.init_beta, cfg.target_kl, cfg.beta_lr], dtype=np.float64)
    scalar_keys = None
    for u in range(n_updates):
        obs_u8, actions, rewards, dones = fx.make_batch(L, B, A, seed=11 + u)
        tasks = fx.make_tasks(L, B, C, seed=11 + u)
        noise = fx.make_noise(L, B, H, A, seed=101 + u)
        feeder.load(noise, T, H)
        record["clip_calls"].clear()
        record["total_norms"].clear()
        logger.kv.clear()
        tk = torch.from_numpy(tasks)
        beliefs, post = algo.train_dynamics(tk, torch.from_numpy(fx.preprocess_u8(obs_u8)), torch.from_numpy(actions),
                                            torch.from_numpy(rewards), torch.from_numpy(1 - dones))
        algo.train_actor_critic(tk[1:].flatten(0, 1), beliefs.flatten(0, 1), post.flatten(0, 1))
        assert not feeder.queue, "noise left over: draw order differs from SURVEY 8c"
        keys = sorted(logger.kv.keys())
        scalar_keys = scalar_keys or keys
        assert keys == scalar_keys
        g[f"u{u}/scalars"] = np.array([logger.kv[k] for k in keys], dtype=np.float64)
        if hasattr(algo, "log_beta"):
            g[f"u{u}/log_beta"] = algo.log_beta.detach().numpy().astype(np.float64)
        g[f"u{u}/total_norms"] = np.array(record["total_norms"], dtype=np.float64)
        mn = module_norms(algo, record["clip_calls"][0], "model")
        mn.update(module_norms(algo, record["clip_calls"][1], "actor_model"))
        mn.update(module_norms(algo, record["clip_calls"][2], "value_model"))
        g[f"u{u}/module_grad_norms"] = np.array([mn[m] for m in fx.MODULES], dtype=np.float64)
        g[f"u{u}/beliefs"] = beliefs.numpy().copy()
        g[f"u{u}/posterior_states"] = post.numpy().copy()
        print(f"  [{os.path.basename(out_path)}] update {u}: "
              + " ".join(f"{k.split('/')[-1]}={logger.kv[k]:.6g}" for k in keys), flush=True)
    g["scalar_keys"] = np.array(scalar_keys)
    names, sums, abssums = [], [], []
    for mod in fx.MODULES:
        for k, v in getattr(algo, mod).state_dict().items():
            names.append(f"{mod}.{k}")
            sums.append(float(v.double().sum()))
            abssums.append(float(v.double().abs().sum()))
    g["param_names"], g["param_sums"], g["param_abssums"] = np.array(names), np.array(sums), np.array(abssums)
    np.savez_compressed(out_path, **g)
    print(f"wrote {out_path} ({os.path.getsize(out_path)} bytes)")


def zero_grad_like_torch_1_12():
    """Context manager: torch.optim.Optimizer.zero_grad() defaults to set_to_none=False, as in the torch==1.12.1 the
    reference pins (requirements.txt:17) -- gradients are ZEROED, not dropped."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        orig = torch.optim.Optimizer.zero_grad

        def zero_grad(self, set_to_none=False):
            return orig(self, set_to_none=set_to_none)

        torch.optim.Optimizer.zero_grad = zero_grad
        try:
            yield
        finally:
            torch.optim.Optimizer.zero_grad = orig

    return cm()


def run_tia_case(TIA, L, B, H, A, n_updates, feeder, record, out_path, **cfg_over):
    """The reference's TIA (tia.py) on seeded parameters / batches / noise: logged scalars, the pre-clip total
    norms of its clip_grad_norm_ calls (model, distractor reward x tia_reward_train_steps, actor, value), the task
    latents and per-tensor checksums of every module after the last update."""
    cfg = fx.default_config(algo="tia", batch_size=B, chunk_size=L, horizon=H, **cfg_over)
    logger = RecLogger()
    algo = TIA(cfg, FakeEnv(A), FakeEnv(A), logger)
    mods = fx.MODULES + fx.TIA_EXTRA_MODULES
    params = fx.make_params(A, seed=7, tia=True)
    for mod in mods:
        m = getattr(algo, mod)
        assert list(m.state_dict().keys()) == list(params[mod].keys()), (mod, list(m.state_dict().keys()))
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params[mod].items()})
    assert [id(q) for mod in fx.TIA_MODEL_MODULES for q in getattr(algo, mod).parameters()] == [id(q) for q in algo.model_params]
    T = L - 1
    g = OrderedDict()
    g["meta"] = np.array([L, B, H, A, n_updates, cfg.tia_reward_train_steps], dtype=np.int64)
    g["coefs"] = np.array([cfg.tia_obs_coef, cfg.tia_adv_coef], dtype=np.float64)
    scalar_keys = None
    for u in range(n_updates):
        obs_u8, actions, rewards, dones = fx.make_batch(L, B, A, seed=11 + u)
        noise = fx.make_noise(L, B, H, A, seed=101 + u, tia=True)
        feeder.load(noise, T, H)
        record["clip_calls"].clear()
        record["total_norms"].clear()
        logger.kv.clear()
        obs = torch.from_numpy(fx.preprocess_u8(obs_u8))
        beliefs, post = algo.train_dynamics(obs, torch.from_numpy(actions), torch.from_numpy(rewards),
                                            torch.from_numpy(1 - dones))
        algo.train_actor_critic(beliefs.flatten(0, 1), post.flatten(0, 1))
        assert not feeder.queue, "noise left over: draw order differs"
        keys = sorted(logger.kv.keys())
        scalar_keys = scalar_keys or keys
        assert keys == scalar_keys
        g[f"u{u}/scalars"] = np.array([logger.kv[k] for k in keys], dtype=np.float64)
        g[f"u{u}/total_norms"] = np.array(record["total_norms"], dtype=np.float64)
        g[f"u{u}/beliefs"] = beliefs.numpy().copy()
        g[f"u{u}/posterior_states"] = post.numpy().copy()
        print(f"  [{os.path.basename(out_path)}] update {u}: "
              + " ".join(f"{k.split('/')[-1]}={logger.kv[k]:.6g}" for k in keys), flush=True)
    g["scalar_keys"] = np.array(scalar_keys)
    names, sums, abssums = [], [], []
    for mod in mods:
        for k, v in getattr(algo, mod).state_dict().items():
            names.append(f"{mod}.{k}")
            sums.append(float(v.double().sum()))
            abssums.append(float(v.double().abs().sum()))
    g["param_names"] = np.array(names)
    g["param_sums"] = np.array(sums, dtype=np.float64)
    g["param_abssums"] = np.array(abssums, dtype=np.float64)
    np.savez_compressed(out_path, **g)
    print(f"wrote {out_path} ({os.path.getsize(out_path)} bytes)")


def run_finetune_case(L, B, A, n_updates, feeder, record, out_path):
    """The reference's FinetunedRePo.train_encoder (repo_adapt.py:26-94) on seeded parameters / batches / noise."""
    from algorithms.repo.repo_adapt import FinetunedRePo

    H = 5
    cfg = fx.default_config(algo="repo", batch_size=B, chunk_size=L, horizon=H, init_beta=0.05, target_kl=0.1)
    logger = RecLogger()
    algo = FinetunedRePo(cfg, FakeEnv(A), FakeEnv(A), logger)
    load_params(algo, fx.make_params(A, seed=7))
    T = L - 1
    g = OrderedDict()
    g["meta"] = np.array([L, B, H, A, n_updates], dtype=np.int64)
    g["cfg"] = np.array([cfg.init_beta, cfg.target_kl], dtype=np.float64)
    keys = None
    for u in range(n_updates):
        obs_u8, actions, rewards, dones = fx.make_batch(L, B, A, seed=11 + u)
        noise = fx.make_noise(L, B, H, A, seed=101 + u)
        feeder.queue = [x for t in range(T) for x in (noise["obs_prior"][t], noise["obs_post"][t])]
        record["clip_calls"].clear()
        record["total_norms"].clear()
        logger.kv.clear()
        algo.train_encoder(torch.from_numpy(fx.preprocess_u8(obs_u8)), torch.from_numpy(actions), torch.from_numpy(rewards),
                           torch.from_numpy(1 - dones))
        assert not feeder.queue
        keys = keys or sorted(logger.kv.keys())
        g[f"u{u}/scalars"] = np.array([logger.kv[k] for k in keys], dtype=np.float64)
        g[f"u{u}/total_norms"] = np.array(record["total_norms"], dtype=np.float64)
        g[f"u{u}/log_beta"] = np.array(algo.log_beta.item(), dtype=np.float64)
        print(f"  [{os.path.basename(out_path)}] update {u}: " + " ".join(f"{k.split('/')[-1]}={logger.kv[k]:.6g}" for k in keys),
              flush=True)
    g["scalar_keys"] = np.array(keys)
    names, sums, abssums = [], [], []
    for mod in fx.MODULES:
        for k, v in getattr(algo, mod).state_dict().items():
            names.append(f"{mod}.{k}")
            sums.append(float(v.double().sum()))
            abssums.append(float(v.double().abs().sum()))
    g["param_names"], g["param_sums"], g["param_abssums"] = np.array(names), np.array(sums), np.array(abssums)
    np.savez_compressed(out_path, **g)
    print(f"wrote {out_path} ({os.path.getsize(out_path)} bytes)")


def run_tia_zeros(TIA, feeder, record):
    # TIA under the zero_grad() of the torch==1.12.1 the reference pins (gradients zeroed, not set to None): every
    # world-model parameter takes zero-gradient Adam steps in the fitting loop, the distractor reward head one in the
    # main step from the second update on.  Two fitting steps, three updates.
    with zero_grad_like_torch_1_12():
        run_tia_case(TIA, 8, 4, 5, 6, 3, feeder, record, os.path.join(OUT, "tia_zeros.npz"), tia_reward_train_steps=2)


def run_mt_cases(feeder, record):
    # multitask (f4): MultitaskDreamer at the defaults; MultitaskRePo with a beta large enough for the per-task
    # multipliers to matter and a target below the KL, 3 tasks (every multitask environment of the reference has 3)
    run_mt_case("dreamer_multitask", 8, 4, 5, 6, 3, 3, feeder, record, os.path.join(OUT, "mt_dreamer_tiny.npz"))
    run_mt_case("repo_multitask", 8, 4, 5, 6, 3, 3, feeder, record, os.path.join(OUT, "mt_repo_tiny.npz"),
                init_beta=0.05, target_kl=0.3, beta_lr=1e-2)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Dreamer, RePo, TIA = import_reference()
    feeder = NoiseFeeder()
    record = {"clip_calls": [], "total_norms": []}
    install_patches(feeder, record)

    if "--finetune-only" in sys.argv:
        run_finetune_case(8, 4, 6, 3, feeder, record, os.path.join(OUT, "finetune_tiny.npz"))
        return
    if "--mt-only" in sys.argv:
        run_mt_cases(feeder, record)
        return
    if "--tia-only" in sys.argv:
        run_tia_case(TIA, 8, 4, 5, 6, 3, feeder, record, os.path.join(OUT, "tia_tiny.npz"))
        run_tia_case(TIA, 6, 3, 4, 7, 2, feeder, record, os.path.join(OUT, "tia_coefs.npz"), tia_obs_coef=0.5,
                     tia_adv_coef=2.0, tia_reward_train_steps=2)
        run_tia_zeros(TIA, feeder, record)
        return
    # tiny unit-test size, full latents
    run_case(RePo, "repo", 8, 4, 5, 6, 3, True, feeder, record, os.path.join(OUT, "repo_tiny.npz"))
    # config 5: Dreamer objective (free-nats KL, attached decoder)
    run_case(Dreamer, "dreamer", 8, 4, 5, 6, 3, True, feeder, record, os.path.join(OUT, "dreamer_tiny.npz"))
    # ragged / odd sizes: B not a multiple of anything, A=7 (config 4's action size)
    run_case(RePo, "repo", 6, 3, 3, 7, 2, True, feeder, record, os.path.join(OUT, "repo_odd.npz"))
    # config 1 shapes, scalar + sliced latents only
    if "--skip-c1" not in sys.argv:  # the one fixture that takes a minute of CPU
        run_case(RePo, "repo", 50, 16, 15, 6, 2, False, feeder, record, os.path.join(OUT, "repo_c1.npz"))
    # FinetunedRePo (f4): encoder-only adaptation, beta large enough for the KL term to matter
    run_finetune_case(8, 4, 6, 3, feeder, record, os.path.join(OUT, "finetune_tiny.npz"))
    # TIA (f4): default coefficients, and non-default ones with two distractor-reward fitting steps
    run_tia_case(TIA, 8, 4, 5, 6, 3, feeder, record, os.path.join(OUT, "tia_tiny.npz"))
    run_tia_case(TIA, 6, 3, 4, 7, 2, feeder, record, os.path.join(OUT, "tia_coefs.npz"), tia_obs_coef=0.5,
                 tia_adv_coef=2.0, tia_reward_train_steps=2)
    run_tia_zeros(TIA, feeder, record)
    run_mt_cases(feeder, record)


if __name__ == "__main__":
    main()
