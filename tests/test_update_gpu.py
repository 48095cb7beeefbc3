"""Whole-update parity on the GPU: the HIP agent (through the C ABI) against
 (a) the committed golden vectors produced by the REFERENCE (tests/golden/*.npz), and
 (b) the CPU oracle run on the same seeded batches and noise.

Tolerance (north_star): every per-step loss within 1e-3 relative of the reference on identical
replay batches; log_beta within 1e-5 absolute.  Latents after the first update are compared at
1e-4; after later updates at 2e-3 absolute (Adam's sign-like early steps amplify fp32 rounding;
see tests/test_oracle_golden.py for the same effect between two CPU runs).
"""
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle.repo_oracle import OracleAgent
from tests.util import log

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _poison_lds():
    """Start every test from NaN-filled LDS on all CUs: reads of never-written LDS cannot hide."""
    from repo_amd._lib import lib

    assert lib().repo_debug_poison_lds(torch.cuda.current_stream().cuda_stream) == 0
    yield


class Space:
    def __init__(self, shape):
        self.shape = shape


class Env:
    def __init__(self, A, image=64):
        self.observation_space = Space((3, image, image))
        self.action_space = Space((A,))


class Logger:
    dir = "/tmp"

    def __init__(self):
        self.kv = {}
        self.nonfinite = []

    def record(self, k, v, exclude=None):
        self.kv[k] = v
        if isinstance(v, (float, np.floating)) and not math.isfinite(v):
            self.nonfinite.append((k, float(v)))    # every non-finite scalar ever logged, not only the last value per key

    def dump(self, step=None):
        pass


def make_agent(algo, L, B, H, A, seed=7, image=64):
    from repo_amd.algorithms.repo import Dreamer, RePo
    from repo_amd.common.utils import set_gpu_mode

    set_gpu_mode(True)
    cfg = fx.default_config(algo=algo, batch_size=B, chunk_size=L, horizon=H)
    agent = (RePo if algo == "repo" else Dreamer)(cfg, Env(A, image), Env(A, image), Logger())
    params = fx.make_params(A, seed, image)
    for mod in fx.MODULES:
        sd = {k: torch.from_numpy(v) for k, v in params[mod].items()}
        agent._load_module(getattr(agent, mod), sd)
    return agent, cfg


def dev_batch(L, B, A, seed, u8=True, image=64):
    obs, act, rew, done = fx.make_batch(L, B, A, seed=seed, image=image)
    o = torch.from_numpy(obs if u8 else fx.preprocess_u8(obs)).cuda()
    return (o, torch.from_numpy(act).cuda(), torch.from_numpy(rew).cuda(), torch.from_numpy(done).cuda()), (obs, act, rew, done)


def dev_noise(L, B, H, A, seed):
    n = fx.make_noise(L, B, H, A, seed=seed)
    return {k: torch.from_numpy(v).cuda() for k, v in n.items()}, n


CASES = [("repo_tiny.npz", "repo"), ("dreamer_tiny.npz", "dreamer"), ("repo_odd.npz", "repo"), ("repo_c1.npz", "repo")]


@pytest.mark.parametrize("fname,algo", CASES)
def test_update_matches_reference_goldens(golden_dir, fname, algo):
    g = np.load(os.path.join(golden_dir, fname))
    L, B, H, A, n_updates = (int(x) for x in g["meta"])
    agent, cfg = make_agent(algo, L, B, H, A)
    keys = [str(k) for k in g["scalar_keys"]]
    full = fname != "repo_c1.npz"
    worst = 0.0
    for u in range(n_updates):
        batch, _ = dev_batch(L, B, A, 11 + u, u8=(u % 2 == 0))  # alternate uint8 / float32 frames
        agent.noise_source, _ = dev_noise(L, B, H, A, 101 + u)
        beliefs, post = agent.train_dynamics(batch[0], batch[1], batch[2], 1.0 - batch[3])
        agent.train_actor_critic(beliefs.flatten(0, 1), post.flatten(0, 1))
        scal = agent.last_scalars
        bl, ps = beliefs.cpu().numpy(), post.cpu().numpy()
        if not full:
            bl, ps = bl[::7, ::3, :8], ps[::7, ::3, :8]
        atol = 1e-4 if u == 0 else 2e-3
        np.testing.assert_allclose(bl, g[f"u{u}/beliefs"], rtol=1e-3, atol=atol)
        np.testing.assert_allclose(ps, g[f"u{u}/posterior_states"], rtol=1e-3, atol=atol)
        want = g[f"u{u}/scalars"]
        for k, w in zip(keys, want):
            got = scal[k]
            r = abs(got - w) / (abs(w) + 1e-12)
            worst = max(worst, r)
            log(f"[{fname}] update {u} {k}: got {got:.7g} ref {w:.7g} rel {r:.2e}")
            assert r < 1e-3, (fname, u, k, got, w)
        if algo == "repo":
            assert abs(float(agent.log_beta) - float(g[f"u{u}/log_beta"])) < 1e-5
        tn = g[f"u{u}/total_norms"]
        gn = agent.last_grad_norms
        for name, w in zip(("model", "actor", "value"), tn):
            r = abs(gn[name] - w) / w
            log(f"[{fname}] update {u} grad-norm {name}: got {gn[name]:.6g} ref {w:.6g} rel {r:.2e}")
            assert r < 2e-3
    log(f"[{fname}] worst scalar rel err {worst:.2e}")
    # parameter checksums after the last Adam step
    names = [str(n) for n in g["param_names"]]
    have = {}
    for m in fx.MODULES:
        for k, v in getattr(agent, m).state_dict().items():
            have[f"{m}.{k}"] = (float(v.double().sum()), float(v.double().abs().sum()))
    for n, s_, a_ in zip(names, g["param_sums"], g["param_abssums"]):
        assert abs(have[n][1] - a_) <= 1e-3 * abs(a_) + 1e-6, (n, have[n][1], a_)
        assert abs(have[n][0] - s_) <= 1e-3 * abs(a_) + 1e-6, (n, have[n][0], s_)


# image=128: the build-defined 128 x 128 conv stack (BASELINE config 4's frame size).  The reference has no such
# model (its flatten hard-codes 64 x 64), so this variant's parity is pinned by the oracle only -- "parity unpinned
# by the reference" (DESIGN.md section 6).
@pytest.mark.parametrize("algo,image,A", [("repo", 64, 6), ("dreamer", 64, 6), ("repo", 128, 7), ("dreamer", 128, 7)])
def test_update_matches_oracle_latents_and_grads(algo, image, A):
    L, B, H = (10, 5, 6) if image == 64 else (7, 3, 5)
    agent, cfg = make_agent(algo, L, B, H, A, image=image)
    oracle = OracleAgent(cfg, A, seed=7, image=image)
    for u in range(2):
        batch, host = dev_batch(L, B, A, 40 + u, u8=(u == 0), image=image)
        agent.noise_source, nz = dev_noise(L, B, H, A, 140 + u)
        # gradients of THIS update before the optimiser touches them: snapshot via a hook on step
        beliefs, post = agent.train_dynamics(batch[0], batch[1], batch[2], 1.0 - batch[3])
        g_model = agent.model_optimizer.grad.clone()
        agent.train_actor_critic(beliefs.flatten(0, 1), post.flatten(0, 1))
        g_actor = agent.actor_optimizer.grad.clone()
        g_value = agent.value_optimizer.grad.clone()
        ob, op_, oscal = oracle.update(*host, nz)
        tol = 1e-4 if u == 0 else 2e-3
        np.testing.assert_allclose(beliefs.cpu().numpy(), ob.numpy(), rtol=1e-3, atol=tol)
        np.testing.assert_allclose(post.cpu().numpy(), op_.numpy(), rtol=1e-3, atol=tol)
        for k, w in oscal.items():
            got = agent.last_scalars[k]
            assert abs(got - w) <= 1e-3 * abs(w) + 1e-7, (u, k, got, w)

        def flat(grads, opt):
            out = torch.zeros(opt.numel)
            for gr, o, p in zip(grads, opt.offsets, opt.params):
                if gr is not None:
                    out[o : o + p.numel()] = gr.reshape(-1)
            return out

        for name, got, want in (
            ("model", g_model, flat(oracle.last["model_grads"], agent.model_optimizer)),
            ("actor", g_actor, flat(oracle.last["actor_grads"], agent.actor_optimizer)),
            ("value", g_value, flat(oracle.last["value_grads"], agent.value_optimizer)),
        ):
            e = ((got.cpu() - want).norm() / want.norm()).item()
            log(f"[oracle {algo} {image}x{image}] update {u} flat grad {name}: l2 rel {e:.2e}")
            # normwise (a ReLU unit within rounding of zero may flip between two fp32 runs); observed 2e-7 .. 3e-6
            assert e < 1e-3, (name, e)


def test_full_size_update_properties():
    """BASELINE config 2 shapes (B=50, L=50, H=15): size-independent properties."""
    L, B, H, A = 50, 50, 15, 6
    agent, cfg = make_agent("repo", L, B, H, A)
    batch, _ = dev_batch(L, B, A, 1234)
    p0 = agent.model_optimizer.flat.clone()
    agent.update(batch)
    s1 = dict(agent.last_scalars)
    assert all(np.isfinite(v) for v in s1.values()), s1
    # sanity anchors observed on the reference with uniform-random frames
    assert 13000 < s1["train/obs_loss"] < 13800
    assert 0.05 < s1["train/kl_div"] < 1.0
    assert abs(s1["train/beta"] - 1e-5) < 1e-7
    # every parameter moved by at most ~lr (Adam's first step is sign-like) and most of them did move
    dlt = (agent.model_optimizer.flat - p0).abs()
    assert float(dlt.max()) <= 3e-4 * 1.01
    assert float((dlt > 0).float().mean()) > 0.9
    # determinism: same batch + same injected noise on a fresh agent gives bit-identical scalars
    noise, _ = dev_noise(L, B, H, A, 5)
    outs = []
    for _ in range(2):
        ag, _ = make_agent("repo", L, B, H, A)
        ag.noise_source = noise
        ag.update(batch)
        outs.append(dict(ag.last_scalars))
    assert outs[0] == outs[1], (outs[0], outs[1])


@pytest.mark.parametrize("image", [64, 128])
def test_module_autograd_wrappers(image):
    """encoder / obs_model / reward_model / transition_model.observe as autograd nodes."""
    from repo_amd.algorithms.repo.models.utils import bottle
    from oracle import repo_oracle as ro

    L, B, H, A = 6, 3, 4, 6
    agent, cfg = make_agent("dreamer", L, B, H, A, image=image)
    batch, host = dev_batch(L, B, A, 77, u8=False, image=image)
    obs, act, rew, done = batch
    nz, nzh = dev_noise(L, B, H, A, 177)
    T = L - 1
    embeds = bottle(agent.encoder, (obs,))
    b0 = torch.zeros(B, cfg.belief_size, device="cuda")
    s0 = torch.zeros(B, cfg.state_size, device="cuda")
    outs = agent.transition_model.observe(b0, s0, act[:-1], embeds[1:], 1 - done[:-1], noise=(nz["obs_prior"], nz["obs_post"]))
    beliefs, post = outs[0], outs[4]
    recon = bottle(agent.obs_model, (beliefs, post))
    rp = bottle(agent.reward_model, (beliefs, post))
    loss = (0.5 * (recon - obs[1:]) ** 2).sum((2, 3, 4)).mean() + (rp**2).mean() + outs[2].pow(2).mean()
    for p in agent.model_params:
        p.grad = None
    loss.backward()
    # oracle
    o = OracleAgent(cfg, A, seed=7, image=image)
    ho = torch.from_numpy(fx.preprocess_u8(host[0]))
    oe = ro.encoder_fwd(o.p["encoder"], ho.reshape(L * B, 3, image, image)).reshape(L, B, -1)
    oo = ro.observe(o.p["transition_model"], torch.zeros(B, 200), torch.zeros(B, 30), torch.from_numpy(host[1])[:-1], oe[1:],
                    1 - torch.from_numpy(host[3])[:-1], torch.from_numpy(nzh["obs_prior"]), torch.from_numpy(nzh["obs_post"]))
    fb, fs = oo[0].reshape(T * B, -1), oo[4].reshape(T * B, -1)
    orec = ro.decoder_fwd(o.p["obs_model"], fb, fs).reshape(T, B, 3, image, image)
    orp = ro.scalar_head(o.p["reward_model"], fb, fs).reshape(T, B)
    ol = (0.5 * (orec - ho[1:]) ** 2).sum((2, 3, 4)).mean() + (orp**2).mean() + oo[2].pow(2).mean()
    ol.backward()
    assert abs(loss.item() - ol.item()) < 1e-4 * abs(ol.item())
    got = torch.cat([p.grad.reshape(-1) for p in agent.model_params]).cpu()
    want = torch.cat([q.grad.reshape(-1) for q in o.model_params])
    e = ((got - want).norm() / want.norm()).item()
    log(f"[autograd wrappers] loss {loss.item():.6g} vs {ol.item():.6g}; model grad l2 rel {e:.2e}")
    assert e < 5e-3


def test_pipelined_updates_are_bit_deterministic():
    """Two identically seeded agents, 25 pipelined updates each (in-kernel Philox noise, two 16-row groups in the
    column-split scans): bit-identical parameters.  A stale or torn read in the scans' all-gathers through L2
    (csrc/scan_cs.hip), or any other race between the update's streams, would show here (tools/determinism_soak.py is
    the long version: 400 updates at B=50, 600 at B=7)."""
    L, B, H, A = 12, 20, 6, 6
    batches = [dev_batch(L, B, A, 300 + i)[0] for i in range(3)]
    finals = []
    for _ in range(2):
        agent, _cfg = make_agent("repo", L, B, H, A)
        agent.seed_noise(99)
        for i in range(25):
            agent.update(batches[i % 3], join=False)
        agent.synchronize()
        torch.cuda.synchronize()
        assert all(np.isfinite(v) for v in agent.last_scalars.values())
        finals.append([o.flat.clone() for o in (agent.model_optimizer, agent.actor_optimizer, agent.value_optimizer)]
                      + [agent.log_beta.clone().reshape(1)])
    for a, b in zip(*finals):
        assert torch.equal(a, b)
