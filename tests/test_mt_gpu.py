"""Multitask agents (MultitaskDreamer / MultitaskRePo, /root/reference/algorithms/repo/
dreamer_mt.py, repo_mt.py) on the GPU: the FiLM and per-task-dual kernels against plain PyTorch restatements, the
conditioned rollout against the oracle, and whole updates against (a) the golden vectors the REFERENCE's classes
produced (tests/golden/mt_*.npz) and (b) the CPU oracle's gradients on the same seeded batches and noise.
Tolerances as in tests/test_update_gpu.py (north_star: per-step losses within 1e-3 relative)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures as fx
from oracle import repo_oracle as ro
from tests.test_update_gpu import Env, Logger, dev_batch, dev_noise
from tests.util import l2err, log, relerr

pytestmark = pytest.mark.gpu


class MtEnv(Env):
    def __init__(self, A, C):
        super().__init__(A)
        self.num_tasks = C


def make_mt(algo, L, B, H, A, C, seed=7, **over):
    from repo_amd.algorithms.repo import MultitaskDreamer, MultitaskRePo
    from repo_amd.common.utils import set_gpu_mode

    set_gpu_mode(True)
    cfg = fx.default_config(algo=algo, batch_size=B, chunk_size=L, horizon=H, share_repr=False, **over)
    env = MtEnv(A, C)
    agent = (MultitaskRePo if algo == "repo_multitask" else MultitaskDreamer)(cfg, env, env, Logger())
    params = fx.make_params(A, seed, cond=C)
    for mod in fx.MODULES:
        agent._load_module(getattr(agent, mod), {k: torch.from_numpy(v) for k, v in params[mod].items()})
    return agent, cfg


def dev_tasks(L, B, C, seed):
    t = fx.make_tasks(L, B, C, seed=seed)
    return torch.from_numpy(t).cuda(), t


@pytest.mark.parametrize("n,C,hw", [(5, 32, 31), (3, 64, 14), (7, 128, 6), (4, 256, 2), (6, 128, 5), (2, 64, 13), (3, 32, 30)])
def test_film_fwd_bwd_match_torch(n, C, hw):
    """repo_film_fwd / repo_film_bwd on every modulated plane size of the two conv stacks (encoder 31/14/6/2, decoder
    5/13/30) against autograd, gamma / beta taken from the middle of a wider FiLM row as the stacks do."""
    from repo_amd import ops

    rs = np.random.RandomState(n * 1000 + C + hw)
    ld, goff, boff = 2 * C + 24, 5, C + 19
    y = torch.from_numpy(rs.standard_normal((n, C, hw, hw)).astype(np.float32)).cuda()
    film = torch.from_numpy(rs.standard_normal((n, ld)).astype(np.float32) * 0.7).cuda()
    up = torch.from_numpy(rs.standard_normal((n, C, hw, hw)).astype(np.float32)).cuda()
    yd, fd = y.double().requires_grad_(True), film.double().requires_grad_(True)
    g, b = fd[:, goff : goff + C], fd[:, boff : boff + C]
    want = torch.relu((1 + g[..., None, None]) * yd + b[..., None, None])
    want.backward(up.double())
    h = ops.film_fwd(y, film, goff, boff)
    assert relerr(h, want) < 1e-6
    dfilm = torch.full_like(film, 7.0)   # only the two (n, C) blocks are written
    dh = ops.relu_mask(up, h)
    dy = ops.film_bwd(dh, y, film, goff, boff, dfilm)
    assert relerr(dy, yd.grad) < 1e-6
    assert relerr(dfilm[:, goff : goff + C], fd.grad[:, goff : goff + C]) < 2e-6
    assert relerr(dfilm[:, boff : boff + C], fd.grad[:, boff : boff + C]) < 2e-6
    mask = torch.ones_like(film, dtype=torch.bool)
    mask[:, goff : goff + C] = False
    mask[:, boff : boff + C] = False
    assert bool((dfilm[mask] == 7.0).all())


def test_film_epilogue_equals_conv_then_film():
    """REPO_EPI_FILM_RELU (the FiLM + ReLU epilogue of the conv / dense kernels, repo_film_tables) against the two-kernel form
    it replaces -- conv with a bias-only epilogue, then repo_film_fwd -- on every modulated layer of both stacks (uint8
    and float frames for the first), and repo_film_bwd_h (y recovered from the output) against repo_film_bwd on the saved y."""
    from repo_amd import ops
    from repo_amd.algorithms.repo.models.conditional import DEC_CHANNELS, ENC_CHANNELS, film_offsets

    rs = np.random.RandomState(11)
    n = 9
    f = lambda *s_, sc=1.0: torch.from_numpy((rs.standard_normal(s_) * sc).astype(np.float32)).cuda()  # noqa: E731
    # ---- encoder
    film = f(n, 2 * sum(ENC_CHANNELS), sc=0.5)
    tabs = ops.film_tables(film, ENC_CHANNELS)
    offs = film_offsets(ENC_CHANNELS)
    for l, (g, b) in enumerate(offs):
        C = ENC_CHANNELS[l]
        assert torch.equal(tabs[l][:, 0], 1 + film[:, g : g + C]) and torch.equal(tabs[l][:, 1], film[:, b : b + C])
    layers = (ops.ENC1, ops.ENC2, ops.ENC3, ops.ENC4)
    x8 = torch.from_numpy(rs.randint(0, 256, (n, 3, 64, 64)).astype(np.uint8)).cuda()
    for first in (x8, ((x8.float() / 255) * 2) - 1):
        x = first
        for l, layer in enumerate(layers):
            (cb, hb, _), (cs, hs_, _) = ops.conv_shapes(layer)
            w, bias = f(cs, cb, 4, 4, sc=0.1), f(cs)
            y = ops.conv_down(layer, x, w, bias, epi=ops.EPI_NONE)
            want = ops.film_fwd(y, film, *offs[l])
            got = ops.conv_down(layer, x, w, bias, epi=ops.EPI_FILM_RELU, aux=tabs[l])
            assert relerr(got, want) < 2e-6, (l, float(relerr(got, want)))
            assert float((got == 0).float().mean()) > 0.2          # the ReLU is live
            dh = ops.relu_mask(f(*want.shape), want)
            dfa, dfb = torch.zeros_like(film), torch.zeros_like(film)
            dya = ops.film_bwd(dh, y, film, *offs[l], dfa)
            dyb = ops.film_bwd_h(dh, want, film, *offs[l], dfb)
            assert torch.equal(dya, dyb)
            g, b = offs[l]
            assert relerr(dfb[:, g : g + cs], dfa[:, g : g + cs]) < 2e-5 and torch.equal(dfb[:, b : b + cs], dfa[:, b : b + cs])
            x = want
    # ---- decoder: the 1024 -> 128 x 5 x 5 layer as a dense product, then the two scatter-kernel layers
    film = f(n * 40, 2 * sum(DEC_CHANNELS), sc=0.5)
    rows = film.shape[0]
    tabs = ops.film_tables(film, DEC_CHANNELS)
    offs = film_offsets(DEC_CHANNELS)
    h0, w1, b1 = f(rows, 1024, sc=0.3), f(1024, 3200, sc=0.05), f(128)
    y1 = ops.gemm(h0, w1, bias=b1, bias_div=25).view(rows, 128, 5, 5)
    want = ops.film_fwd(y1, film, *offs[0])
    got = ops.gemm(h0, w1, bias=b1, bias_div=25, epi=ops.EPI_FILM_RELU, aux=tabs[0].view(rows, -1)).view(rows, 128, 5, 5)
    assert relerr(got, want) < 2e-6
    x = want
    for l, layer in ((1, ops.DEC2), (2, ops.DEC3)):
        (cb, hb, _), (cs, hs_, _) = ops.conv_shapes(layer)
        ks = ops.CONV_GEO[layer][3]
        w, bias = f(cs, cb, ks, ks, sc=0.1), f(cb)
        y = ops.conv_up(layer, x, w, bias, epi=ops.EPI_NONE)
        want = ops.film_fwd(y, film, *offs[l])
        got = ops.conv_up(layer, x, w, bias, epi=ops.EPI_FILM_RELU, aux=tabs[l])
        assert relerr(got, want) < 2e-6, (layer, float(relerr(got, want)))
        x = want


def test_film_backward_from_h_is_exact_on_gated_off_channels():
    """A FiLM layer can gate a channel off for a task: 1 + gamma -> 0.  The fused form keeps only h = relu((1 + gamma) y
    + beta); recovering y = (h - beta) / (1 + gamma) there loses it (error ~ eps |beta| / |(1 + gamma) y|; nothing at
    all at exactly 0, where the reference still has d gamma = sum dh * y != 0 whenever beta > 0).  repo_film_bwd_h
    recomputes y from the layer itself on planes with |1 + gamma| < 1e-3: every modulated layer of both stacks, with planes
    at gamma = -1 and -1 +- 1e-4 (recomputed: within 1e-5 of fp64 torch autograd of the layer), -1 +- 0.05 (recovered:
    the same bar) and -1 +- 2e-3 (recovered just outside the threshold: the documented eps |beta| / |(1 + gamma) y|)."""
    from repo_amd import ops
    from repo_amd.algorithms.repo.models.conditional import DEC_CHANNELS, ENC_CHANNELS, film_offsets

    rs = np.random.RandomState(23)
    f = lambda *s_, sc=1.0: torch.from_numpy((rs.standard_normal(s_) * sc).astype(np.float32)).cuda()  # noqa: E731
    adversarial = np.array([-1.0, -1.0 + 1e-4, -1.0 - 1e-4, -0.95, -1.05, -0.998, -1.002, -0.93], dtype=np.float32)

    def spiked_film(n, channels):
        film = (rs.standard_normal((n, 2 * sum(channels))) * 0.5).astype(np.float32)
        tot = sum(channels)
        film[:, tot:] = np.abs(film[:, tot:]) + 0.2          # beta > 0: a gated-off plane is ACTIVE (h = beta > 0)
        o = 0
        for C in channels:        # every layer of the stack gets the adversarial planes, in every image
            for i in range(n):
                film[i, o + rs.choice(C, size=len(adversarial), replace=False)] = adversarial
            o += C
        return torch.from_numpy(film).cuda()

    def check(name, y_fn, x, w, bias, film, off, exact):
        """y_fn(x64, w64, b64) -> the layer's conv output in fp64 (autograd); FiLM + ReLU + a random cotangent on top."""
        g_off, b_off = off
        xc, fc = x.cpu(), film.cpu()
        xs = xc.double() if xc.dtype != torch.uint8 else ((xc.double() / 255) * 2 - 1)
        gam = fc[:, g_off:].double().clone().requires_grad_(True)
        bet = fc[:, b_off:].double().clone().requires_grad_(True)
        y = y_fn(xs, w.cpu().double(), bias.cpu().double())           # fp64 on the host
        C = y.shape[1]
        sh = (y.shape[0], C) + (1,) * (y.dim() - 2)
        hh = torch.relu((1 + gam[:, :C].reshape(sh)) * y + bet[:, :C].reshape(sh))
        cot = f(*y.shape).cpu().double()
        (hh * cot).sum().backward()
        h32 = hh.detach().float().contiguous().cuda()
        dh = ops.relu_mask(cot.float().contiguous().cuda(), h32)
        dfilm = torch.zeros_like(film)
        dy = ops.film_bwd_h(dh, h32, film, g_off, b_off, dfilm, exact=exact)
        want_g, want_b = gam.grad[:, :C].cuda(), bet.grad[:, :C].cuda()
        got_g, got_b = dfilm[:, g_off : g_off + C].double(), dfilm[:, b_off : b_off + C].double()
        near = ((1 + film[:, g_off : g_off + C]).abs() > 1e-3) & ((1 + film[:, g_off : g_off + C]).abs() < 1e-2)
        eg = float(((got_g - want_g).abs() * ~near).max() / want_g.abs().max())
        enear = float(((got_g - want_g).abs() * near).max() / want_g.abs().max())   # recovered at |1 + gamma| = 2e-3
        eb = float((got_b - want_b).abs().max() / want_b.abs().max())
        scale = (1 + film[:, g_off : g_off + C]).reshape(sh)
        edy = float((dy - dh * scale).abs().max())
        gated = (1 + film[:, g_off : g_off + C]).abs() < 1e-3
        assert bool(gated.any()) and float(want_g[gated].abs().max()) > 1e-2 * float(want_g.abs().max())  # they matter
        log(f"[film gated-off] {name}: d gamma {eg:.2e} (planes at |1 + gamma| = 2e-3: {enear:.2e}) d beta {eb:.2e}")
        assert eg < 1e-5 and eb < 1e-5 and edy == 0.0 and enear < 5e-4, (name, eg, eb, edy, enear)
        # without the layer's description the exactly-gated planes lose their gamma gradient (documented, kind 0)
        dfilm0 = torch.zeros_like(film)
        ops.film_bwd_h(dh, h32, film, g_off, b_off, dfilm0)
        zero = (1 + film[:, g_off : g_off + C]) == 0
        assert bool(zero.any()) and float(dfilm0[:, g_off : g_off + C][zero].abs().max()) == 0.0

    n = 5
    film = spiked_film(n, ENC_CHANNELS)
    offs = film_offsets(ENC_CHANNELS)
    x8 = torch.from_numpy(rs.randint(0, 256, (n, 3, 64, 64)).astype(np.uint8)).cuda()
    conv = lambda xs, w, b: F.conv2d(xs, w, b, stride=2)     # noqa: E731
    for l, layer in enumerate((ops.ENC1, ops.ENC2, ops.ENC3, ops.ENC4)):
        (cb, hb, _), (cs, hs_, _) = ops.conv_shapes(layer)
        w, bias = f(cs, cb, 4, 4, sc=0.1), f(cs)
        inputs = (x8, ((x8.float() / 255) * 2 - 1).contiguous()) if l == 0 else (f(n, cb, hb, hb).abs().contiguous(),)
        for x in inputs:
            check(f"enc{l + 1} {x.dtype}", conv, x, w, bias, film, offs[l], (ops.FILM_CONV_DOWN, layer, x, w, bias))
    rows = 11
    film = spiked_film(rows, DEC_CHANNELS)
    offs = film_offsets(DEC_CHANNELS)
    h0, w1, b1 = f(rows, 1024, sc=0.3), f(1024, 3200, sc=0.05), f(128)
    dense = lambda xs, w, b: (xs @ w).view(rows, 128, 25) + b.view(1, 128, 1)   # noqa: E731
    check("dec1 (dense)", dense, h0, w1, b1, film, offs[0], (ops.FILM_DENSE, 1024, h0, w1, b1))
    # ... and with one bias per output element, as the composed decoder head describes the layer (functional.dec_head_compose)
    check("dec1 (dense, per-element bias)", dense, h0, w1, b1, film, offs[0],
          (ops.FILM_DENSE, (1024, True), h0, w1, b1.repeat_interleave(25).contiguous()))
    for l, layer in ((1, ops.DEC2), (2, ops.DEC3)):
        (cb, hb, _), (cs, hs_, _) = ops.conv_shapes(layer)
        ks = ops.CONV_GEO[layer][3]
        w, bias, x = f(cs, cb, ks, ks, sc=0.1), f(cb), f(rows, cs, hs_, hs_).abs().contiguous()
        tconv = lambda xs, w_, b: F.conv_transpose2d(xs, w_, b, stride=2)   # noqa: E731
        check(f"dec{l + 1}", tconv, x, w, bias, film, offs[l], (ops.FILM_CONV_UP, layer, x, w, bias))


@pytest.mark.parametrize("rows,C", [(1, 1), (37, 3), (2450, 3), (500, 13)])
def test_kl_balance_tasks_and_dual_step_match_torch(rows, C):
    """repo_kl_balance_tasks + repo_dual_step_tasks against repo_mt.py:75-99 written with autograd + torch.optim.Adam."""
    from repo_amd import ops

    S = 30
    rs = np.random.RandomState(rows + C)
    f = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32))  # noqa: E731
    pm, qm = f(rows, S), f(rows, S)
    ps, qs = f(rows, S).abs() + 0.2, f(rows, S).abs() + 0.2
    tasks = torch.from_numpy(np.eye(C, dtype=np.float32)[rs.randint(0, C, rows)])
    lb0 = torch.from_numpy(rs.uniform(-3, -1, C).astype(np.float32))
    alpha, target, scale = 5 / 6, 0.7, 1.0 / rows
    P = [t.double().requires_grad_(True) for t in (pm, ps, qm, qs)]
    lb = lb0.double().requires_grad_(True)
    kl_prior = ro.normal_kl(P[2].detach(), P[3].detach(), P[0], P[1]).sum(1)
    kl_post = ro.normal_kl(P[2], P[3], P[0].detach(), P[1].detach()).sum(1)
    kl_div = alpha * kl_prior + (1 - alpha) * kl_post
    viol = kl_div - target
    lbr = tasks.double() @ lb
    kl_loss = (lbr.exp().detach() * viol).mean()
    kl_loss.backward()
    beta_loss = -(lbr * viol.detach()).mean()
    opt = torch.optim.Adam([lb], lr=1e-2)
    opt.zero_grad()
    beta_loss.backward()
    lb_grad = lb.grad.clone()
    opt.step()
    dlb = lb0.cuda().clone()
    sums, g = ops.kl_balance_tasks(pm.cuda(), ps.cuda(), qm.cuda(), qs.cuda(), alpha, dlb, tasks.cuda(), target, scale)
    for got, want in zip(g, P):
        assert relerr(got, want.grad) < 2e-5
    m, v = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    out = ops.dual_step_tasks(dlb, m, v, sums, rows, 1e-2, (0.9, 0.999), 1e-8, 1).cpu().double()
    assert abs(out[0] - kl_div.mean().item()) < 1e-5 * abs(kl_div.mean().item()) + 1e-7
    assert abs(out[1] - kl_loss.item()) < 2e-5 * abs(kl_loss.item()) + 1e-7
    assert abs(out[2] - beta_loss.item()) < 2e-5 * abs(beta_loss.item()) + 1e-6
    # tasks without a row in the batch take a zero gradient: Adam leaves them where they were
    touched = lb_grad != 0
    np.testing.assert_allclose(dlb.cpu().double().numpy(), lb.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(out[3:].numpy(), lb.detach().exp().numpy(), rtol=1e-5)
    assert bool((dlb.cpu()[~touched] == lb0[~touched]).all())


@pytest.mark.parametrize("A,C", [(6, 3), (2, 3), (2, 10), (9, 13)])
def test_conditioned_rollout_matches_oracle(A, C):
    """repo_rssm_imagine_fwd / _bwd with cond (ConditionalTransitionModel.imagine + ConditionalActorModel,
    models/rssm.py:221-249): every saved tensor against the oracle's conditioned rollout, the actor-output gradient and
    the start-state gradient against autograd through it.  A = 6: the persistent engines (the condition in their K
    padding); A = 2 -- the reference's pointmass and dmc-mixed multitask suites (tabletop/pointmass.py:114) -- and
    C = 13: outside their shapes, the per-step engine with widened rows."""
    from repo_amd import ops

    D, S, Hm, N = 200, 30, 4, 37
    P = fx.make_params(A, 7, cond=C)
    rp = {k: torch.from_numpy(v).requires_grad_(False) for k, v in P["transition_model"].items()}
    ap = {k: torch.from_numpy(v).requires_grad_(True) for k, v in P["actor_model"].items()}
    rs = np.random.RandomState(0)
    f = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32))  # noqa: E731
    b0, s0 = (f(N, D) * 0.3).requires_grad_(True), f(N, S).requires_grad_(True)
    cond = torch.from_numpy(np.eye(C, dtype=np.float32)[rs.randint(0, C, N)])
    ea, ep = f(Hm, N, A), f(Hm, N, S)
    ib, ist, im, isd = ro.cond_imagine(rp, ap, b0, s0, cond, Hm + 1, ea, ep)
    up_f, up_m, up_s = f(Hm, N, D + S), f(Hm, N, S), f(Hm, N, S)
    loss = (torch.cat([ib, ist], 2) * up_f).sum() + (im * up_m).sum() + (isd * up_s).sum()
    loss.backward()
    cu = lambda d: [t.detach().cuda() for t in d.values()]  # noqa: E731
    sv = ops.rssm_imagine_fwd(cu(rp), cu(ap), b0.detach().cuda(), s0.detach().cuda(), ea.cuda(), ep.cuda(),
                              cond=cond.cuda())
    assert sv.xsa.shape == (Hm * N, S + A + C)
    assert relerr(sv.featx[1:, :, :D], ib) < 2e-5 and relerr(sv.featx[1:, :, D:], ist) < 2e-5
    assert relerr(sv.prior_mean, im) < 2e-5 and relerr(sv.prior_std, isd) < 2e-5
    assert torch.equal(sv.xsa.view(Hm, N, -1)[:, :, S + A :], cond.cuda().expand(Hm, N, C))
    d_araw, dfeat0 = ops.rssm_imagine_bwd(cu(rp), sv, up_f.cuda(), dprior_mean=up_m.cuda(), dprior_std=up_s.cuda(),
                                          want_dfeat0=True)
    assert l2err(dfeat0[:, :D], b0.grad) < 2e-5 and l2err(dfeat0[:, D:], s0.grad) < 2e-5
    # actor gradients: finish with the trunk backward over [belief | state | cond] rows
    F_ = D + S
    xw = torch.empty(Hm * N, F_ + C).cuda()
    xw[:, :F_] = sv.featx[:Hm].reshape(Hm * N, F_)
    xw.view(Hm, N, F_ + C)[:, :, F_:] = cond.cuda()
    ga = [torch.zeros_like(t) for t in cu(ap)]
    ops.mlp_bwd(cu(ap), xw, [sv.a_hidden[l] for l in range(sv.a_hidden.shape[0])], d_araw, dparams=ga, dx=None)
    for got, (name, want) in zip(ga, ap.items()):
        assert l2err(got, want.grad) < 5e-5, name


@pytest.mark.parametrize("fname,algo", [("mt_dreamer_tiny.npz", "dreamer_multitask"), ("mt_repo_tiny.npz", "repo_multitask")])
def test_mt_update_matches_reference_goldens(golden_dir, fname, algo):
    g = np.load(os.path.join(golden_dir, fname))
    L, B, H, A, n_updates, C = (int(x) for x in g["meta"])
    init_beta, target_kl, beta_lr = (float(x) for x in g["cfg"])
    agent, cfg = make_mt(algo, L, B, H, A, C, init_beta=init_beta, target_kl=target_kl, beta_lr=beta_lr)
    keys = [str(k) for k in g["scalar_keys"]]
    for u in range(n_updates):
        batch, _ = dev_batch(L, B, A, 11 + u, u8=(u % 2 == 0))
        tasks, _ = dev_tasks(L, B, C, 11 + u)
        agent.noise_source, _ = dev_noise(L, B, H, A, 101 + u)
        beliefs, post = agent.train_dynamics(tasks, batch[0], batch[1], batch[2], 1.0 - batch[3])
        agent.train_actor_critic(tasks[1:].flatten(0, 1), beliefs.flatten(0, 1), post.flatten(0, 1))
        scal = agent.last_scalars
        assert sorted(scal) == keys, (sorted(scal), keys)
        atol = 1e-4 if u == 0 else 2e-3
        np.testing.assert_allclose(beliefs.cpu().numpy(), g[f"u{u}/beliefs"], rtol=1e-3, atol=atol)
        np.testing.assert_allclose(post.cpu().numpy(), g[f"u{u}/posterior_states"], rtol=1e-3, atol=atol)
        for k, w in zip(keys, g[f"u{u}/scalars"]):
            r = abs(scal[k] - w) / (abs(w) + 1e-12)
            log(f"[{fname}] update {u} {k}: got {scal[k]:.7g} ref {w:.7g} rel {r:.2e}")
            assert r < 1e-3, (fname, u, k, scal[k], w)
        if algo == "repo_multitask":
            np.testing.assert_allclose(agent.log_beta.cpu().numpy(), g[f"u{u}/log_beta"], rtol=0, atol=1e-5)
        gn = agent.last_grad_norms
        for name, w in zip(("model", "actor", "value"), g[f"u{u}/total_norms"]):
            assert abs(gn[name] - w) < 2e-3 * w, (name, gn[name], w)
    torch.cuda.synchronize()
    sums = {f"{m}.{k}": float(v.double().sum()) for m in fx.MODULES for k, v in getattr(agent, m).state_dict().items()}
    abss = {f"{m}.{k}": float(v.double().abs().sum()) for m in fx.MODULES for k, v in getattr(agent, m).state_dict().items()}
    for n, s_, a_ in zip((str(n) for n in g["param_names"]), g["param_sums"], g["param_abssums"]):
        assert abs(abss[n] - a_) <= 1e-3 * abs(a_) + 1e-7, (n, abss[n], a_)
        assert abs(sums[n] - s_) <= 1e-3 * abs(a_) + 1e-7, (n, sums[n], s_)


@pytest.mark.parametrize("algo,A", [("dreamer_multitask", 6), ("repo_multitask", 6), ("repo_multitask", 2)])
def test_mt_update_matches_oracle_grads(algo, A):
    """Flat pre-clip gradients of the three optimisers against the oracle's autograd, per module too (the FiLM layers'
    and the condition columns' gradients vanish in the flat norm), with the KL term active.  A = 2: the action size of
    two of the reference's three multitask suites (the rollout then runs on the per-step engine)."""
    L, B, H, C = 9, 5, 5, 3
    over = dict(init_beta=0.05, target_kl=0.3, beta_lr=1e-2, free_nats=0.1)
    agent, cfg = make_mt(algo, L, B, H, A, C, **over)
    oracle = ro.OracleMultitask(cfg, A, C, seed=7)
    for u in range(2):
        batch, host = dev_batch(L, B, A, 60 + u, u8=(u == 0))
        tasks, htasks = dev_tasks(L, B, C, 60 + u)
        agent.noise_source, nz = dev_noise(L, B, H, A, 160 + u)
        snap = {}
        for name in ("model", "actor", "value"):
            opt = getattr(agent, f"{name}_optimizer")
            orig = opt.clip_and_step

            def hooked(norm, opt=opt, orig=orig, name=name):
                snap[name] = opt.grad.clone()
                orig(norm)

            opt.clip_and_step = hooked
        agent.update((tasks, *batch))
        for name in ("model", "actor", "value"):
            del getattr(agent, f"{name}_optimizer").clip_and_step
        got = dict(agent.last_scalars)
        _, _, want = oracle.update(htasks, *host, nz)
        for k, w in want.items():
            assert abs(got[k] - w) <= 1e-3 * abs(w) + 1e-7, (u, k, got[k], w)
        for name, oparams, ograds in (("model", oracle.model_params, oracle.last["model_grads"]),
                                      ("actor", oracle.actor_params, oracle.last["actor_grads"]),
                                      ("value", oracle.value_params, oracle.last["value_grads"])):
            opt = getattr(agent, f"{name}_optimizer")
            flat = torch.zeros(opt.numel)
            for o, p, gr in zip(opt.offsets, opt.params, ograds):
                assert tuple(gr.shape) == tuple(p.shape)
                flat[o : o + p.numel()] = gr.reshape(-1)
            e = ((snap[name].cpu() - flat).norm() / flat.norm()).item()
            log(f"[oracle {algo}] update {u} flat grad {name}: l2 rel {e:.2e}")
            assert e < 1e-3, (name, e)
            for o, p, gr, q in zip(opt.offsets, opt.params, ograds, oparams):
                a = snap[name][o : o + p.numel()].cpu()
                em = ((a - gr.reshape(-1)).norm() / (gr.norm() + 1e-12)).item()
                assert em < 5e-3, (name, tuple(p.shape), em)
        if algo == "repo_multitask":
            np.testing.assert_allclose(agent.log_beta.cpu().numpy(), oracle.log_beta.detach().numpy(), atol=1e-5)


def test_mt_full_size_b50_matches_oracle_scalars():
    """MultitaskRePo at the headline batch shape (B=50, L=50, H=15, A=6, 3 tasks; `bench.py --config mt`): one update
    against the CPU oracle -- scalars within 1e-3, pre-clip gradient norms within 2e-3."""
    L, B, H, A, C = 50, 50, 15, 6, 3
    agent, cfg = make_mt("repo_multitask", L, B, H, A, C)
    oracle = ro.OracleMultitask(cfg, A, C, seed=7)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    batch, host = dev_batch(L, B, A, 2468)
    tasks, htasks = dev_tasks(L, B, C, 2468)
    agent.noise_source, nz = dev_noise(L, B, H, A, 99)
    agent.update((tasks, *batch))
    got = dict(agent.last_scalars)
    want = oracle.update(htasks, *host, nz)[2]
    for k, w in want.items():
        r = abs(got[k] - w) / (abs(w) + 1e-12)
        log(f"[mt B=50] {k}: got {got[k]:.7g} oracle {w:.7g} rel {r:.2e}")
        assert r < 1e-3, (k, got[k], w)
    for name in ("model", "actor", "value"):
        w = oracle.last[f"{name}_total_norm"]
        assert abs(agent.last_grad_norms[name] - w) < 2e-3 * w, (name, agent.last_grad_norms[name], w)


class FakeMtEnv:
    """Three tasks, episodes of 7 steps, random frames; the reference's MultitaskEnv surface (environments/mt_env.py)."""

    def __init__(self, A, seed=0):
        self.observation_space = Env(A).observation_space
        self.action_space = type("S", (), {"shape": (A,), "sample": lambda s: np.random.uniform(-1, 1, A).astype(np.float32)})()
        self._tasks = ["walk", "run", "stand"]
        self._ind, self._t, self.rs = None, 0, np.random.RandomState(seed)

    num_tasks = 3

    @property
    def task(self):
        return self._tasks[self._ind]

    @property
    def task_one_hot(self):
        v = np.zeros(3, dtype=np.float32)
        v[self._ind] = 1
        return v

    def sample_task(self, round_robin=False):
        nxt = 0 if self._ind is None else (self._ind + 1) % 3
        return self._tasks[nxt if round_robin else self.rs.randint(3)]

    def reset(self, task=None):
        self._ind = self._tasks.index(task if task is not None else self.sample_task())
        self._t = 0
        return self.rs.randint(0, 256, (3, 64, 64)).astype(np.uint8)

    def step(self, action):
        assert np.asarray(action).shape == (6,)
        self._t += 1
        return self.rs.randint(0, 256, (3, 64, 64)).astype(np.uint8), float(self._ind), self._t >= 7, {"success": self._ind == 1}


@pytest.mark.loops
def test_mt_train_eval_loops_buffer_and_checkpoint(tmp_path):
    """train() / eval_agent() end to end on a fake multitask environment: seed data with task labels, batches from the
    HBM mirror of the five-field ring equal to host sampling, per-task logging in the reference's keys, acting path
    (HIP graph) equal to the eager modules, checkpoint round trip incl. the per-task log_beta vector."""
    from repo_amd.algorithms.repo import MultitaskRePo
    from repo_amd.common.utils import set_gpu_mode

    set_gpu_mode(True)
    A = 6
    cfg = fx.default_config(algo="repo_multitask", batch_size=3, chunk_size=5, horizon=4, share_repr=False, prefill=30,
                            num_steps=16, train_every=8, eval_every=16, checkpoint_every=16, log_every=8, train_steps=2,
                            replay_size=200, action_noise=0.3)
    logger = Logger()
    logger.dir = str(tmp_path)
    np.random.seed(0)
    torch.manual_seed(0)
    agent = MultitaskRePo(cfg, FakeMtEnv(A, 1), FakeMtEnv(A, 2), logger)
    agent.train()
    assert len(agent.buffer) >= 30 + 16
    assert set(np.unique(agent.buffer.tasks[: len(agent.buffer)].sum(1))) == {1.0}
    assert not logger.nonfinite, logger.nonfinite     # nothing the loops EVER logged may be garbage
    assert np.isfinite(agent.buffer.actions[: len(agent.buffer)]).all()
    for k in ("train/beta_0", "train/beta_2", "train/kl_div", "train/obs_loss", "train/actor_loss"):
        assert k in logger.kv and math.isfinite(logger.kv[k]), k
    assert any(k.startswith("test/return_") for k in logger.kv) and any(k.startswith("train/return_") for k in logger.kv)
    assert sum(k.startswith("test/video_") for k in logger.kv) == 3
    # device-mirror batches == host sampling, task labels included
    np.random.seed(5)
    want = agent.buffer.sample(3, 5)
    np.random.seed(5)
    got = agent.buffer.sample_to_device(3, 5, agent.device)
    torch.cuda.synchronize()
    assert len(got) == 5 and got[0].shape == (5, 3, 3)
    for g_, w in zip(got, want):
        assert np.array_equal(g_.cpu().numpy(), w.astype(g_.cpu().numpy().dtype))
    # acting path: graph replay == eager modules
    b, s, a = agent.init_latent_and_action()
    frame = torch.rand(1, 3, 64, 64, device="cuda") * 2 - 1
    task = torch.tensor([[0.0, 1.0, 0.0]], device="cuda")
    out_g = agent.update_latent_and_select_action(b, s, a, frame, task, False)
    with torch.no_grad():
        emb = agent.encoder(frame, task)
        outs = agent.transition_model.observe(b, s, a.unsqueeze(0), task.unsqueeze(0), emb.unsqueeze(0))
    assert out_g[0].shape == (1, 200) and out_g[2].shape == (1, A)
    assert torch.isfinite(out_g[2]).all() and outs[0].shape == (1, 1, 200)
    # checkpoint round trip
    sd = agent.get_param_dict()
    assert sd["log_beta"].shape == (3,) and sd["log_beta"].requires_grad
    assert "film.weight" in sd["encoder"] and sd["transition_model"]["fc_embed_state_action.weight"].shape == (200, 39)
    torch.save(sd, os.path.join(tmp_path, "models.pt"))
    other = MultitaskRePo(cfg, FakeMtEnv(A, 1), FakeMtEnv(A, 2), Logger())
    other.load_param_dict(torch.load(os.path.join(tmp_path, "models.pt"), map_location="cuda", weights_only=False))
    assert torch.equal(other.log_beta, agent.log_beta) and other.beta_optimizer.step_count == agent.beta_optimizer.step_count
    assert torch.equal(other.model_optimizer.flat, agent.model_optimizer.flat)
    assert torch.equal(other.model_optimizer.exp_avg_sq, agent.model_optimizer.exp_avg_sq)
