"""Parity of the fused RSSM / head / loss / optimiser kernels against the CPU oracle
(oracle/repo_oracle.py, itself pinned to the reference by tests/test_oracle_golden.py).

Tolerances: forward values 1e-5 relative to the largest element (fp32 accumulation order
differs); gradients 1e-4 normwise (49-step BPTT compounds rounding).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures as fx
from oracle import repo_oracle as ro
from tests.util import l2err, log, relerr, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _poison_lds():
    """Start every test from NaN-filled LDS on all CUs: reads of never-written LDS cannot hide."""
    from repo_amd._lib import lib

    assert lib().repo_debug_poison_lds(torch.cuda.current_stream().cuda_stream) == 0
    yield

FTOL = 1e-5
GTOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from repo_amd import ops as o

    return o


def tparams(mod, A, seed=7, requires_grad=True):
    p = fx.make_params(A, seed)[mod]
    return {k: torch.tensor(v, requires_grad=requires_grad) for k, v in p.items()}


def cu(d):
    return [v.detach().cuda().contiguous() for v in d.values()]


@pytest.mark.parametrize("T,B,A", [(7, 4, 6), (5, 3, 7), (49, 16, 6), (3, 130, 6), (6, 51, 6)])
def test_observe_fwd_bwd(ops, T, B, A):
    rs = np.random.RandomState(T * 100 + B)
    D, S, E = 200, 30, 1024
    p = tparams("transition_model", A)
    actions = rnd(rs, T, B, A)
    nonterms = torch.from_numpy((rs.uniform(size=(T, B, 1)) > 0.2).astype(np.float32))
    embeds = F.relu(rnd(rs, T, B, E)).requires_grad_(True)
    e1, e2 = rnd(rs, T, B, S), rnd(rs, T, B, S)
    b0, s0 = rnd(rs, B, D, scale=0.3), rnd(rs, B, S)
    outs = ro.observe(p, b0, s0, actions, embeds, nonterms, e1, e2)
    sv = ops.rssm_observe_fwd(cu(p), b0.cuda(), s0.cuda(), actions.cuda(), nonterms.cuda(), embeds.detach().cuda(),
                              e1.cuda(), e2.cuda())
    got = [sv.featx[1:, :, :D], sv.prior_state, sv.prior_mean, sv.prior_std, sv.featx[1:, :, D:], sv.post_mean,
           sv.post_std]
    names = ["beliefs", "prior_states", "prior_means", "prior_stds", "post_states", "post_means", "post_stds"]
    for n, g, w in zip(names, got, outs):
        e = relerr(g, w)
        log(f"observe T={T} B={B} {n}: {e:.2e}")
        assert e < FTOL, n
    # backward: random upstream gradients on every output
    ups = [rnd(rs, *o.shape, scale=0.1) for o in outs]
    loss = sum((o * u).sum() for o, u in zip(outs, ups))
    loss.backward()
    dparams = [torch.zeros_like(v).cuda() for v in p.values()]
    dfeat = torch.cat([ups[0], ups[4]], dim=2).cuda().contiguous()
    dembeds = torch.empty(T, B, E).cuda()
    ops.rssm_observe_bwd(cu(p), sv, dparams, dfeat=dfeat, dprior_state=ups[1].cuda(), dpm=ups[2].cuda(),
                         dps=ups[3].cuda(), dqm=ups[5].cuda(), dqs=ups[6].cuda(), dembeds=dembeds)
    for (k, v), g in zip(p.items(), dparams):
        e = l2err(g, v.grad)
        log(f"observe bwd T={T} B={B} d{k}: {e:.2e}")
        assert e < GTOL, k
    e = l2err(dembeds, embeds.grad)
    log(f"observe bwd T={T} B={B} dembeds: {e:.2e}")
    assert e < GTOL


# rows >= 16384 run 32-row tiles (csrc/mlp16.hip); 16400 and 17001 leave a ragged last tile.  rows >= 4096: the weight
# gradients of the 200-wide layers take the direct kernel (csrc/wgrad_direct.h): 4097 and 17001 end on a single row
# (a half k-step), 4100 / 64 row ranges leaves the last ranges empty
@pytest.mark.parametrize("mod,L,rows", [("reward_model", 4, 333), ("actor_model", 5, 1000), ("value_model", 4, 64),
                                        ("value_model", 4, 16400), ("actor_model", 5, 17001), ("reward_model", 4, 1),
                                        ("value_model", 4, 4097), ("actor_model", 5, 4100)])
def test_mlp_fwd_bwd(ops, mod, L, rows):
    A = 6
    rs = np.random.RandomState(rows)
    p = tparams(mod, A)
    feat = rnd(rs, rows, 230).requires_grad_(True)
    want = ro.mlp_head(p, feat[:, :200], feat[:, 200:], L)
    out, hid = ops.mlp_fwd(cu(p), feat.detach().cuda())
    assert relerr(out, want) < FTOL
    up = rnd(rs, *want.shape)
    (want * up).sum().backward()
    dparams = [torch.full_like(v, 7.0).cuda() for v in p.values()]
    dx = torch.ones(rows, 230).cuda()
    ops.mlp_bwd(cu(p), feat.detach().cuda(), hid, up.cuda(), dparams=dparams, dx=dx, accumulate_dx=True)
    for (k, v), g in zip(p.items(), dparams):
        e = l2err(g, v.grad)
        log(f"mlp {mod} d{k}: {e:.2e}")
        assert e < GTOL
    assert l2err(dx, feat.grad + 1) < GTOL
    # frozen weights / detached input variants run
    ops.mlp_bwd(cu(p), feat.detach().cuda(), hid, up.cuda(), dparams=None, dx=dx)
    assert l2err(dx, feat.grad) < GTOL
    dparams = [torch.full_like(v, 7.0).cuda() for v in p.values()]
    ops.mlp_bwd(cu(p), feat.detach().cuda(), hid, up.cuda(), dparams=dparams, dx=None)
    for (k, v), g in zip(p.items(), dparams):
        assert l2err(g, v.grad) < GTOL, k


@pytest.mark.parametrize("rows,rows_w", [(34300, 31850), (700, 650), (4100, 4100), (17, 3), (20011, 1)])
def test_scalar_head_two_upstream_gradients_through_one_chain(ops, rows, rows_w):
    """ops.mlp_bwd(dout_w=...): the value head differentiated for the actor's loss (input gradient, all rows) and the
    critic's loss (weight gradients, the first rows_w rows) in ONE reverse chain -- against autograd of the two losses, and
    against the two separate passes it replaces (dreamer.py:343-373 of the reference)."""
    rs = np.random.RandomState(rows + rows_w)
    p = tparams("value_model", 6)
    feat = rnd(rs, rows, 230).requires_grad_(True)
    want = ro.mlp_head(p, feat[:, :200], feat[:, 200:], 4)
    up_x, up_w = rnd(rs, rows, 1, scale=1e-3), rnd(rs, rows_w, 1, scale=1e-4)
    up_x[::7] = 0.0                                   # rows whose actor-loss gradient is exactly zero keep their critic share
    gx, = torch.autograd.grad((want * up_x).sum(), feat, retain_graph=True)
    gw = torch.autograd.grad((want[:rows_w] * up_w).sum(), list(p.values()))
    fd = feat.detach().cuda()
    _, hid = ops.mlp_fwd(cu(p), fd)
    dparams = [torch.full_like(v, 7.0).cuda() for v in p.values()]
    dx = torch.full((rows, 230), 3.0).cuda()
    ops.mlp_bwd(cu(p), fd, hid, up_x.cuda(), dparams=dparams, dx=dx, dout_w=up_w.cuda())
    for (k, v), g, w in zip(p.items(), dparams, gw):
        e = l2err(g, w)
        log(f"mlp one chain {rows}/{rows_w} d{k}: {e:.2e}")
        assert e < GTOL, (k, e)
    assert l2err(dx, gx) < GTOL
    # the two passes: same numbers up to rounding
    dp2 = [torch.zeros_like(v).cuda() for v in p.values()]
    dx2 = torch.empty(rows, 230).cuda()
    ops.mlp_bwd(cu(p), fd, hid, up_x.cuda(), dparams=None, dx=dx2)
    ops.mlp_bwd(cu(p), fd[:rows_w], [h[:rows_w] for h in hid], up_w.cuda(), dparams=dp2, dx=None)
    assert l2err(dx, dx2) < 1e-6
    for g, g2 in zip(dparams, dp2):
        assert l2err(g, g2) < 2e-6
    # accumulate_dx adds the scaled chain
    dx3 = torch.full((rows, 230), 3.0).cuda()
    ops.mlp_bwd(cu(p), fd, hid, up_x.cuda(), dparams=[torch.zeros_like(v).cuda() for v in p.values()], dx=dx3,
                accumulate_dx=True, dout_w=up_w.cuda())
    assert l2err(dx3, gx + 3.0) < GTOL


@pytest.mark.parametrize("mod,L,rows", [("value_model", 4, 20011), ("actor_model", 5, 9000)])
def test_head_weight_gradients_on_the_bf16_pipe_match_fp64_and_the_fp32_engine(ops, mod, L, rows):
    """The dense heads' weight gradients at update-like row counts (csrc/wgrad_tr.h: both operands staged as they lie, split
    exactly into three bf16 each, fragments by the LDS's transposing read) against an fp64 backward of the same head -- and
    against the fp32-MFMA kernel (csrc/wgrad_direct.h, repo_debug_bgemm(0)) on the SAME operands: the six-product split may
    not be less accurate than fp32 arithmetic by more than 25 % (measured: at or below it).  20011 rows: 42 row ranges of
    477 rows, the last block of every range ragged, the last range short."""
    from repo_amd._lib import lib

    A = 6
    rs = np.random.RandomState(rows)
    p = tparams(mod, A)
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in p.items()}
    feat = rnd(rs, rows, 230)
    feat[::5] *= 8.0
    want = ro.mlp_head(p64, feat.double()[:, :200], feat.double()[:, 200:], L)
    up = rnd(rs, *want.shape)
    (want * up.double()).sum().backward()
    _, hid = ops.mlp_fwd(cu(p), feat.cuda())
    errs = {}
    for engine in (1, 0):
        prev = lib().repo_debug_bgemm(engine)
        try:
            dparams = [torch.zeros_like(v).cuda() for v in p.values()]
            ops.mlp_bwd(cu(p), feat.cuda(), hid, up.cuda(), dparams=dparams, dx=None)
        finally:
            lib().repo_debug_bgemm(prev)
        errs[engine] = max(l2err(g, v.grad) for (k, v), g in zip(p64.items(), dparams) if v.dim() == 2 and v.shape[0] == 200)
        for (k, v), g in zip(p64.items(), dparams):
            assert l2err(g, v.grad) < GTOL, (engine, k)
    log(f"head weight gradients {mod} rows={rows}: max l2 err of the 200-wide layers  bf16x6 {errs[1]:.2e}  fp32 MFMA {errs[0]:.2e}")
    assert errs[1] <= 1.25 * errs[0] + 1e-8, errs


@pytest.mark.parametrize("case", ["odd_shape", "misaligned_hidden"])
def test_mlp_layer_by_layer_path(ops, case):
    """Shapes outside the fused kernel's instantiation (csrc/mlp16.hip) and hidden buffers that are not 16-byte
    aligned run layer by layer on the GEMM engine: same results."""
    rs = np.random.RandomState(5)
    if case == "odd_shape":
        rows, dims = 77, [50, 64, 64, 3]
    else:
        rows, dims = 300, [230, 200, 200, 200, 1]
    L = len(dims) - 1
    p = {}
    for i in range(L):
        p[f"fc{i + 1}.weight"] = rnd(rs, dims[i + 1], dims[i], scale=dims[i] ** -0.5).requires_grad_(True)
        p[f"fc{i + 1}.bias"] = rnd(rs, dims[i + 1], scale=0.1).requires_grad_(True)
    x = rnd(rs, rows, dims[0]).requires_grad_(True)
    want = ro.mlp_head(p, x[:, :dims[0] - 1], x[:, dims[0] - 1:], L)
    hid = None
    if case == "misaligned_hidden":  # views that start one float into their allocation
        hid = [torch.empty(rows * dims[i + 1] + 1, device="cuda")[1:].view(rows, dims[i + 1]) for i in range(L - 1)]
    out, hid = ops.mlp_fwd(cu(p), x.detach().cuda(), hid=hid)
    assert relerr(out, want) < FTOL
    up = rnd(rs, *want.shape)
    (want * up).sum().backward()
    dparams = [torch.full_like(v, 3.0).cuda() for v in p.values()]
    dx = torch.empty(rows, dims[0]).cuda()
    ops.mlp_bwd(cu(p), x.detach().cuda(), hid, up.cuda(), dparams=dparams, dx=dx)
    for (k, v), g in zip(p.items(), dparams):
        assert l2err(g, v.grad) < GTOL, k
    assert l2err(dx, x.grad) < GTOL


@pytest.fixture(params=[1, 0], ids=["rowtile32", "rowtile16"])
def rollout_engine(request):
    """Both persistent rollout engines: 32-row tiles on the bf16 matrix pipe (csrc/imagine32.hip, the default) and
    16-row tiles on the fp32 MFMA (csrc/imagine16.hip)."""
    from repo_amd._lib import lib

    prev = lib().repo_debug_rowtile32(request.param)
    yield request.param
    lib().repo_debug_rowtile32(prev)


@pytest.mark.parametrize("Hm,N,A", [(4, 28, 6), (14, 300, 6), (2, 15, 7)])
def test_imagine_fwd_bwd(ops, rollout_engine, Hm, N, A):
    rs = np.random.RandomState(Hm * 10 + N)
    D, S = 200, 30
    rp = tparams("transition_model", A, requires_grad=False)
    ap = tparams("actor_model", A)
    b0 = rnd(rs, N, D, scale=0.3).requires_grad_(True)
    s0 = rnd(rs, N, S).requires_grad_(True)
    ea, ep = rnd(rs, Hm, N, A), rnd(rs, Hm, N, S)
    ib, istate, im, isd = ro.imagine(rp, ap, b0, s0, Hm + 1, ea, ep)
    sv = ops.rssm_imagine_fwd(cu(rp), cu(ap), b0.detach().cuda(), s0.detach().cuda(), ea.cuda(), ep.cuda())
    for n, g, w in [("beliefs", sv.featx[1:, :, :D], ib), ("states", sv.featx[1:, :, D:], istate),
                    ("means", sv.prior_mean, im), ("stds", sv.prior_std, isd)]:
        e = relerr(g, w)
        log(f"imagine Hm={Hm} N={N} {n}: {e:.2e}")
        assert e < FTOL
    # everything the forward saves for the backward is an output of the entry point: the actor's raw head outputs
    # (2A columns: a ragged last quad for A = 7) and its hidden activations against the oracle's MLP on the same rows
    feat_all = sv.featx[:Hm].reshape(Hm * N, D + S).cpu()
    with torch.no_grad():
        raw_want = ro.mlp_head({k: v.detach() for k, v in ap.items()}, feat_all[:, :D], feat_all[:, D:], 5)
    e = relerr(sv.a_raw[:Hm * N], raw_want)
    log(f"imagine Hm={Hm} N={N} A={A} saved actor raw outputs: {e:.2e}")
    assert e < FTOL
    ub, us, um, usd = (rnd(rs, *x.shape, scale=0.1) for x in (ib, istate, im, isd))
    ((ib * ub).sum() + (istate * us).sum() + (im * um).sum() + (isd * usd).sum()).backward()
    dfeat = torch.cat([ub, us], dim=2).cuda().contiguous()
    d_araw, dfeat0 = ops.rssm_imagine_bwd(cu(rp), sv, dfeat, dprior_mean=um.cuda(), dprior_std=usd.cuda(),
                                          want_dfeat0=True)
    e1, e2 = l2err(dfeat0[:, :D], b0.grad), l2err(dfeat0[:, D:], s0.grad)
    log(f"imagine bwd Hm={Hm} N={N}: dbelief0 {e1:.2e} dstate0 {e2:.2e}")
    assert e1 < GTOL and e2 < GTOL
    # deferred actor backward over all steps
    dap = [torch.zeros_like(v).cuda() for v in ap.values()]
    x = sv.featx[:Hm].reshape(Hm * N, D + S)
    hid = [sv.a_hidden[l] for l in range(sv.a_hidden.shape[0])]
    ops.mlp_bwd(cu(ap), x, hid, d_araw, dparams=dap, dx=None)
    for (k, v), g in zip(ap.items(), dap):
        e = l2err(g, v.grad)
        log(f"imagine bwd Hm={Hm} N={N} actor d{k}: {e:.2e}")
        assert e < GTOL


def test_rollout_engines_agree_at_full_size_with_in_kernel_noise_and_condition(ops):
    """The 32-row bf16x6 rollout (forward and reverse) against the 16-row fp32-MFMA one at the update's size (2450 start
    states, a ragged last tile of 18 rows, 14 steps), noise drawn in the kernels from the same Philox stream, with and
    without a condition: every saved tensor and both gradients agree to fp32 rounding."""
    from repo_amd._lib import lib

    rs = np.random.RandomState(5)
    Hm, N, A, D, S = 14, 2450, 6, 200, 30
    for C in (0, 3):
        P = fx.make_params(A, 7, cond=C)
        rp = [torch.tensor(v).cuda() for v in P["transition_model"].values()]
        ap = [torch.tensor(v).cuda() for v in P["actor_model"].values()]
        b0, s0 = rnd(rs, N, D, scale=0.3).cuda(), rnd(rs, N, S).cuda()
        cond = None
        if C:
            cond = torch.zeros(N, C)
            cond[torch.arange(N), torch.from_numpy(rs.randint(0, C, size=N))] = 1.0
            cond = cond.cuda()
        dfeat = rnd(rs, Hm, N, D + S, scale=0.01).cuda()
        dpm, dps = rnd(rs, Hm, N, S, scale=0.01).cuda(), rnd(rs, Hm, N, S, scale=0.01).cuda()
        got = {}
        for engine in (0, 1):
            prev = lib().repo_debug_rowtile32(engine)
            try:
                sv = ops.rssm_imagine_fwd(rp, ap, b0, s0, None, None, noise=(77, 1 << 20), horizon=Hm, cond=cond)
                d_araw, dfeat0 = ops.rssm_imagine_bwd(rp, sv, dfeat, dpm, dps, want_dfeat0=True)
                torch.cuda.synchronize()
            finally:
                lib().repo_debug_rowtile32(prev)
            got[engine] = {n: getattr(sv, n).clone() for n in ("featx", "prior_mean", "prior_std", "a_hidden", "a_raw",
                                                                "a_mean", "a_std", "xsa", "e", "gates", "hp")}
            got[engine].update(d_araw=d_araw.clone(), dfeat0=dfeat0.clone())
        for name, want in got[0].items():
            x = got[1][name]
            assert torch.isfinite(x).all(), (C, name)
            e = (x - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
            log(f"rollout engines C={C} {name}: max |32-row - 16-row| / max |x| = {e:.2e}")
            assert e < (2e-5 if name in ("d_araw", "dfeat0") else 1e-5), (C, name, e)


def test_actor_head_and_entropy(ops):
    rs = np.random.RandomState(2)
    rows, A, NS = 500, 6, 100
    raw = rnd(rs, rows, 2 * A, scale=2.0)
    raw[0, :A] = 40.0  # saturate: tanh(u) == 1 -> clamp branch, zero gradient through x
    raw[1, :A] = -40.0
    rawt = raw.clone().requires_grad_(True)
    mean = 5.0 * torch.tanh(rawt[:, :A] / 5.0)
    std = F.softplus(rawt[:, A:]) + 0.1
    eps = rnd(rs, NS, rows, A)
    ent = ro.tanh_normal_entropy(mean, std, eps)
    ent.sum().backward()
    m, s, _ = ops.actor_head_fwd(raw.cuda())
    assert relerr(m, mean) < FTOL and relerr(s, std) < FTOL
    out, dm, ds = ops.tanh_normal_entropy(m, s, eps.cuda(), gscale=1.0)
    e = abs(out.item() - ent.sum().item()) / abs(ent.sum().item())
    log(f"tanh_normal_entropy sum: {e:.2e}")
    assert e < 1e-4
    draw = ops.actor_head_bwd(m, s, dmean=dm, dstd=ds)
    e = l2err(draw, rawt.grad)
    log(f"tanh_normal_entropy d raw: {e:.2e}")
    assert e < 2e-3  # 100-sample sums of tanh/atanh round trips: libm differences at |u| > 5


def test_losses(ops):
    rs = np.random.RandomState(3)
    rows, S = 343, 30
    pm, qm = rnd(rs, rows, S).requires_grad_(True), rnd(rs, rows, S).requires_grad_(True)
    ps = (rnd(rs, rows, S).abs() + 0.1).requires_grad_(True)
    qs = (rnd(rs, rows, S).abs() + 0.1).requires_grad_(True)
    lb = torch.tensor(math.log(0.3))
    alpha = 5 / 6
    klp = ro.normal_kl(qm.detach(), qs.detach(), pm, ps).sum(1).mean()
    klq = ro.normal_kl(qm, qs, pm.detach(), ps.detach()).sum(1).mean()
    (lb.exp() * (alpha * klp + (1 - alpha) * klq)).backward()
    out, g = ops.kl_balance(pm.detach().cuda(), ps.detach().cuda(), qm.detach().cuda(), qs.detach().cuda(), 0, alpha,
                            lb.cuda(), 3.0, 1.0 / rows)
    assert abs(out.item() / rows - klp.item()) < 1e-5 * abs(klp.item())
    for gg, t in zip(g, (pm, ps, qm, qs)):
        assert l2err(gg, t.grad) < 1e-5
    for t in (pm, ps, qm, qs):
        t.grad = None
    # Dreamer free-nats variant: make a third of the rows fall under the threshold
    with torch.no_grad():
        qm[::3] = pm[::3]
        qs[::3] = ps[::3]
    kl = ro.normal_kl(qm, qs, pm, ps).sum(1)
    fn = 3.0
    torch.max(kl, torch.full((1,), fn)).mean().backward()
    out, g = ops.kl_balance(pm.detach().cuda(), ps.detach().cuda(), qm.detach().cuda(), qs.detach().cuda(), 1, 0.0,
                            None, fn, 1.0 / rows)
    assert abs(out.item() / rows - torch.max(kl, torch.full((1,), fn)).mean().item()) < 1e-5 * fn
    for gg, t in zip(g, (pm, ps, qm, qs)):
        assert l2err(gg, t.grad) < 1e-5
    # scalar nll
    pred, tgt = rnd(rs, 777), rnd(rs, 777)
    mask = torch.from_numpy((rs.uniform(size=777) > 0.3).astype(np.float32))
    sums, dp = ops.scalar_nll(pred.cuda(), tgt.cuda(), mask.cuda(), 0.5)
    assert relerr(sums, torch.stack([(0.5 * (pred - tgt) ** 2 * mask).sum(), mask.sum()])) < 1e-5
    assert relerr(dp, (pred - tgt) * mask * 0.5) < 1e-6
    # normal entropy
    sd = rnd(rs, 500, 30).abs() + 0.1
    out, dsd = ops.normal_entropy(sd.cuda(), gscale=2.0, want_grad=True)
    want = (0.5 + 0.5 * math.log(2 * math.pi) + sd.log()).sum()
    assert abs(out.item() - want.item()) < 1e-5 * abs(want.item())
    assert relerr(dsd, 2.0 / sd) < 1e-6
    # lambda return
    Hm, N = 14, 321
    r, v = rnd(rs, Hm, N).requires_grad_(True), rnd(rs, Hm, N).requires_grad_(True)
    disc = 0.99 * torch.ones(Hm, N)
    ret = ro.lambda_return(r[:-1], v[:-1], disc[:-1], v[-1], 0.95)
    gret = -1.0 / ret.numel()
    (gret * ret.sum()).backward()
    returns, dr, dv, rsum = ops.lambda_return(r.detach().cuda(), v.detach().cuda(), 0.99, 0.95, gret)
    assert relerr(returns, ret) < 1e-5
    assert relerr(dr, r.grad) < 1e-5 and relerr(dv, v.grad) < 1e-5
    assert abs(rsum.item() - ret.sum().item()) < 1e-4 * abs(ret.sum().item()) + 1e-3
    # dual step
    lbp = torch.tensor([math.log(1e-5)]).cuda()
    m, vv = torch.zeros(1).cuda(), torch.zeros(1).cuda()
    klsum = torch.tensor([0.2 * 100]).cuda()
    sc = ops.dual_step(lbp, m, vv, klsum, 100, 3.0, 1e-4, 1)
    ref = torch.tensor(math.log(1e-5), requires_grad=True)
    opt = ro.Adam([ref], 1e-4)
    (-ref * (0.2 - 3.0)).backward()
    opt.step()
    assert abs(lbp.item() - ref.item()) < 1e-6
    assert abs(sc[0].item() - 0.2) < 1e-6 and abs(sc[2].item() - (-math.log(1e-5) * (0.2 - 3.0))) < 1e-4


def test_clip_adam(ops):
    rs = np.random.RandomState(4)
    n = 100003
    p0 = rnd(rs, n)
    ref = p0.clone().requires_grad_(True)
    opt = ro.Adam([ref], 3e-4)
    p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step in range(1, 4):
        g = rnd(rs, n, scale=5.0 if step == 2 else 0.01)  # step 2 exceeds max_norm=100 -> clipped
        ref.grad = g.clone()
        total = ro.clip_grad_norm([ref], 100.0)
        opt.step()
        gc = g.cuda()
        sq = ops.grad_sqnorm(gc)
        assert abs(math.sqrt(sq.item()) - float(total)) < 1e-5 * float(total)
        ops.clip_adam(p, gc, m, v, sq, 100.0, 3e-4, step)
        e = (p.cpu() - ref.detach()).abs().max().item()
        log(f"clip_adam step {step}: max abs diff {e:.2e}")
        assert e < 2e-7


# ----------------------------------------------------------------------------- in-kernel Philox noise (8b)
def test_philox_normal_stream_statistics_and_addressing():
    """repo_philox_normal: out[i] is a pure function of (seed, offset + i); moments of N(0,1); distinct seeds and
    disjoint offset ranges are uncorrelated."""
    from repo_amd import ops

    dev = torch.device("cuda")
    n = 1 << 22
    a = ops.philox_normal(n, 1234, 0, dev)
    b = ops.philox_normal(n // 2, 1234, 1001, dev)
    assert torch.equal(a[1001:1001 + n // 2], b)                      # offset addressing, unaligned to the block of 4
    assert torch.equal(ops.philox_normal(n, 1234, 0, dev), a)        # deterministic
    x = a.double()
    assert abs(x.mean().item()) < 3e-3 and abs(x.var().item() - 1.0) < 5e-3
    assert abs((x ** 3).mean().item()) < 1e-2 and abs((x ** 4).mean().item() - 3.0) < 3e-2
    assert x.abs().max().item() < 6.5 and torch.isfinite(a).all()
    c = ops.philox_normal(n, 1235, 0, dev).double()
    assert abs((x * c).mean().item()) < 3e-3                          # other seed
    assert abs((x[:-1] * x[1:]).mean().item()) < 3e-3                 # lag-1
    assert abs((x[:-4] * x[4:]).mean().item()) < 3e-3                 # across counter blocks
    frac = (x.abs() < 1.0).double().mean().item()
    assert abs(frac - 0.682689) < 2e-3


@pytest.mark.parametrize("A,fused", [(6, True), (7, False)])
def test_update_with_in_kernel_noise_equals_explicit_tensors(A, fused):
    """An update whose kernels DRAW their noise (null eps + (seed, offset)) equals, bit for bit, the update fed the
    tensors repo_philox_normal materialises for the same (seed, offset) ranges -- observe fwd/bwd, the rollout
    (fused engine at A=6, per-step engine at A=7) fwd/bwd and the 100-sample entropy."""
    from repo_amd import ops
    from tests.test_update_gpu import dev_batch, make_agent

    L, B, H = 7, 5, 5
    T, S, Hm = L - 1, 30, H - 1
    N = T * B
    batch, _ = dev_batch(L, B, A, 77)
    drawn, _ = make_agent("repo", L, B, H, A)
    fed, _ = make_agent("repo", L, B, H, A)
    seed = 987654321
    drawn._noise_seed = seed
    dev = torch.device("cuda")
    for u in range(2):
        off = drawn._noise_counter
        o_obs, o_img, o_ent = off, off + 2 * T * B * S, off + 2 * T * B * S + Hm * N * (A + S)
        fed.noise_source = {
            "obs_prior": ops.philox_normal(T * B * S, seed, o_obs, dev).view(T, B, S),
            "obs_post": ops.philox_normal(T * B * S, seed, o_obs + T * B * S, dev).view(T, B, S),
            "img_act": ops.philox_normal(Hm * N * A, seed, o_img, dev).view(Hm, N, A),
            "img_prior": ops.philox_normal(Hm * N * S, seed, o_img + Hm * N * A, dev).view(Hm, N, S),
            # sample-fastest: element e's 100 draws are consecutive
            "entropy": ops.philox_normal(100 * Hm * N * A, seed, o_ent, dev).view(Hm * N, A, 100).permute(2, 0, 1).contiguous(),
        }
        drawn.update(batch)
        fed.update(batch)
        assert drawn._noise_counter == o_ent + 100 * Hm * N * A
        assert drawn.last_scalars == fed.last_scalars, (u, drawn.last_scalars, fed.last_scalars)
        for name in ("model_optimizer", "actor_optimizer", "value_optimizer"):
            assert torch.equal(getattr(drawn, name).flat, getattr(fed, name).flat), (u, name)


def test_full_size_pipelined_in_kernel_noise_equals_explicit_tensors_20_updates():
    """BASELINE config 2 shapes (B=50, L=50, H=15): 20 PIPELINED updates (update(join=False), as train_agent()
    issues them) whose kernels draw their Philox noise equal, bit for bit, 20 updates fed the materialised
    tensors of the same (seed, offset) ranges: scalars of every update and all parameters at the end."""
    from repo_amd import ops
    from tests.test_update_gpu import dev_batch, make_agent

    L, B, H, A = 50, 50, 15, 6
    T, S, Hm = L - 1, 30, H - 1
    N = T * B
    dev = torch.device("cuda")
    batches = [dev_batch(L, B, A, 500 + i)[0] for i in range(3)]
    drawn, _ = make_agent("repo", L, B, H, A)
    fed, _ = make_agent("repo", L, B, H, A)
    seed = 20261003
    drawn.seed_noise(seed)
    per_update = 2 * T * B * S + Hm * N * (A + S) + 100 * Hm * N * A
    s_drawn, s_fed = [], []
    for u in range(20):
        drawn.update(batches[u % 3], join=False)
        s_drawn.append(dict(drawn.last_scalars) if u % 5 == 4 else None)  # reading scalars syncs: only now and then
    drawn.synchronize()
    assert drawn._noise_counter == 20 * per_update
    for u in range(20):
        off = u * per_update
        o_img, o_ent = off + 2 * T * B * S, off + 2 * T * B * S + Hm * N * (A + S)
        fed.noise_source = {
            "obs_prior": ops.philox_normal(T * B * S, seed, off, dev).view(T, B, S),
            "obs_post": ops.philox_normal(T * B * S, seed, off + T * B * S, dev).view(T, B, S),
            "img_act": ops.philox_normal(Hm * N * A, seed, o_img, dev).view(Hm, N, A),
            "img_prior": ops.philox_normal(Hm * N * S, seed, o_img + Hm * N * A, dev).view(Hm, N, S),
            "entropy": ops.philox_normal(100 * Hm * N * A, seed, o_ent, dev).view(Hm * N, A, 100).permute(2, 0, 1).contiguous(),
        }
        fed.update(batches[u % 3], join=False)
        s_fed.append(dict(fed.last_scalars) if u % 5 == 4 else None)
    fed.synchronize()
    torch.cuda.synchronize()
    assert s_drawn == s_fed
    assert all(np.isfinite(v) for v in s_fed[-1].values())
    for name in ("model_optimizer", "actor_optimizer", "value_optimizer"):
        assert torch.equal(getattr(drawn, name).flat, getattr(fed, name).flat), name
    assert float(drawn.log_beta) == float(fed.log_beta)


def test_observe_with_grad_inputs_and_no_observations_raises():
    """observe(observations=None) is forward-only here; the reference's branch is differentiable (rssm.py:112-146):
    a caller that would backpropagate through it is told, instead of getting detached tensors silently."""
    from tests.test_update_gpu import make_agent

    agent, _ = make_agent("repo", 8, 4, 5, 6)
    b0 = torch.zeros(2, 200, device="cuda", requires_grad=True)
    s0 = torch.zeros(2, 30, device="cuda")
    act = torch.zeros(3, 2, 6, device="cuda")
    with pytest.raises(NotImplementedError):
        agent.transition_model.observe(b0, s0, act, None, None)
    with torch.no_grad():
        assert len(agent.transition_model.observe(b0, s0, act, None, None)) == 4
    assert len(agent.transition_model.observe(b0.detach(), s0, act, None, None)) == 4


def test_observe_without_observations_prior_only_rollout():
    """TransitionModel.observe(observations=None) (reference rssm.py:112-146): open-loop rollout under given actions,
    the next step fed the nonterminal-masked PRIOR sample; four outputs."""
    from tests.test_update_gpu import make_agent

    A, T, B = 6, 9, 5
    agent, cfg = make_agent("repo", 8, 4, 5, A)
    p = ro.OracleAgent(cfg, A, seed=7).p["transition_model"]
    rs = np.random.RandomState(3)
    b0 = torch.from_numpy(rs.standard_normal((B, 200)).astype(np.float32) * 0.3)
    s0 = torch.from_numpy(rs.standard_normal((B, 30)).astype(np.float32))
    act = torch.from_numpy(rs.uniform(-1, 1, (T, B, A)).astype(np.float32))
    non = torch.ones(T, B, 1)
    non[3, 1] = 0
    non[5, 4] = 0
    eps = torch.from_numpy(rs.standard_normal((T, B, 30)).astype(np.float32))
    with torch.no_grad():
        bel, st, want = b0, s0, [[], [], [], []]
        for t in range(T):
            bel = ro.compute_belief(p, bel, st * non[t], act[t])
            smp, mean, std = ro.gaussian_head(p, "fc_embed_belief_prior", "fc_state_prior", bel, eps[t])
            st = smp
            for lst, v in zip(want, (bel, smp, mean, std)):
                lst.append(v)
        want = [torch.stack(w) for w in want]
    got = agent.transition_model.observe(b0.cuda(), s0.cuda(), act.cuda(), None, non.cuda(), noise=(eps.cuda(),))
    assert len(got) == 4
    for g, w, name in zip(got, want, ("beliefs", "prior_states", "prior_means", "prior_std_devs")):
        assert tuple(g.shape) == tuple(w.shape)
        np.testing.assert_allclose(g.cpu().numpy(), w.numpy(), rtol=2e-4, atol=2e-5, err_msg=name)


def test_prior_head_hoisted_out_of_the_scan_matches_in_scan_prior():
    """repo_rssm_observe_fwd(prior_only = 2) + repo_rssm_prior_head (the prior head of all steps as two GEMMs on a side
    stream) == the scan that evaluates the prior head step by step: same posterior path bit for bit, prior
    mean / std / sample and the saved hidden activation within GEMM-order rounding; the reverse scan runs on either."""
    from repo_amd import ops
    from tests.test_update_gpu import make_agent

    A, T, B = 6, 9, 5
    agent, cfg = make_agent("repo", 8, 4, 5, A)
    p = [t.detach() for t in agent.transition_model.plist()]
    rs = np.random.RandomState(11)
    dev = lambda a: torch.from_numpy(a.astype(np.float32)).cuda()  # noqa: E731
    b0, s0 = dev(rs.standard_normal((B, 200)) * 0.3), dev(rs.standard_normal((B, 30)))
    act, non = dev(rs.uniform(-1, 1, (T, B, A))), torch.ones(T, B).cuda()
    non[3, 1] = 0
    emb = dev(rs.standard_normal((T, B, 1024)) * 0.5)
    ep, eq = dev(rs.standard_normal((T, B, 30))), dev(rs.standard_normal((T, B, 30)))
    side = torch.cuda.Stream()
    for eps in ((ep, eq), (None, None)):  # explicit noise tensors / in-kernel Philox
        a = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, eps[0], eps[1], 0.1, noise=(77, 1000))
        h = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, eps[0], eps[1], 0.1, noise=(77, 1000), prior_stream=side)
        assert h.prior_ready is side and a.prior_ready is None
        torch.cuda.current_stream().wait_stream(side)
        for name in ("featx", "post_mean", "post_std", "hq", "gates", "e", "xsa"):
            assert torch.equal(getattr(a, name), getattr(h, name)), name
        for name in ("prior_mean", "prior_std", "prior_state", "hp"):
            np.testing.assert_allclose(getattr(h, name).cpu().numpy(), getattr(a, name).cpu().numpy(), rtol=2e-5, atol=2e-6,
                                       err_msg=name)


@pytest.mark.parametrize("T,B,A,philox", [(9, 3, 6, False), (49, 7, 6, True), (12, 16, 7, False), (6, 17, 6, True),
                                          (5, 50, 6, False)])
def test_column_split_scan_matches_row_scan(monkeypatch, T, B, A, philox):
    """csrc/scan_cs.hip (weight-stationary MFMA scan: 13 column-owner workgroups per 16 rows, two all-gathers per step)
    against rssm.hip's row scan: every saved tensor within MFMA-vs-FMA summation-order rounding, with explicit noise
    tensors and with in-kernel Philox noise; ragged row groups (17 = 16 + 1, 50 = 3 x 16 + 2) and A = 7 included."""
    from repo_amd import ops

    D, S, E = 200, 30, 1024
    rs = np.random.RandomState(1000 * T + B)
    p = cu(tparams("transition_model", A, requires_grad=False))
    dev = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()  # noqa: E731
    b0, s0 = dev(rs.standard_normal((B, D)) * 0.3), dev(rs.standard_normal((B, S)))
    act = dev(rs.uniform(-1, 1, (T, B, A)))
    non = dev(rs.uniform(size=(T, B)) > 0.15)
    emb = dev(np.maximum(rs.standard_normal((T, B, E)), 0))
    eps = (None, None) if philox else (dev(rs.standard_normal((T, B, S))), dev(rs.standard_normal((T, B, S))))
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("REPO_SCAN_CS", mode)
        out[mode] = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, eps[0], eps[1], 0.1, noise=(5, 64))
    torch.cuda.synchronize()
    worst = 0.0
    for name in ("featx", "post_mean", "post_std", "hq", "gates", "e", "xsa", "prior_mean", "prior_std", "prior_state", "hp"):
        a, b = getattr(out["0"], name), getattr(out["1"], name)
        assert torch.isfinite(b).all(), name
        err = float((a - b).abs().max() / (a.abs().max() + 1e-12))
        worst = max(worst, err)
        assert err < 2e-5, (name, err)
    log(f"column-split scan T={T} B={B} A={A} philox={philox}: worst rel-to-max error vs the row scan {worst:.2e}")


@pytest.mark.parametrize("T,B,A,philox", [(9, 3, 6, False), (49, 7, 6, True), (6, 17, 7, False), (5, 50, 6, True)])
def test_column_split_reverse_scan_matches_row_scan(monkeypatch, T, B, A, philox):
    """Forward + reverse scan on the column-split engine (csrc/scan_cs.hip) against rssm.hip's row scans: all 14
    parameter gradients, d embeds and the gradients into the carried belief / state, with every upstream gradient
    non-zero (KL terms, prior sample, decoder / reward gradients on [belief | state])."""
    from repo_amd import ops

    D, S, E = 200, 30, 1024
    rs = np.random.RandomState(2000 * T + B)
    p = cu(tparams("transition_model", A, requires_grad=False))
    dev = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()  # noqa: E731
    b0, s0 = dev(rs.standard_normal((B, D)) * 0.3), dev(rs.standard_normal((B, S)))
    act = dev(rs.uniform(-1, 1, (T, B, A)))
    non = dev(rs.uniform(size=(T, B)) > 0.15)
    emb = dev(np.maximum(rs.standard_normal((T, B, E)), 0))
    eps = (None, None) if philox else (dev(rs.standard_normal((T, B, S))), dev(rs.standard_normal((T, B, S))))
    ups = {k: dev(rs.standard_normal(shp) * 0.1) for k, shp in
           (("dfeat", (T, B, D + S)), ("dprior_state", (T, B, S)), ("dpm", (T, B, S)), ("dps", (T, B, S)),
            ("dqm", (T, B, S)), ("dqs", (T, B, S)))}
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("REPO_SCAN_CS", mode)
        sv = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, eps[0], eps[1], 0.1, noise=(9, 128))
        assert sv.cs == (mode == "1")
        g = [torch.zeros_like(t) for t in p]
        dembeds, dpb, dps_ = torch.empty(T, B, E).cuda(), torch.empty(B, D).cuda(), torch.empty(B, S).cuda()
        ops.rssm_observe_bwd(p, sv, g, dembeds=dembeds, dprev_belief=dpb, dprev_state=dps_, **ups)
        res[mode] = g + [dembeds, dpb, dps_]
    torch.cuda.synchronize()
    names = list(fx.param_shapes(A)["transition_model"].keys()) + ["dembeds", "dprev_belief", "dprev_state"]
    worst = 0.0
    for n, a, b in zip(names, res["0"], res["1"]):
        assert torch.isfinite(b).all(), n
        err = float((a - b).norm() / (a.norm() + 1e-20))
        worst = max(worst, err)
        assert err < 2e-5, (n, err)
    log(f"column-split reverse scan T={T} B={B} A={A} philox={philox}: worst l2 error vs the row scans {worst:.2e}")


def test_scan_engine_selection(monkeypatch):
    """REPO_SCAN_CS=auto: the column-split engine up to 64 rows at the reference's widths, the row scan beyond and for
    `observations=None`; both give the same numbers at the boundary; T = 1 (the acting step) works on either."""
    from repo_amd import ops

    A, D, S, E = 6, 200, 30, 1024
    p = cu(tparams("transition_model", A, requires_grad=False))
    rs = np.random.RandomState(3)
    dev = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()  # noqa: E731
    monkeypatch.setenv("REPO_SCAN_CS", "auto")
    for T, B, want_cs in ((1, 1, True), (3, 64, True), (3, 65, False)):
        b0, s0 = dev(rs.standard_normal((B, D)) * 0.3), dev(rs.standard_normal((B, S)))
        act, non = dev(rs.uniform(-1, 1, (T, B, A))), torch.ones(T, B).cuda()
        emb = dev(np.maximum(rs.standard_normal((T, B, E)), 0))
        sv = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, None, None, 0.1, noise=(3, 0))
        assert sv.cs == want_cs, (B, sv.cs)
        monkeypatch.setenv("REPO_SCAN_CS", "0" if want_cs else "1")
        other = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, None, None, 0.1, noise=(3, 0))
        monkeypatch.setenv("REPO_SCAN_CS", "auto")
        assert other.cs != want_cs
        for name in ("featx", "post_mean", "post_std", "prior_mean", "prior_std"):
            a, b = getattr(sv, name), getattr(other, name)
            assert float((a - b).abs().max() / (a.abs().max() + 1e-12)) < 2e-5, (B, name)
    prior = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, None, None, 0.1, noise=(3, 0), prior_only=True)
    assert not prior.cs


def test_scan_timeout_reaches_the_host():
    """The column-split scans wait on peer workgroups with bounded spins; a timeout is an ASYNCHRONOUS error and must
    reach the host as an exception, not as NaN losses (include/repo_hip.h: the `status` word of repo_rssm_observe_fwd /
    _bwd).  A spin limit of 0 (debug entry) makes the first unanswered poll give up: the forward scan raises bit 1,
    the reverse scan bit 2; ops.check_scan_status and the agent's per-update scalar read-back both raise RepoHipError;
    with the limit restored and the word cleared the same calls are clean again."""
    from repo_amd import ops
    from repo_amd._lib import RepoHipError, lib
    from tests.test_update_gpu import dev_batch, make_agent

    A, D, S, E, T, B = 6, 200, 30, 1024, 6, 16
    p = cu(tparams("transition_model", A, requires_grad=False))
    rs = np.random.RandomState(5)
    dev = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()  # noqa: E731
    b0, s0 = dev(rs.standard_normal((B, D)) * 0.3), dev(rs.standard_normal((B, S)))
    act, non = dev(rs.uniform(-1, 1, (T, B, A))), torch.ones(T, B).cuda()
    emb = dev(np.maximum(rs.standard_normal((T, B, E)), 0))
    word = ops.scan_status(b0.device)
    word.zero_()
    grads = [torch.zeros_like(t) for t in p]
    dfeat = dev(rs.standard_normal((T, B, D + S)))
    try:
        sv = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, None, None, 0.1, noise=(3, 0))
        assert sv.cs
        ops.check_scan_status(b0.device)                       # clean run: nothing raised
        assert lib().repo_debug_scan_spin_limit(0) == 1 << 22
        ops.rssm_observe_bwd(p, sv, grads, dfeat=dfeat)
        torch.cuda.synchronize()
        assert int(word.item()) == 2, int(word.item())         # reverse scan only
        with pytest.raises(RepoHipError, match="reverse"):
            ops.check_scan_status(b0.device)
        assert int(word.item()) == 0                           # reported once: the read clears the word
        bad = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, None, None, 0.1, noise=(3, 0))
        torch.cuda.synchronize()
        assert int(word.item()) == 1                           # forward scan
        assert not bool(torch.isfinite(bad.featx).all())       # and nothing plausible is left behind
        # the agent: the word travels inside the update's one scalar copy
        word.zero_()
        agent, _ = make_agent("repo", 6, 3, 4, 6)
        batch, _ = dev_batch(6, 3, 6, 11)
        agent.update(batch)
        with pytest.raises(RepoHipError, match="column-split observe scan"):
            agent.last_scalars
    finally:
        lib().repo_debug_scan_spin_limit(-1)
        torch.cuda.synchronize()
        word.zero_()
    good = ops.rssm_observe_fwd(p, b0, s0, act, non, emb, None, None, 0.1, noise=(3, 0))
    ops.rssm_observe_bwd(p, good, grads, dfeat=dfeat)
    ops.check_scan_status(b0.device)
    assert torch.equal(good.featx, sv.featx)
    agent, _ = make_agent("repo", 6, 3, 4, 6)
    agent.update(dev_batch(6, 3, 6, 11)[0])
    assert all(math.isfinite(v) for v in agent.last_scalars.values())


def _agent_state(agent):
    opts = [agent.model_optimizer, agent.actor_optimizer, agent.value_optimizer]
    ts = [t.clone() for o in opts for t in (o.flat, o.exp_avg, o.exp_avg_sq)]
    ts += [agent.log_beta.clone(), agent.beta_optimizer.exp_avg.clone(), agent.beta_optimizer.exp_avg_sq.clone()]
    counts = [o.step_count for o in opts] + [agent.beta_optimizer.step_count, agent._noise_counter]
    return ts, counts


def test_scan_timeout_skips_every_step_and_the_update_can_be_retried(monkeypatch):
    """A scan timeout must cost the caller a retry, not the model (include/repo_hip.h, repo_clip_adam's
    skip_if_nonzero): with the spin limit forced to 0 the update's gradients are NaN, every optimiser step of that
    update (model, dual, actor, critic) skips itself on the device, the agent raises RepoHipError -- and parameters,
    moments, log_beta, step counts and the noise offset are what they were.  The SAME update retried on the row-scan
    engine (REPO_SCAN_CS=0) then equals, bit for bit, a twin agent that never faulted."""
    from repo_amd import ops
    from repo_amd._lib import RepoHipError, lib
    from tests.test_update_gpu import dev_batch, make_agent

    L, B, H, A = 6, 3, 4, 6
    word = ops.scan_status(torch.device("cuda", 0))
    word.zero_()
    torch.manual_seed(3)
    agent, _ = make_agent("repo", L, B, H, A)
    torch.manual_seed(3)
    twin, _ = make_agent("repo", L, B, H, A)
    b0, b1 = dev_batch(L, B, A, 11)[0], dev_batch(L, B, A, 12)[0]
    for ag in (agent, twin):   # one clean update: non-zero moments, step counts 1
        ag.update(b0)
        assert all(math.isfinite(v) for v in ag.last_scalars.values())
    before, counts = _agent_state(agent)
    try:
        lib().repo_debug_scan_spin_limit(0)
        agent.update(b1)
        with pytest.raises(RepoHipError, match="optimiser steps were skipped.*rolled back"):
            agent.last_scalars
    finally:
        lib().repo_debug_scan_spin_limit(-1)
        torch.cuda.synchronize()
    assert int(word.item()) == 0                       # the fault was reported once; the device word is clear again
    after, counts_after = _agent_state(agent)
    assert counts_after == counts, (counts, counts_after)
    for i, (a, b) in enumerate(zip(before, after)):
        assert torch.equal(a, b), i                    # nothing was written, NaN or otherwise
    monkeypatch.setenv("REPO_SCAN_CS", "0")
    for ag in (agent, twin):
        ag.update(b1)
    sa, st = agent.last_scalars, twin.last_scalars
    assert sa == st and all(math.isfinite(v) for v in sa.values()), (sa, st)
    for i, (a, b) in enumerate(zip(_agent_state(agent)[0], _agent_state(twin)[0])):
        assert torch.equal(a, b), i
    assert _agent_state(agent)[1] == _agent_state(twin)[1]


def test_scan_timeout_with_a_later_update_in_flight_keeps_the_model_finite():
    """Pipelined updates (train_agent's join=False): the fault of update k is raised while update k+1 is in flight;
    update k's steps were skipped, update k+1 (clean) ran on the unchanged parameters; nothing is NaN afterwards."""
    from repo_amd import ops
    from repo_amd._lib import RepoHipError, lib
    from tests.test_update_gpu import dev_batch, make_agent

    L, B, H, A = 6, 3, 4, 6
    ops.scan_status(torch.device("cuda", 0)).zero_()
    agent, _ = make_agent("repo", L, B, H, A)
    b0, b1 = dev_batch(L, B, A, 11)[0], dev_batch(L, B, A, 12)[0]
    agent.update(b0, join=False)
    try:
        lib().repo_debug_scan_spin_limit(0)
        agent.update(b1, join=False)            # faults; flushes update 0's log (clean)
        torch.cuda.synchronize()
    finally:
        lib().repo_debug_scan_spin_limit(-1)
    with pytest.raises(RepoHipError, match="later update was already in flight"):
        agent.update(b0, join=False)            # enqueued clean; its log call flushes the faulted update's
    agent.synchronize()
    assert all(math.isfinite(v) for v in agent.last_scalars.values())
    for o in (agent.model_optimizer, agent.actor_optimizer, agent.value_optimizer):
        assert bool(torch.isfinite(o.flat).all()) and bool(torch.isfinite(o.exp_avg).all())
    assert bool(torch.isfinite(agent.log_beta))
