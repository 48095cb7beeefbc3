"""torch.autograd glue: the reference's module-level calls (encoder(obs), obs_model(b, s),
reward_model(b, s), transition_model.observe(...)) keep working with autograd, each as ONE
node whose forward/backward are the fused HIP passes.  The agents' update does not go
through these nodes (it schedules forward and backward by hand on flat gradient buffers);
they exist so that code written against the reference's modules -- e.g. the sibling
algorithms built on the same models -- can differentiate through them.
"""
import torch

from ... import functional as Fn
from ... import ops


def _det(ts):
    return [t.detach() for t in ts]


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obs, *params):
        p = _det(params)
        obs = obs.contiguous()
        embeds, saved = Fn.encoder_fwd(p, obs)
        ctx.obs, ctx.p, ctx.saved, ctx.dbg = obs, p, saved, ops.debug_snapshot()
        return embeds

    @staticmethod
    def backward(ctx, dembeds):
        g = [torch.empty_like(t) for t in ctx.p]
        with ops.debug_scope(ctx.dbg):
            Fn.encoder_bwd(ctx.p, ctx.obs, ctx.saved, dembeds.contiguous(), g)
        return (None, *g)


def encoder_apply(mod, observation):
    return _EncoderFn.apply(observation, *mod.plist())


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, *params):
        p = _det(params)
        recon, saved = Fn.decoder_fwd(p, feat)
        ctx.feat, ctx.p, ctx.saved, ctx.dbg = feat, p, saved, ops.debug_snapshot()
        return recon

    @staticmethod
    def backward(ctx, drecon):
        g = [torch.empty_like(t) for t in ctx.p]
        dfeat = torch.empty_like(ctx.feat) if ctx.needs_input_grad[0] else None
        with ops.debug_scope(ctx.dbg):
            Fn.decoder_bwd(ctx.p, ctx.feat, (*ctx.saved, drecon.contiguous()), g, dfeat=dfeat)
        return (dfeat, *g)


def decoder_apply(mod, belief, state):
    feat = torch.cat([belief, state], dim=1).contiguous()
    return _DecoderFn.apply(feat, *mod.plist())


class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, *params):
        p = _det(params)
        out, hid = ops.mlp_fwd(p, feat)
        ctx.feat, ctx.p, ctx.hid, ctx.dbg = feat, p, hid, ops.debug_snapshot()
        return out

    @staticmethod
    def backward(ctx, dout):
        need_w = any(ctx.needs_input_grad[1:])
        g = [torch.empty_like(t) for t in ctx.p] if need_w else None
        dfeat = torch.empty_like(ctx.feat) if ctx.needs_input_grad[0] else None
        with ops.debug_scope(ctx.dbg):
            ops.mlp_bwd(ctx.p, ctx.feat, ctx.hid, dout.contiguous(), dparams=g, dx=dfeat)
        return (dfeat, *(g if g is not None else [None] * len(ctx.p)))


def mlp_apply(mod, belief, state):
    feat = torch.cat([belief, state], dim=1).contiguous()
    return _MlpFn.apply(feat, *mod.plist())


class _ObserveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prev_belief, prev_state, actions, embeds, nonterms, eps_prior, eps_post, min_std, *params):
        p = _det(params)
        sv = ops.rssm_observe_fwd(p, prev_belief.contiguous(), prev_state.contiguous(), actions.contiguous(),
                                  nonterms.contiguous(), embeds.contiguous(), eps_prior.contiguous(),
                                  eps_post.contiguous(), min_std)
        ctx.p, ctx.sv, ctx.min_std, ctx.dbg = p, sv, min_std, ops.debug_snapshot()
        D = sv.D
        outs = (sv.featx[1:, :, :D], sv.prior_state, sv.prior_mean, sv.prior_std, sv.featx[1:, :, D:], sv.post_mean,
                sv.post_std)
        return tuple(o.contiguous() for o in outs)

    @staticmethod
    def backward(ctx, db, dps_, dpm, dpsd, dqs_, dqm, dqsd):
        sv = ctx.sv
        g = [torch.empty_like(t) for t in ctx.p]
        dfeat = torch.cat([db, dqs_], dim=2).contiguous()
        dembeds = torch.empty_like(sv.embeds) if ctx.needs_input_grad[3] else None
        dpb = torch.empty_like(sv.featx[0, :, : sv.D]).contiguous() if ctx.needs_input_grad[0] else None
        dpst = torch.empty_like(sv.featx[0, :, sv.D :]).contiguous() if ctx.needs_input_grad[1] else None
        with ops.debug_scope(ctx.dbg):   # the engine switches / spin limit of the thread that ran the forward
            ops.rssm_observe_bwd(ctx.p, sv, g, dfeat=dfeat, dprior_state=dps_.contiguous(), dpm=dpm.contiguous(),
                                 dps=dpsd.contiguous(), dqm=dqm.contiguous(), dqs=dqsd.contiguous(), dembeds=dembeds,
                                 dprev_belief=dpb, dprev_state=dpst, min_std=ctx.min_std)
        return (dpb, dpst, None, dembeds, None, None, None, None, *g)


def observe_apply(mod, prev_belief, prev_state, actions, observations, nonterminals, noise):
    T, B = actions.shape[:2]
    dev = actions.device
    S = mod.state_size
    if nonterminals is None:
        nonterminals = torch.ones(T, B, 1, device=dev)
    if noise is None:
        noise = (torch.randn(T, B, S, device=dev), torch.randn(T, B, S, device=dev))
    outs = _ObserveFn.apply(prev_belief, prev_state, actions, observations, nonterminals.reshape(T, B), noise[0],
                            noise[1], float(mod.min_std_dev), *mod.plist())
    return list(outs)


def tanh_normal_mode(mean, std, samples, eps=None):
    if eps is None:
        eps = torch.randn(samples, *mean.shape, device=mean.device)
    return ops.tanh_normal_mode(mean.contiguous(), std.contiguous(), eps.contiguous())
