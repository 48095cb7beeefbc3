"""TIA (task-informed abstractions) on the MI355X kernels.

Same surface as the reference's `TIA(Dreamer)` (/root/reference/algorithms/repo/tia.py:17-238): a second
("distractor") RSSM filters the same embeddings; the frame is reconstructed as a learned per-pixel blend of a task
decoder and a distractor decoder (both 6-channel: recon | mask; mask_head = Conv2d(6,1,1)+Sigmoid, tia.py:69,123-127),
a third decoder reconstructs the frame from the distractor latents alone (tia.py:135-145), the distractor reward head
is trained adversarially (frozen head, tia_adv_coef * +log_prob, tia.py:150-158) and then fitted for
`tia_reward_train_steps` extra steps on detached latents (tia.py:184-196).  The actor-critic half is Dreamer's, on the
task latents.

Everything runs on the kernels of the Dreamer/RePo update: two observe scans (forward + reverse), three decoder
passes (layer T_DEC4 = the 6-channel output conv; the plain decoder keeps its fused output+NLL kernel), one fused
blend + mask-head + NLL pass (repo_tia_blend_nll), the reward heads, two KL reductions.

Optimiser state: the reference holds every model parameter in ONE torch Adam whose per-parameter step counts
differ (the distractor reward head is skipped by the model step -- its gradients are None under FreezeParameters --
and stepped `tia_reward_train_steps` times afterwards).  Here that is two FlatAdam groups; get_param_dict /
load_param_dict merge / split them in the reference's parameter order (tia.py:71-82).

zero_grad semantics: the description above is the reference run under torch >= 2.0, where
`optimizer.zero_grad()` sets gradients to None -- what this container runs and what the goldens tia_tiny / tia_coefs
pin.  The reference PINS torch==1.12.1 (requirements.txt:17), whose zero_grad() ZEROES gradients: there every
world-model parameter also takes a zero-gradient Adam step in each fitting iteration (tia.py:184-196: moments decay,
the parameter moves by lr * m_hat / (sqrt(v_hat) + eps), the step count advances), and from the second update on the
distractor reward head takes one in the main step.  `config.zero_grad_set_to_none = False` reproduces that behaviour
(two extra clip_adam launches on zeroed gradient buffers per update; golden tia_zeros.npz, generated from the reference
with zero_grad patched to set_to_none=False); the default True matches torch >= 2.0.
"""
import math
import os

import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from .dreamer import LOG_2PI, Dreamer
from .models.decoder import ObservationModel, RewardModel, TIAObservationModel
from .models.rssm import TransitionModel
from .models.utils import FlatAdam, adam_param_group


TIA_EXTRA_KEYS = ("distractor_transition_model", "distractor_obs_model", "distractor_only_obs_model",
                  "distractor_reward_model", "mask_head")


class TIA(Dreamer):
    _N_SCANS = 2   # task + distractor observe scans draw noise (Dreamer._noise_stride)

    # ------------------------------------------------------------------ construction
    def build_models(self, config, env):
        super().build_models(config, env)
        c, dev = config, self.device
        if env.observation_space.shape[-1] != 64:
            raise NotImplementedError("TIA is built for the reference's 64 x 64 frames")
        obs_size = env.observation_space.shape
        A = self.action_size
        # same construction order as the reference (tia.py:27-69) => same default init under a seed
        self.obs_model = TIAObservationModel(c.belief_size, c.state_size, c.embedding_size,
                                             c.cnn_activation_function).to(dev)
        self.distractor_transition_model = TransitionModel(
            c.belief_size, c.state_size, A, c.hidden_size, c.embedding_size, c.dense_activation_function).to(dev)
        self.distractor_obs_model = TIAObservationModel(c.belief_size, c.state_size, c.embedding_size,
                                                        c.cnn_activation_function).to(dev)
        self.distractor_only_obs_model = ObservationModel(False, obs_size, c.belief_size, c.state_size,
                                                          c.embedding_size, c.cnn_activation_function).to(dev)
        self.distractor_reward_model = RewardModel(c.belief_size, c.state_size, c.hidden_size,
                                                   c.dense_activation_function).to(dev)
        self.mask_head = nn.Sequential(nn.Conv2d(6, 1, 1), nn.Sigmoid()).to(dev)
        # reference order of model_params (tia.py:71-81); the distractor reward head is its own group (see module doc)
        self._ref_model_modules = (
            self.encoder, self.transition_model, self.reward_model, self.obs_model,
            self.distractor_transition_model, self.distractor_reward_model, self.distractor_obs_model,
            self.distractor_only_obs_model, self.mask_head,
        )
        self.model_params = [p for m in self._ref_model_modules for p in m.parameters()]
        main = [p for m in self._ref_model_modules if m is not self.distractor_reward_model for p in m.parameters()]
        self.model_optimizer = FlatAdam(main, lr=c.model_lr)
        self.d_reward_optimizer = FlatAdam(list(self.distractor_reward_model.parameters()), lr=c.model_lr)
        # data parallel: the model gradient leaves as two buckets, cut behind the encoder (the first module of the flat
        # buffer): [both filters, the three decoders, the task reward head, the mask head) is final when the two
        # backward chains join and goes out beside the encoder backward; the encoder's share follows in line
        self._model_cut = self.model_optimizer.offsets[len(list(self.encoder.parameters()))]
        self._dp_two_buckets = os.environ.get("REPO_DP_BUCKETS", "2") != "1"
        self._d_reward_has_grad = False   # zero_grad_set_to_none=False: the head's gradient exists (is not None) from
                                          # the first fitting step on

    def toggle_train(self, train=True):
        super().toggle_train(train)
        for m in (self.distractor_transition_model, self.distractor_obs_model, self.distractor_only_obs_model,
                  self.distractor_reward_model, self.mask_head):
            m.train(train)

    def _mask_pg(self):
        conv = self.mask_head[0]
        return conv.weight, conv.bias

    # ------------------------------------------------------------------ world model
    def _observe(self, model, actions, nonterms, embeds, T, B, noise):
        c, dev = self.c, self.device
        pr, _ = self._pg(model)
        b0, s0 = self._zero_state(B)
        return ops.rssm_observe_fwd(
            pr, b0, s0, actions[:-1].contiguous(), nonterms[:-1].reshape(T, B).contiguous(), embeds.view(T, B, -1),
            noise[0], noise[1], model.min_std_dev, noise=noise[2],
        )

    def train_dynamics(self, obs, actions, rewards, nonterms):
        """TIA world-model step (reference tia.py:84-209).  Returns the detached TASK (beliefs, posterior_states)."""
        c, dev = self.c, self.device
        obs, actions, rewards, nonterms = self._prep_batch(obs, actions, rewards, nonterms)
        L, B = obs.shape[:2]
        T = L - 1
        rows = T * B
        grow = self._global_rows(rows)
        D, S = c.belief_size, c.state_size
        inv = 1.0 / grow
        frames = obs[1:].reshape(rows, *obs.shape[2:])
        pe, ge = self._pg(self.encoder)
        embeds, enc_saved = Fn.encoder_fwd(pe, frames)
        # -- both filters over the same embeddings (noise order: task scan, then distractor scan: tia.py:88-121).
        #    A scan is a latency-bound kernel on ~25 CUs: the distractor's runs on the side stream beside the task's
        main = torch.cuda.current_stream(dev)
        side2 = self._side_stream
        sv_t_noise = (self._noise("obs_prior", (T, B, S)), self._noise("obs_post", (T, B, S)), self._draw(2 * T * B * S))
        sv_d_noise = (self._noise("d_obs_prior", (T, B, S)), self._noise("d_obs_post", (T, B, S)),
                      self._draw(2 * T * B * S))
        if side2 is not None:
            side2.wait_stream(main)
            with torch.cuda.stream(side2):
                sv_d = self._observe(self.distractor_transition_model, actions, nonterms, embeds, T, B, sv_d_noise)
            for name in sv_d.__slots__:   # allocated on the side stream, consumed on this one
                t = getattr(sv_d, name, None)
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
        sv_t = self._observe(self.transition_model, actions, nonterms, embeds, T, B, sv_t_noise)
        if side2 is not None:
            main.wait_stream(side2)
        else:
            sv_d = self._observe(self.distractor_transition_model, actions, nonterms, embeds, T, B, sv_d_noise)
        feat_t = sv_t.featx[1:].reshape(rows, D + S)
        feat_d = sv_d.featx[1:].reshape(rows, D + S)
        # -- masked joint reconstruction (tia.py:123-133)
        pd_t, gd_t = self._pg(self.obs_model)
        pd_d, gd_d = self._pg(self.distractor_obs_model)
        t_out, saved_t = Fn.decoder_fwd(pd_t, feat_t)
        d_out, saved_d = Fn.decoder_fwd(pd_d, feat_d)
        mw, mb = self._mask_pg()
        wb = torch.cat([mw.detach().reshape(6), mb.detach().reshape(1)])
        sums8, dt_out, dd_out, _ = ops.tia_blend_nll(t_out, d_out, wb, frames, inv, inplace=True)
        mw.grad.view(-1).copy_(sums8[1:7])
        mb.grad.view(-1).copy_(sums8[7:8])
        # -- distractor-only reconstruction (tia.py:135-145), weight tia_obs_coef
        pd_o, gd_o = self._pg(self.distractor_only_obs_model)
        nll_o, saved_o = Fn.decoder_fwd_nll(pd_o, feat_d, frames, float(c.tia_obs_coef) * inv)
        # -- reward heads (tia.py:147-158): the task head fits, the (frozen) distractor head is maximally wrong
        r_tgt = rewards[:-1].reshape(-1).contiguous()
        r_mask = nonterms[:-1].reshape(-1).contiguous()
        pw_t, gw_t = self._pg(self.reward_model)
        pw_d, gw_d = self._pg(self.distractor_reward_model)
        rt_pred, rt_hid = ops.mlp_fwd(pw_t, feat_t)
        rt_sums, drew_t = ops.scalar_nll(rt_pred.view(-1), r_tgt, r_mask, inv)
        rd_pred, rd_hid = ops.mlp_fwd(pw_d, feat_d)
        rd_sums, drew_d = ops.scalar_nll(rd_pred.view(-1), r_tgt, r_mask, -float(c.tia_adv_coef) * inv)
        # -- KL with free nats, per filter (tia.py:160-172); the raw means are logged
        kl_t, klg_t = ops.kl_balance(sv_t.prior_mean, sv_t.prior_std, sv_t.post_mean, sv_t.post_std, 1, 0.0, None,
                                     float(c.free_nats), inv)
        kl_d, klg_d = ops.kl_balance(sv_d.prior_mean, sv_d.prior_std, sv_d.post_mean, sv_d.post_std, 1, 0.0, None,
                                     float(c.free_nats), inv)
        klraw_t, _ = ops.kl_balance(sv_t.prior_mean, sv_t.prior_std, sv_t.post_mean, sv_t.post_std, 0, 0.0, None, 0.0,
                                    0.0, want_grads=False)
        klraw_d, _ = ops.kl_balance(sv_d.prior_mean, sv_d.prior_std, sv_d.post_mean, sv_d.post_std, 0, 0.0, None, 0.0,
                                    0.0, want_grads=False)
        # -- backward, task side: reward head -> decoder (attached) -> reverse scan
        pr_t, gr_t = self._pg(self.transition_model)
        pr_d, gr_d = self._pg(self.distractor_transition_model)
        side = self._wgrad_side(B)
        dfeat_t = torch.empty(rows, D + S, device=dev)
        dembeds = torch.empty(rows, c.embedding_size, device=dev)
        dembeds_d = torch.empty(rows, c.embedding_size, device=dev)
        dfeat_d = torch.empty(rows, D + S, device=dev)
        # The task chain (reward head, task decoder, the task filter's reverse scan) and the distractor chain (reward,
        # two decoders, reverse scan) meet only in the encoder backward: the task chain runs on the side stream with
        # its weight gradients in line, the distractor chain here (REPO_TIA_SPLIT=0: only the task's reverse scan goes
        # to the side stream)
        rs = side2 if side2 is not None else main
        split = os.environ.get("REPO_TIA_SPLIT", "1") == "1" and side2 is not None
        if not split:
            ops.mlp_bwd(pw_t, feat_t, rt_hid, drew_t.view(rows, 1), dparams=gw_t, dx=dfeat_t)
            Fn.decoder_bwd(pd_t, feat_t, (*saved_t, dt_out), gd_t, dfeat=dfeat_t, accumulate_dfeat=True, side=side)
        rs.wait_stream(main)
        with torch.cuda.stream(rs):
            if split:
                ops.mlp_bwd(pw_t, feat_t, rt_hid, drew_t.view(rows, 1), dparams=gw_t, dx=dfeat_t)
                Fn.decoder_bwd(pd_t, feat_t, (*saved_t, dt_out), gd_t, dfeat=dfeat_t, accumulate_dfeat=True, side=None)
            ops.rssm_observe_bwd(pr_t, sv_t, gr_t, dfeat=dfeat_t, dpm=klg_t[0], dps=klg_t[1], dqm=klg_t[2],
                                 dqs=klg_t[3], dembeds=dembeds, min_std=self.transition_model.min_std_dev)
        # -- distractor side: adversarial reward (input gradient only), both decoders, reverse scan
        ops.mlp_bwd(pw_d, feat_d, rd_hid, drew_d.view(rows, 1), dparams=None, dx=dfeat_d)
        Fn.decoder_bwd(pd_d, feat_d, (*saved_d, dd_out), gd_d, dfeat=dfeat_d, accumulate_dfeat=True, side=side)
        Fn.decoder_bwd(pd_o, feat_d, saved_o, gd_o, dfeat=dfeat_d, accumulate_dfeat=True, side=side)
        ops.rssm_observe_bwd(pr_d, sv_d, gr_d, dfeat=dfeat_d, dpm=klg_d[0], dps=klg_d[1], dqm=klg_d[2], dqs=klg_d[3],
                             dembeds=dembeds_d, min_std=self.distractor_transition_model.min_std_dev)
        main.wait_stream(rs)
        self._model_bucket_begin(tail=True)   # everything but the encoder's gradients is final: 9/10 of the buffer
        dembeds.add_(dembeds_d)
        Fn.encoder_bwd(pe, frames, enc_saved, dembeds, ge, side=side)
        self._model_step()
        opt = self.d_reward_optimizer
        zeros_mode = not bool(getattr(c, "zero_grad_set_to_none", True))
        if zeros_mode and self._d_reward_has_grad:
            # torch 1.12.1: the head's gradient, zeroed by zero_grad() and untouched under FreezeParameters, takes a
            # zero-gradient Adam step inside model_optimizer.step() (tia.py:182)
            opt.grad.zero_()
            opt.step()   # (no clip pass: it would overwrite the logged gradient norm, and scales zeros)
        # -- fit the distractor reward head on the detached latents (tia.py:184-196)
        last = rd_sums
        for it in range(int(c.tia_reward_train_steps)):
            if it > 0:
                rd_pred, rd_hid = ops.mlp_fwd(pw_d, feat_d)
            last, drew = ops.scalar_nll(rd_pred.view(-1), r_tgt, r_mask, inv)
            ops.mlp_bwd(pw_d, feat_d, rd_hid, drew.view(rows, 1), dparams=gw_d, dx=None)
            self._allreduce(opt.grad)
            opt.clip_and_step(c.grad_clip_norm)
            self._d_reward_has_grad = True
            if zeros_mode:
                # ... and every OTHER model parameter a zero-gradient step in this model_optimizer.step() (tia.py:196)
                if self._ev_ac_done is not None:   # the previous update's imagination may still read the parameters
                    torch.cuda.current_stream(self.device).wait_event(self._ev_ac_done)
                self.model_optimizer.grad.zero_()
                self.model_optimizer.step()
        # -- logging: the base layout carries the task-side sums, the rest rides as extras
        self._pending_model = (torch.cat([sums8[0:1], rt_sums, kl_t, self.model_optimizer.sqnorm]), None, grow)
        self._pending_extra = (torch.cat([nll_o, rd_sums, kl_d, klraw_t, klraw_d, last]),
                               opt.sqnorm.clone() if int(c.tia_reward_train_steps) > 0 else opt.sqnorm[:0])
        return sv_t.featx[1:, :, :D], sv_t.featx[1:, :, D:]

    def _extra_scalars(self, out, sums, norms, grow):
        """tia.py:198-208.  `out` holds the base keys computed from the task-side sums; sums = [nll_o, rsq_d, rmask_d,
        kl_d, klraw_t, klraw_d, last rsq_d, last rmask_d]."""
        c = self.c
        nll_o, rsq_d, rmask_d, kl_d, klraw_t, klraw_d, lrsq, lrmask = sums
        t_reward = out["train/reward_loss"]
        d_adv = -(rsq_d + 0.5 * LOG_2PI * rmask_d) / grow           # + log_prob * mask, mean
        out["train/d_obs_loss"] = nll_o / grow + 0.5 * LOG_2PI * self._npix
        out["train/t_reward_loss"] = t_reward
        out["train/reward_loss"] = t_reward + c.tia_adv_coef * d_adv
        # the reference logs the variable its fitting loop last assigned (tia.py:190-193,203)
        out["train/d_reward_loss"] = ((lrsq + 0.5 * LOG_2PI * lrmask) / grow if int(c.tia_reward_train_steps) > 0
                                      else d_adv)
        out["train/kl_loss"] = out["train/kl_loss"] + kl_d / grow
        out["train/t_kl_div"] = klraw_t / grow
        out["train/d_kl_div"] = klraw_d / grow
        out["train/model_loss"] = (out["train/obs_loss"] + c.tia_obs_coef * out["train/d_obs_loss"]
                                   + out["train/reward_loss"] + out["train/kl_loss"])
        if norms:
            self._d_reward_grad_norm = math.sqrt(max(norms[0], 0.0))

    # ------------------------------------------------------------------ evaluation / checkpoints
    def _reconstruct(self, belief, state):
        return self.obs_model(belief, state)[0]   # the task decoder's recon half (tia.py:227)

    def _merged_model_state(self):
        """The two Adam groups as ONE torch.optim.Adam state dict over model_params in the reference's order."""
        main, dr = self.model_optimizer, self.d_reward_optimizer
        state, i = {}, 0
        where = {id(p): (main, k) for k, p in enumerate(main.params)}
        where.update({id(p): (dr, k) for k, p in enumerate(dr.params)})
        for p in self.model_params:
            opt, k = where[id(p)]
            if opt.step_count > 0:
                o, n = opt.offsets[k], p.numel()
                state[i] = {"step": torch.tensor(float(opt.step_count)),
                            "exp_avg": opt.exp_avg[o : o + n].view(p.shape).clone(),
                            "exp_avg_sq": opt.exp_avg_sq[o : o + n].view(p.shape).clone()}
            i += 1
        return {"state": state, "param_groups": [adam_param_group(main.lr, main.betas, main.eps, len(self.model_params))]}

    def get_param_dict(self):
        # the reference's TIA saves Dreamer's keys only (tia.py inherits get_param_dict: the distractor modules are
        # not in its checkpoints); the model optimiser's state covers all of model_params
        params = super().get_param_dict()
        params["model_optimizer"] = self._merged_model_state()
        # EXTRA keys behind the reference's (its loader reads its own keys only, so the layout stays compatible): without
        # them a resumed TIA run restores the Adam moments of five modules whose weights restart from their initial values
        # (cloned like Dreamer.get_param_dict's entries: the parameters are views of FlatAdam's flat buffer, a held
        # dict must not follow later updates, and torch.save would serialise the whole flat storage per view)
        for k in TIA_EXTRA_KEYS:
            params[k] = {name: v.detach().clone() for name, v in getattr(self, k).state_dict().items()}
        return params

    def load_param_dict(self, params):
        merged = params["model_optimizer"]
        main, dr = self.model_optimizer, self.d_reward_optimizer
        idx = {id(p): i for i, p in enumerate(self.model_params)}
        for opt in (main, dr):
            sub = {"state": {}, "param_groups": merged["param_groups"]}
            for k, p in enumerate(opt.params):
                st = merged["state"].get(idx[id(p)])
                if st is not None:
                    sub["state"][k] = st
            opt.load_state_dict(sub)
        super().load_param_dict({**params, "model_optimizer": main.state_dict()})
        missing = [k for k in TIA_EXTRA_KEYS if k not in params]
        for k in TIA_EXTRA_KEYS:
            if k in params:
                self._load_module(getattr(self, k), params[k])
        if missing:   # a reference-written checkpoint: the reference resumes the same way
            import warnings
            warnings.warn(f"TIA checkpoint without {missing}: these modules keep their current weights while their "
                          "Adam state is restored (the reference's TIA saves Dreamer's keys only)")
