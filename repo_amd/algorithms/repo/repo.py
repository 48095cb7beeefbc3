"""RePo: Dreamer with a posterior-predictability constraint instead of reconstruction.

Reference: /root/reference/algorithms/repo/repo.py:13-124.  The decoder is a *probe* on detached
latents (:46-48), the KL is split into a prior-training and a posterior-training half with
weight alpha = prior_train_steps/(1+prior_train_steps) (:64-81), the constraint
KL <= target_kl enters the loss through beta = exp(log_beta) (:82-83) and log_beta follows
dual ascent with its own Adam (:93-96).
"""
import os

import numpy as np
import torch

from ... import functional as Fn
from ... import ops
from .dreamer import Dreamer
from .models.utils import adam_param_group


class _ScalarAdam:
    """Adam state of the single dual variable; the step itself runs inside repo_dual_step."""

    def __init__(self, param, lr, betas=(0.9, 0.999), eps=1e-8):
        self.param = param
        self.lr, self.betas, self.eps = float(lr), tuple(betas), float(eps)
        self.step_count = 0
        self.exp_avg = torch.zeros(1, dtype=torch.float32, device=param.device)
        self.exp_avg_sq = torch.zeros(1, dtype=torch.float32, device=param.device)

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            state[0] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg.reshape(()).clone(),
                        "exp_avg_sq": self.exp_avg_sq.reshape(()).clone()}
        return {"state": state, "param_groups": [adam_param_group(self.lr, self.betas, self.eps, 1)]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        st = sd["state"].get(0)
        if st is not None:
            self.step_count = int(float(st["step"]))
            self.exp_avg.copy_(st["exp_avg"].reshape(1))
            self.exp_avg_sq.copy_(st["exp_avg_sq"].reshape(1))


class RePo(Dreamer):
    def build_models(self, config, env):
        super().build_models(config, env)
        # scalar dual variable, kept as a 0-dim tensor like the reference (repo.py:17-22)
        self.log_beta = torch.tensor(np.log(config.init_beta), dtype=torch.float, device=self.device)
        self.beta_optimizer = _ScalarAdam(self.log_beta, lr=self.c.beta_lr)
        self._dual_out = torch.zeros(4, dtype=torch.float32, device=self.device)

    def _train_dynamics_split(self, obs, actions, rewards, nonterms):
        """RePo's decoder is a probe on DETACHED latents (repo.py:46-48), so behind the forward scan there are two
        independent chains: [decoder forward + NLL -> decoder backward] and [reward / KL backward -> reverse scan ->
        encoder backward].  They run on two streams (the second on the side stream, its weight gradients in line)
        instead of decoder forward -> (reverse scan || decoder backward) -> encoder backward: 8.27 -> 8.03 ms per update
        (A/B on one box, round 3; sharing the weight-gradient stream between the chains: 8.17).  REPO_WM_SPLIT=0
        restores the serial order.  Data parallel: the decoder + reward-head bucket is exchanged in line on the
        decoder's stream as soon as that chain ends, beside the other chain's encoder backward."""
        c, dev = self.c, self.device
        L, B = obs.shape[:2]
        T = L - 1
        rows = T * B
        grow = self._global_rows(rows)
        D, S = c.belief_size, c.state_size
        frames = obs[1:].reshape(rows, *obs.shape[2:])
        pe, ge = self._pg(self.encoder)
        embeds, enc_saved = Fn.encoder_fwd(pe, frames)
        pr, gr = self._pg(self.transition_model)
        pd, gd = self._pg(self.obs_model)
        # the decoder's composed first layers depend on the parameters only: made here, under the latency-bound scan
        head = Fn.dec_head_compose(pd) if Fn._dec_compose(rows) else None
        sv = ops.rssm_observe_fwd(
            pr, *self._zero_state(B), actions[:-1].contiguous(),
            nonterms[:-1].reshape(T, B).contiguous(), embeds.view(T, B, -1), self._noise("obs_prior", (T, B, S)),
            self._noise("obs_post", (T, B, S)), self.transition_model.min_std_dev, noise=self._draw(2 * T * B * S))
        feat = sv.featx[1:].reshape(rows, D + S)
        pw, gw = self._pg(self.reward_model)
        r_pred, r_hid = ops.mlp_fwd(pw, feat)
        rew_sums, drew = ops.scalar_nll(r_pred.view(-1), rewards[:-1].reshape(-1).contiguous(),
                                        nonterms[:-1].reshape(-1).contiguous(), 1.0 / grow)
        alpha = c.prior_train_steps / (1 + c.prior_train_steps)
        kl_sum, klg = ops.kl_balance(sv.prior_mean, sv.prior_std, sv.post_mean, sv.post_std, 0, alpha, self.log_beta, 0.0,
                                     1.0 / grow)
        main, side = torch.cuda.current_stream(dev), self._side_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dfeat = torch.empty(rows, D + S, device=dev)
            ops.mlp_bwd(pw, feat, r_hid, drew.view(rows, 1), dparams=gw, dx=dfeat)
            ev_rew = torch.cuda.Event()
            ev_rew.record(side)   # the reward head's gradients (part of the decoder bucket) are final
            dembeds = torch.empty(rows, c.embedding_size, device=dev)
            ops.rssm_observe_bwd(pr, sv, gr, dfeat=dfeat, dpm=klg[0], dps=klg[1], dqm=klg[2], dqs=klg[3], dembeds=dembeds,
                                 min_std=self.transition_model.min_std_dev)
            Fn.encoder_bwd(pe, frames, enc_saved, dembeds, ge, side=None)
        nll_sum, dec_saved = Fn.decoder_fwd_nll(pd, feat, frames, 1.0 / grow, head=head)
        Fn.decoder_bwd(pd, feat, dec_saved, gd, side=self._wgrad_side(B))
        if self.dp is not None and self._dp_two_buckets:
            main.wait_event(ev_rew)
            g, cut = self.model_optimizer.grad, self._model_cut
            self._model_works.append(self.dp.all_reduce_begin(g[cut:], stream=main))
        main.wait_stream(side)
        self._model_step()
        kl_global = kl_sum
        if self.dp is not None:
            kl_global = kl_sum.clone()  # keep the local partial sum for the (summed) scalar log
            self._allreduce(kl_global)
        bo = self.beta_optimizer
        bo.step_count += 1
        ops.dual_step(self.log_beta, bo.exp_avg, bo.exp_avg_sq, kl_global, grow, c.target_kl, bo.lr, bo.step_count,
                      betas=bo.betas, eps=bo.eps, out=self._dual_out, skip=self._ustatus)
        self._pending_model = (torch.cat([nll_sum, rew_sums, kl_sum, self.model_optimizer.sqnorm]), self._dual_out.clone(),
                               grow)
        return sv.featx[1:, :, :D], sv.featx[1:, :, D:]

    def train_dynamics(self, obs, actions, rewards, nonterms):
        c = self.c
        obs, actions, rewards, nonterms = self._prep_batch(obs, actions, rewards, nonterms)
        if os.environ.get("REPO_WM_SPLIT", "1") == "1" and self._side_stream is not None:
            return self._train_dynamics_split(obs, actions, rewards, nonterms)
        st = self._world_model_forward(obs, actions, rewards, nonterms)
        sv, grow = st["sv"], st["grow"]
        alpha = c.prior_train_steps / (1 + c.prior_train_steps)
        if sv.prior_ready is not None:  # the prior head ran on the side stream, beside the decoder
            torch.cuda.current_stream(self.device).wait_stream(sv.prior_ready)
        # gradients use beta BEFORE the dual update (kl_loss = exp(log_beta).detach() * viol, repo.py:83)
        kl_sum, kl_grads = ops.kl_balance(sv.prior_mean, sv.prior_std, sv.post_mean, sv.post_std, 0, alpha,
                                          self.log_beta, 0.0, 1.0 / grow)
        self._world_model_backward(st, kl_grads, decoder_attached=False)
        self._model_step()
        # dual ascent on the global-batch KL
        kl_global = kl_sum
        if self.dp is not None:
            kl_global = kl_sum.clone()  # keep the local partial sum for the (summed) scalar log
            self._allreduce(kl_global)
        bo = self.beta_optimizer
        bo.step_count += 1
        ops.dual_step(self.log_beta, bo.exp_avg, bo.exp_avg_sq, kl_global, grow, c.target_kl, bo.lr, bo.step_count,
                      betas=bo.betas, eps=bo.eps, out=self._dual_out, skip=self._ustatus)
        self._pending_model = (torch.cat([st["nll_sum"], st["rew_sums"], kl_sum, self.model_optimizer.sqnorm]),
                               self._dual_out.clone(), grow)
        D = c.belief_size
        return sv.featx[1:, :, :D], sv.featx[1:, :, D:]

    def get_param_dict(self):
        params = super().get_param_dict()
        # a leaf that requires grad, like the reference's (repo.py:17-22): its load_param_dict REBINDS
        # self.log_beta to this tensor and differentiates through it on the next dual step
        params["log_beta"] = self.log_beta.detach().clone().requires_grad_(True)
        params["beta_optimizer"] = self.beta_optimizer.state_dict()
        return params

    def load_param_dict(self, params):
        super().load_param_dict(params)
        with torch.no_grad():
            self.log_beta.copy_(params["log_beta"].to(self.device))
        self.beta_optimizer.load_state_dict(params["beta_optimizer"])
