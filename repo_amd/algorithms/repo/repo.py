"""RePo: Dreamer with a posterior-predictability constraint instead of reconstruction.

Reference: /root/reference/algorithms/repo/repo.py:13-124.  The decoder is a *probe* on detached
latents (:46-48), the KL is split into a prior-training and a posterior-training half with
weight alpha = prior_train_steps/(1+prior_train_steps) (:64-81), the constraint
KL <= target_kl enters the loss through beta = exp(log_beta) (:82-83) and log_beta follows
dual ascent with its own Adam (:93-96).
"""
import numpy as np
import torch

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

    def train_dynamics(self, obs, actions, rewards, nonterms):
        c = self.c
        obs, actions, rewards, nonterms = self._prep_batch(obs, actions, rewards, nonterms)
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
                      betas=bo.betas, eps=bo.eps, out=self._dual_out)
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
