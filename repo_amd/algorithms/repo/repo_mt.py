"""MultitaskRePo: MultitaskDreamer with RePo's posterior-predictability constraint and ONE Lagrange multiplier PER TASK.

Reference: /root/reference/algorithms/repo/repo_mt.py:13-135.  The decoder is a probe on detached latents (:57-64), the KL
is RePo's balanced pair (:75-86), and with the task-conditioned representation (share_repr=False) `log_beta` is a
vector over tasks: a row's multiplier is exp(tasks_row @ log_beta) (:89-93), the dual loss
-(log_beta_row * viol_row).mean() (:98-99), and `train/beta_<i>` is logged per task (:111-112).
Kernels: repo_kl_balance_tasks (values, the four gradients and the per-task violation sums in one pass) and
repo_dual_step_tasks (Adam on the C-vector, on the device); everything else is MultitaskDreamer's.
"""
import numpy as np
import torch

from ... import ops
from .dreamer_mt import MultitaskDreamer
from .models.utils import adam_param_group


class _VectorAdam:
    """Adam state of the per-task dual variables; the step itself runs inside repo_dual_step_tasks."""

    def __init__(self, param, lr, betas=(0.9, 0.999), eps=1e-8):
        self.param = param
        self.lr, self.betas, self.eps = float(lr), tuple(betas), float(eps)
        self.step_count = 0
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            state[0] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg.clone(),
                        "exp_avg_sq": self.exp_avg_sq.clone()}
        return {"state": state, "param_groups": [adam_param_group(self.lr, self.betas, self.eps, 1)]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        st = sd["state"].get(0)
        if st is not None:
            self.step_count = int(float(st["step"]))
            self.exp_avg.copy_(st["exp_avg"].reshape(self.exp_avg.shape))
            self.exp_avg_sq.copy_(st["exp_avg_sq"].reshape(self.exp_avg_sq.shape))


class MultitaskRePo(MultitaskDreamer):
    def build_models(self, config, env):
        super().build_models(config, env)
        C = self.num_tasks
        # one dual variable per task (repo_mt.py:24-32)
        self.log_beta = torch.full((C,), float(np.log(config.init_beta)), dtype=torch.float, device=self.device)
        self.beta_optimizer = _VectorAdam(self.log_beta, lr=self.c.beta_lr)
        self._dual_out = torch.zeros(3 + C, dtype=torch.float32, device=self.device)

    def train_dynamics(self, tasks, obs, actions, rewards, nonterms):
        c = self.c
        tasks, obs, actions, rewards, nonterms = self._prep_mt_batch(tasks, obs, actions, rewards, nonterms)
        st = self._mt_forward(tasks, obs, actions, rewards, nonterms)
        sv, grow = st["sv"], st["grow"]
        alpha = c.prior_train_steps / (1 + c.prior_train_steps)
        # gradients use the multipliers BEFORE the dual update (kl_loss = (log_beta.exp().detach() * viol).mean(),
        # repo_mt.py:89-93); st["cond"] = tasks[1:] (:55-56)
        sums, kl_grads = ops.kl_balance_tasks(sv.prior_mean, sv.prior_std, sv.post_mean, sv.post_std, alpha,
                                              self.log_beta, st["cond"], c.target_kl, 1.0 / grow)
        self._mt_backward(st, kl_grads, decoder_attached=False)
        self._model_step()
        sums_global = sums
        if self.dp is not None:
            sums_global = sums.clone()   # keep the local partial KL sum for the (summed) scalar log
            self._allreduce(sums_global)
        bo = self.beta_optimizer
        bo.step_count += 1
        ops.dual_step_tasks(self.log_beta, bo.exp_avg, bo.exp_avg_sq, sums_global, grow, bo.lr, bo.betas, bo.eps,
                            bo.step_count, out=self._dual_out, skip=self._ustatus)
        self._pending_model = (torch.cat([st["nll_sum"], st["rew_sums"], sums[:1], self.model_optimizer.sqnorm]),
                               self._dual_out.clone(), grow)
        D = c.belief_size
        return sv.featx[1:, :, :D], sv.featx[1:, :, D:]

    def _dual_scalars(self, out, dual):
        out["train/kl_div"], out["train/kl_loss"], out["train/beta_loss"] = dual[0], dual[1], dual[2]
        for i, b in enumerate(dual[3:]):
            out[f"train/beta_{i}"] = b

    def get_param_dict(self):
        params = super().get_param_dict()
        params["log_beta"] = self.log_beta.detach().clone().requires_grad_(True)
        params["beta_optimizer"] = self.beta_optimizer.state_dict()
        return params

    def load_param_dict(self, params):
        super().load_param_dict(params)
        with torch.no_grad():
            self.log_beta.copy_(params["log_beta"].to(self.device).reshape(self.log_beta.shape))
        self.beta_optimizer.load_state_dict(params["beta_optimizer"])
