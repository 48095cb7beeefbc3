"""Test-time adaptation of a trained RePo agent by fine-tuning its ENCODER only.

Reference: `FinetunedRePo`, /root/reference/algorithms/repo/repo_adapt.py:26-127 (driven by experiments/adapt_repo.py:220):
the source agent's world model, reward head, actor and critic stay frozen; on target-domain replay the encoder is
trained to keep the reward predictable and the posterior close to the (frozen) prior -- reward NLL + beta * (KL -
target_kl), the full KL gradient through both arguments (repo_adapt.py:63-76) -- with the same dual ascent on log_beta as
RePo.  Here: encoder forward, the observe scan, the reward head's input gradient, the KL reduction, the reverse scan (for
its gradient into the embeddings; the frozen weights' gradients it also forms are discarded) and the encoder backward,
all on the update's kernels; one Adam over the encoder's slice of the model buffer (`FlatAdam.view`).
`CalibratedRePo` (repo_adapt.py:136-596: paired calibration data, a VDB discriminator) is not built.
"""
import os

import torch

from ... import functional as Fn
from ... import ops
from .dreamer import LOG_2PI
from .models.utils import FlatAdam
from .repo import RePo


class FinetunedRePo(RePo):
    def build_models(self, config, env):
        super().build_models(config, env)
        n_enc = len(list(self.encoder.parameters()))
        self.encoder_optimizer = FlatAdam.view(self.model_optimizer, n_enc, lr=config.model_lr)
        self._scratch_gr = None
        self._enc_log = None
        self._enc_host = torch.empty(8, dtype=torch.float32).pin_memory()

    def train_encoder(self, obs, actions, rewards, nonterms):
        """repo_adapt.py:31-94.  obs (L,B,3,64,64) float32 in [-1,1] or uint8."""
        c, dev = self.c, self.device
        obs, actions, rewards, nonterms = self._prep_batch(obs, actions, rewards, nonterms)
        L, B = obs.shape[:2]
        T = L - 1
        rows = T * B
        grow = self._global_rows(rows)
        D, S = c.belief_size, c.state_size
        frames = obs[1:].reshape(rows, *obs.shape[2:])
        pe, ge = self._pg(self.encoder)
        embeds, enc_saved = Fn.encoder_fwd(pe, frames)
        pr, _ = self._pg(self.transition_model)
        b0, s0 = self._zero_state(B)
        sv = ops.rssm_observe_fwd(
            pr, b0, s0, actions[:-1].contiguous(), nonterms[:-1].reshape(T, B).contiguous(), embeds.view(T, B, -1),
            self._noise("obs_prior", (T, B, S)), self._noise("obs_post", (T, B, S)), self.transition_model.min_std_dev,
            noise=self._draw(2 * T * B * S))
        feat = sv.featx[1:].reshape(rows, D + S)
        # reward NLL through the frozen head: only its input gradient
        pw, _ = self._pg(self.reward_model)
        r_pred, r_hid = ops.mlp_fwd(pw, feat)
        rew_sums, drew = ops.scalar_nll(r_pred.view(-1), rewards[:-1].reshape(-1).contiguous(),
                                        nonterms[:-1].reshape(-1).contiguous(), 1.0 / grow)
        dfeat = torch.empty(rows, D + S, device=dev)
        ops.mlp_bwd(pw, feat, r_hid, drew.view(rows, 1), dparams=None, dx=dfeat)
        # beta * KL(post || prior), gradient through BOTH arguments: the balanced form with alpha = 1/2, scale 2
        kl_sum, klg = ops.kl_balance(sv.prior_mean, sv.prior_std, sv.post_mean, sv.post_std, 0, 0.5, self.log_beta, 0.0,
                                     2.0 / grow)
        if self._scratch_gr is None:
            self._scratch_gr = [torch.empty_like(t) for t in pr]   # gradients of the frozen filter: discarded
        dembeds = torch.empty(rows, c.embedding_size, device=dev)
        ops.rssm_observe_bwd(pr, sv, self._scratch_gr, dfeat=dfeat, dpm=klg[0], dps=klg[1], dqm=klg[2], dqs=klg[3],
                             dembeds=dembeds, min_std=self.transition_model.min_std_dev)
        Fn.encoder_bwd(pe, frames, enc_saved, dembeds, ge, side=self._wgrad_side(B))
        opt = self.encoder_optimizer
        self._take_status()   # both scans are behind us on this stream: the step and the dual step skip on a fault
        self._allreduce(opt.grad)
        opt.clip_and_step(c.grad_clip_norm)
        kl_global = kl_sum
        if self.dp is not None:
            kl_global = kl_sum.clone()
            self._allreduce(kl_global)
        bo = self.beta_optimizer
        bo.step_count += 1
        ops.dual_step(self.log_beta, bo.exp_avg, bo.exp_avg_sq, kl_global, grow, c.target_kl, bo.lr, bo.step_count,
                      betas=bo.betas, eps=bo.eps, out=self._dual_out, skip=self._ustatus)
        # logging: one asynchronous copy, read when first needed
        self._flush_enc_log()
        buf = torch.cat([rew_sums, self._dual_out, opt.sqnorm, self._ustatus.view(torch.float32)])
        self._allreduce_scalars(buf, n_sum=2)
        self._enc_host[:8].copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._enc_log = (ev, grow, self._restore_point)

    def _flush_enc_log(self):
        if self._enc_log is None:
            return
        ev, grow, restore = self._enc_log
        self._enc_log = None
        ev.synchronize()
        self._raise_update_fault(int(self._enc_host[7:8].view(torch.int32).item()), restore)
        rsq, rmask, kl_div, kl_loss, beta_loss, beta, gsq = self._enc_host[:7].tolist()
        reward_loss = (rsq + 0.5 * LOG_2PI * rmask) / grow
        out = {"train/reward_loss": reward_loss, "train/kl_loss": kl_loss, "train/kl_div": kl_div,
               "train/encoder_loss": reward_loss + kl_loss, "train/beta": beta, "train/beta_loss": beta_loss}
        self._last_scalars = out
        self.last_grad_norms = {"encoder": max(gsq, 0.0) ** 0.5}
        for k, v in out.items():
            self.logger.record(k, v)

    @property
    def last_scalars(self):
        self._flush_enc_log()
        return self._last_scalars

    def train_agent(self):
        """repo_adapt.py:96-107: `train_steps` encoder steps on fresh target-domain batches."""
        c = self.c
        B, L = c.batch_size, c.chunk_size
        for _ in range(c.train_steps):
            obs, actions, rewards, dones = self.buffer.sample_to_device(B, L, self.device)
            self.train_encoder(obs, actions, rewards, 1.0 - dones.float())
        self._flush_enc_log()

    def train(self):
        self.load_source_models()
        super().train()

    def load_source_models(self):
        """repo_adapt.py:113-126: the source agent's six modules from `source_dir/models.pt` (reference key layout)."""
        path = os.path.join(self.c.source_dir, "models.pt")
        if os.path.exists(path):
            ckpt = torch.load(path, map_location=self.device, weights_only=False)
            print(f"Loaded checkpoint from {path}")
            for name in ("encoder", "transition_model", "obs_model", "reward_model", "actor_model", "value_model"):
                self._load_module(getattr(self, name), ckpt[name])
