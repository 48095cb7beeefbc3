"""MultitaskDreamer: Dreamer with every module conditioned on the task one-hot, on the MI355X kernels.

Same surface as the reference's class (/root/reference/algorithms/repo/dreamer_mt.py:28-429): constructor
(config, env, eval_env, logger) with `env.num_tasks` / `env.task_one_hot` / `env.task`, the
MultitaskSequenceReplayBuffer, `train_dynamics(tasks, obs, actions, rewards, nonterms)`,
`train_actor_critic(tasks, beliefs, posterior_states)`, `update_latent_and_select_action(belief, state, action, obs,
task, explore)`, `train_agent`, `train`, `eval_agent` (one episode per task, round robin), checkpoints in the
reference's key layout -- plus `update(batch)` with batch = (tasks, obs, actions, rewards, dones).

How the conditioning maps onto the update's kernels (models/conditional.py has the modules):
 * encoder / decoder: FiLM -- bias-only conv epilogues + repo_film_fwd / repo_film_bwd (repo_amd/functional_mt.py);
 * observe scan: pseudo-actions [action | task] (A + C columns of the same fused scan);
 * reward / value / actor heads: [belief | state | task] rows (in_dim = 230 + C of the same MLP kernels);
 * imagination: repo_rssm_imagine_fwd(cond): the task rides in the K padding of the rollout's tiles.
`config.share_repr=True` is not built: the reference then constructs its RSSM with the WRONG positional argument
(dreamer_mt.py:57-64: `dense_activation_function` lands in DummyConditionalTransitionModel's `condition_size` slot,
so that run's RSSM silently uses ReLU), a configuration no script of the reference enables (train_repo.py:70).
"""
import os

import numpy as np
import torch

from ... import functional_mt as Fm
from ... import ops
from ...common.buffers import MultitaskSequenceReplayBuffer
from ...common.utils import postprocess, preprocess, to_np, to_torch
from .dreamer import LOG_2PI, Dreamer, _as_video
from .models.conditional import (ConditionalActorModel, ConditionalEncoder, ConditionalObservationModel,
                                 ConditionalRewardModel, ConditionalTransitionModel, ConditionalValueModel)


class MultitaskDreamer(Dreamer):
    _LATENT_ENTROPY_SHIFT = -(0.5 + 0.5 * LOG_2PI)   # dreamer_mt.py:258 logs imag_prior_std_devs.log().sum(-1).mean()

    def __init__(self, config, env, eval_env, logger):
        super().__init__(config, env, eval_env, logger)
        self.buffer = MultitaskSequenceReplayBuffer(
            config.replay_size, env.num_tasks, env.observation_space.shape, env.action_space.shape,
            obs_type=np.uint8 if config.pixel_obs else np.float32,
        )
        if getattr(config, "replay_on_device", True):
            self.buffer.enable_device_mirror(self.device)
        # data parallel: the model gradient is exchanged whole (no bucket overlap for the multitask agents)
        self._dp_two_buckets = False

    # ------------------------------------------------------------------ construction
    def _build_modules(self, config, env, obs_size, action_size):
        if getattr(config, "share_repr", False):
            raise NotImplementedError(
                "share_repr=True: the reference builds that RSSM with a misplaced positional argument "
                "(dreamer_mt.py:57-64, its transition model then runs ReLU); only the task-conditioned "
                "representation (share_repr=False, the default of train_repo.py:70) is built")
        dev = self.device
        C = self.num_tasks = int(env.num_tasks)
        if not 1 <= C <= 13:
            # checked HERE, before seed data is collected and buffers are allocated: repo_kl_balance_tasks and the
            # conditioned rollout take at most 13 / 16 condition columns (include/repo_hip.h); the reference's
            # multitask suites all have 3 tasks
            raise NotImplementedError(f"multitask agents: num_tasks = {C} is outside the kernels' 1..13")
        c = config
        # same construction order as the reference (dreamer_mt.py:66-127) => same default init under a seed
        self.encoder = ConditionalEncoder(False, obs_size, c.embedding_size, C, c.cnn_activation_function).to(dev)
        self.transition_model = ConditionalTransitionModel(
            c.belief_size, c.state_size, action_size, c.hidden_size, c.embedding_size, C, c.dense_activation_function
        ).to(dev)
        self.obs_model = ConditionalObservationModel(
            False, obs_size, c.belief_size, c.state_size, c.embedding_size, C, c.cnn_activation_function).to(dev)
        self.reward_model = ConditionalRewardModel(c.belief_size, c.state_size, c.hidden_size, C,
                                                   c.dense_activation_function).to(dev)
        # quirk kept: dense_activation_function lands in the `dist` slot (dreamer_mt.py:110-117)
        self.actor_model = ConditionalActorModel(c.belief_size, c.state_size, c.hidden_size, action_size, C,
                                                 c.dense_activation_function).to(dev)
        self.value_model = ConditionalValueModel(c.belief_size, c.state_size, c.hidden_size, C,
                                                 c.dense_activation_function).to(dev)

    # ------------------------------------------------------------------ world model
    def _mt_forward(self, tasks, obs, actions, rewards, nonterms):
        """Conditioned encoder, observe scan over pseudo-actions, conditioned decoder + NLL, conditioned reward head."""
        c, dev = self.c, self.device
        L, B = obs.shape[:2]
        T = L - 1
        rows = T * B
        grow = self._global_rows(rows)
        D, S = c.belief_size, c.state_size
        F_ = D + S
        st = {"T": T, "B": B, "rows": rows, "grow": grow}
        frames = obs[1:].reshape(rows, *obs.shape[2:])
        # frame t is encoded with tasks[t]; heads and decoder of step t see tasks[t] ("Match task timestep",
        # dreamer_mt.py:186-187); the scan's step t consumes [actions[t-1] | tasks[t-1]] (:176-183)
        cond = tasks[1:].reshape(rows, -1).contiguous()
        st["frames"], st["cond"] = frames, cond
        pe, _ = self._pg(self.encoder)
        embeds, st["enc_saved"] = Fm.cond_encoder_fwd(pe, frames, cond)
        pr, _ = self._pg(self.transition_model)
        pd, _ = self._pg(self.obs_model)
        # the decoder's composed first layers (functional.dec_head_compose): parameters only, made under the scan
        head = Fm.dec_head_compose(pd) if (Fm._fused() and Fm._dec_compose(rows)) else None
        pseudo = torch.cat((actions[:-1], tasks[:-1]), dim=2).contiguous()
        sv = ops.rssm_observe_fwd(
            pr, *self._zero_state(B), pseudo,
            nonterms[:-1].reshape(T, B).contiguous(), embeds.view(T, B, -1),
            self._noise("obs_prior", (T, B, S)), self._noise("obs_post", (T, B, S)), self.transition_model.min_std_dev,
            noise=self._draw(2 * T * B * S),
        )
        st["sv"] = sv
        featc = torch.empty(rows, F_ + cond.shape[1], device=dev)
        featc[:, :F_] = sv.featx[1:].reshape(rows, F_)
        featc[:, F_:] = cond
        st["featc"] = featc
        feat = sv.featx[1:].reshape(rows, F_)
        st["feat"] = feat
        st["nll_sum"], st["dec_saved"] = Fm.cond_decoder_fwd_nll(pd, feat, cond, frames, 1.0 / grow, head=head)
        pw, _ = self._pg(self.reward_model)
        r_pred, st["rew_hid"] = ops.mlp_fwd(pw, featc)
        st["rew_sums"], st["drew"] = ops.scalar_nll(
            r_pred.view(-1), rewards[:-1].reshape(-1).contiguous(), nonterms[:-1].reshape(-1).contiguous(), 1.0 / grow)
        return st

    def _mt_backward(self, st, kl_grads, decoder_attached):
        """reward head -> decoder (its input gradient only if attached) -> reverse scan -> encoder."""
        c, dev = self.c, self.device
        rows, sv, featc, cond = st["rows"], st["sv"], st["featc"], st["cond"]
        F_ = c.belief_size + c.state_size
        wside = self._wgrad_side(rows // sv.featx[1:].shape[0])   # sequences in the batch
        dfeatc = torch.empty_like(featc)
        pw, gw = self._pg(self.reward_model)
        ops.mlp_bwd(pw, featc, st["rew_hid"], st["drew"].view(rows, 1), dparams=gw, dx=dfeatc)
        pd, gd = self._pg(self.obs_model)
        Fm.cond_decoder_bwd(pd, st["feat"], cond, st["dec_saved"], gd, dfeat=dfeatc[:, :F_] if decoder_attached else None,
                            accumulate_dfeat=True, side=wside)
        pr, gr = self._pg(self.transition_model)
        dembeds = torch.empty(rows, c.embedding_size, device=dev)
        dpm, dps, dqm, dqs = kl_grads
        ops.rssm_observe_bwd(pr, sv, gr, dfeat=dfeatc[:, :F_].contiguous(), dpm=dpm, dps=dps, dqm=dqm, dqs=dqs,
                             dembeds=dembeds, min_std=self.transition_model.min_std_dev)
        pe, ge = self._pg(self.encoder)
        Fm.cond_encoder_bwd(pe, st["frames"], cond, st["enc_saved"], dembeds, ge, side=wside)

    def _prep_mt_batch(self, tasks, obs, actions, rewards, nonterms):
        obs, actions, rewards, nonterms = self._prep_batch(obs, actions, rewards, nonterms)
        return tasks.float().contiguous(), obs, actions, rewards, nonterms

    def train_dynamics(self, tasks, obs, actions, rewards, nonterms):
        """MultitaskDreamer world-model step (reference dreamer_mt.py:166-228).  tasks (L,B,C) one-hot, obs
        (L,B,3,64,64) float32 in [-1,1] or uint8; returns detached (beliefs, posterior_states)."""
        c = self.c
        tasks, obs, actions, rewards, nonterms = self._prep_mt_batch(tasks, obs, actions, rewards, nonterms)
        st = self._mt_forward(tasks, obs, actions, rewards, nonterms)
        sv, grow = st["sv"], st["grow"]
        kl_sum, kl_grads = ops.kl_balance(sv.prior_mean, sv.prior_std, sv.post_mean, sv.post_std, 1, 0.0, None,
                                          float(c.free_nats), 1.0 / grow)
        self._mt_backward(st, kl_grads, decoder_attached=True)
        self._model_step()
        self._pending_model = (torch.cat([st["nll_sum"], st["rew_sums"], kl_sum, self.model_optimizer.sqnorm]), None,
                               grow)
        D = c.belief_size
        return sv.featx[1:, :, :D], sv.featx[1:, :, D:]

    # ------------------------------------------------------------------ actor critic
    def train_actor_critic(self, tasks, beliefs, posterior_states):
        """Conditioned imagination + actor and critic steps (reference dreamer_mt.py:230-301); tasks (N, C)."""
        super().train_actor_critic(beliefs, posterior_states, cond=tasks.float().contiguous())

    # ------------------------------------------------------------------ update loop
    def update(self, batch, join=True):
        """One iteration of train_agent's loop body (dreamer_mt.py:303-322) on a device batch
        (tasks (L,B,C), obs, actions, rewards, dones); the two halves on two streams as in Dreamer.update."""
        tasks, obs, actions, rewards, dones = batch
        dev = self.device
        caller = torch.cuda.current_stream(dev)
        wm, ac = self._wm_stream, self._ac_stream
        wm.wait_stream(caller)
        with torch.cuda.stream(wm):
            nonterms = 1.0 - dones.float()
            tasks = tasks.float()
            beliefs, post = self.train_dynamics(tasks, obs, actions, rewards, nonterms)
            start_tasks = tasks[1:].flatten(0, 1).contiguous()
            ev_wm = torch.cuda.Event()
            ev_wm.record(wm)
        for t in (tasks, obs, actions, rewards, dones):
            t.record_stream(wm)
        with torch.cuda.stream(ac):
            ac.wait_event(ev_wm)
            beliefs.record_stream(ac)
            start_tasks.record_stream(ac)
            try:
                self.train_actor_critic(start_tasks, beliefs.flatten(0, 1), post.flatten(0, 1))
            finally:   # also when the previous update's fault leaves through this one's log call (Dreamer.update)
                self._ev_ac_done = torch.cuda.Event()
                self._ev_ac_done.record(ac)
        if join:
            self.synchronize()

    # ------------------------------------------------------------------ acting
    def collect_seed_data(self):
        env, ring = self.env, self.buffer
        obs, mid_episode = env.reset(), True
        while mid_episode or len(ring) < self.c.prefill:
            action = env.action_space.sample()
            following, reward, done, _ = env.step(action)
            ring.push(env.task_one_hot, obs, action, reward, done)
            mid_episode = not done
            obs = env.reset() if done else following

    @torch.no_grad()
    def _act_eager(self, belief, posterior_state, action, obs, task, explore):
        embed = self.encoder(obs, task)
        outs = self.transition_model.observe(belief, posterior_state, action.unsqueeze(0), task.unsqueeze(0),
                                             embed.unsqueeze(0))
        belief, posterior_state = outs[0].squeeze(0), outs[4].squeeze(0)
        action = self.actor_model.get_action(belief, posterior_state, task, det=not explore)
        if explore:   # drawn and clamped unconditionally, as the reference does (dreamer_mt.py:161-163): with
            # action_noise == 0 the draw still advances torch's generator, so seeded runs stay in step
            action = torch.clamp(action + torch.randn_like(action) * self.c.action_noise, -1, 1)
        return belief, posterior_state, action

    def update_latent_and_select_action(self, belief, posterior_state, action, obs, task, explore=False):
        """One filtering step + policy under the task (reference dreamer_mt.py:139-164), replayed from a HIP graph over
        static buffers like the single-task acting path (Dreamer.update_latent_and_select_action)."""
        self.synchronize()
        task = task.float()
        if not self._act_graph_enabled:
            with torch.no_grad():
                return self._act_eager(belief, posterior_state, action, obs, task, explore)
        key = (bool(explore), int(obs.shape[0]), obs.dtype)
        g = self._act_graphs.get(key)
        if g is None:
            g = self._capture_mt_act_graph(belief, posterior_state, action, obs, task, bool(explore))
            self._act_graphs[key] = g
        graph, sin, sout, _scratch = g
        for dst, src in zip(sin, (belief, posterior_state, action, obs, task)):
            dst.copy_(src)
        graph.replay()
        return tuple(t.clone() for t in sout)

    def _capture_mt_act_graph(self, belief, posterior_state, action, obs, task, explore):
        """As Dreamer._capture_act_graph: the graph owns its scratch."""
        sin = tuple(t.detach().to(self.device).clone().contiguous() for t in (belief, posterior_state, action, obs, task))
        graph, sout, scratch = ops.capture_graph(lambda: self._act_eager(*sin, explore), device=self.device)
        return graph, sin, sout, scratch

    def _step_env(self, env, latent, obs, explore):
        """Filter on `obs` under env's current task, act, step once -> (latent, transition pieces)."""
        frame = to_torch(preprocess(obs[None]), device=self.device)
        task = to_torch(np.asarray(env.task_one_hot, dtype=np.float32)[None], device=self.device)
        latent = self.update_latent_and_select_action(*latent, frame, task, explore)
        action = to_np(latent[2])[0]
        if not np.isfinite(action).all():   # fail loudly (rollout.EpisodeDriver.advance)
            raise FloatingPointError(f"the acting path returned a non-finite action {action!r} at environment step {self.step}")
        following, reward, done, info = env.step(action)
        return latent, action, following, reward, done, info

    def train(self):
        """Reference dreamer_mt.py:324-385: one transition per environment step (stored with the task one-hot);
        training, evaluation, checkpointing and log dumps fire on their periods of the step counter, in that order;
        per-task returns."""
        c = self.c
        if c.load_checkpoint:
            self.load_checkpoint()
        if len(self.buffer) == 0:
            self.collect_seed_data()
        periodic = ((c.train_every, self.train_agent), (c.eval_every, self.eval_agent),
                    (c.checkpoint_every, self.save_checkpoint), (c.log_every, self._dump_log))
        env = self.env
        latent = self.init_latent_and_action()
        obs = env.reset()
        task = env.task
        ep_return, ep_success = 0, 0
        while self.step < c.num_steps:
            one_hot = np.array(env.task_one_hot, copy=True)
            latent, action, following, reward, done, info = self._step_env(env, latent, obs, True)
            self.buffer.push(one_hot, obs, action, reward, done)
            obs = following
            ep_return += reward
            ep_success += info.get("success", 0)
            if done:
                self.logger.record(f"train/return_{task}", ep_return)
                self.logger.record(f"train/success_{task}", float(ep_success > 0))
                latent = self.init_latent_and_action()
                obs = env.reset()
                task = env.task
                ep_return, ep_success = 0, 0
            for period, job in periodic:
                if self.step % period == 0:
                    job()
            self.step += 1

    def _reconstruct(self, belief, state, task):
        return self.obs_model(belief, state, task)

    def eval_agent(self):
        """One deterministic-policy episode PER TASK, round robin from the evaluation environment's current task
        (reference dreamer_mt.py:387-429)."""
        self.toggle_train(False)
        env = self.eval_env
        for _ in range(env.num_tasks):
            latent = self.init_latent_and_action()
            task = env.sample_task(round_robin=True)
            obs = env.reset(task=task)
            done, total_reward, total_success, pairs = False, 0, 0, []
            while not done:
                seen = obs
                latent, _, obs, reward, done, info = self._step_env(env, latent, obs, False)
                if self.c.pixel_obs:
                    one_hot = to_torch(np.asarray(env.task_one_hot, dtype=np.float32)[None], device=self.device)
                    with torch.no_grad():
                        recon = self._reconstruct(latent[0], latent[1], one_hot)
                    pairs.append([seen, postprocess(to_np(recon))[0]])
                total_reward += reward
                total_success += info.get("success", 0)
            self.logger.record(f"test/return_{task}", total_reward)
            self.logger.record(f"test/success_{task}", float(total_success > 0))
            if pairs:
                clip = np.stack(pairs).transpose(1, 0, 2, 3, 4)
                self.logger.record(f"test/video_{task}", _as_video(clip, fps=30), exclude="stdout")
        self.toggle_train(True)

    def load_offline_data(self):
        raise NotImplementedError("offline datasets carry no task labels (single-task only in the reference too)")
