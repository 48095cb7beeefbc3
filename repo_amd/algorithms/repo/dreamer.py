"""Dreamer agent on the MI355X kernels.

Same surface as the reference's `Dreamer` (/root/reference/algorithms/repo/dreamer.py:32):
constructor (config, env, eval_env, logger), build_models, train_dynamics, train_actor_critic,
train_agent, train, eval_agent, save/load_checkpoint, get/load_param_dict, plus `update(batch)`
= one train_dynamics + one train_actor_critic.

What differs is HOW an update runs.  The reference builds an autograd graph of ~2000 small ATen
ops per update; here forward and backward are scheduled by hand over the fused HIP passes
(repo_amd.functional / repo_amd.ops), gradients land directly in one flat buffer per optimiser
(FlatAdam), replay frames stay uint8 on the device (normalised inside conv1's loader), and
every logged scalar of an update is read back with ONE device->host copy instead of 11
.item() synchronisations (dreamer.py:292-295,376-379).

Data parallelism: each rank owns a slice of the batch rows; losses are scaled by the GLOBAL
row count so that a sum all-reduce of the flat gradient buffers (RCCL over xGMI) yields the
global-batch gradient before the global-norm clip, and every rank applies the identical
Adam step.
"""
import glob
import math
import os

import numpy as np
import torch

from ... import functional as Fn
from ... import ops
from ...common.buffers import SequenceReplayBuffer
from ...common.utils import get_device, postprocess, to_np
from .models.actor_critic import ActorModel, ValueModel
from .models.decoder import ObservationModel, RewardModel
from .models.encoder import Encoder
from .models.rssm import TransitionModel
from .models.utils import FlatAdam
from .rollout import EpisodeDriver

LOG_2PI = math.log(2.0 * math.pi)


def _d(ts):
    return [t.detach() for t in ts]


def _as_video(frames, fps):
    """The logger's video holder: the host application's `common.logger.Video` when this package runs
    inside the reference's scripts (its logger dispatches on that class, common/logger.py:26-35), else
    a plain (frames, fps) record."""
    try:
        from common.logger import Video  # the application's own logger module, if any
    except (ImportError, AttributeError):  # no such module / a `common` package without a Video class
        from ...common.utils import Video
    return Video(frames, fps)


class Dreamer:
    def __init__(self, config, env, eval_env, logger):
        self.c = config
        self.env = env
        self.eval_env = eval_env
        self.logger = logger
        self.device = get_device()
        if self.device.type != "cuda":
            raise RuntimeError(
                "repo_amd agents run on a HIP device only (call set_gpu_mode(True)); there is no CPU path"
            )
        self.step = 0
        self.dp = None  # set by repo_amd.parallel.attach() for multi-GPU runs
        self.noise_source = None  # tests inject pre-drawn noise here
        # the update's reparameterisation noise is drawn INSIDE the kernels (counter-based Philox: include/repo_hip.h,
        # "reparameterisation noise"); the stream is keyed by torch's seed at construction and the agent owns the
        # counter.  Seed torch BEFORE constructing the agent (torch.manual_seed afterwards does not move this
        # stream; `seed_noise()` does); two agents built under one seed share one stream unless re-seeded.  A
        # resumed run continues BEHIND the noise its optimiser steps already consumed (load_param_dict).
        # REPO_NOISE=torch restores torch.randn tensors (the round-1 behaviour).
        self._noise_seed = int(torch.initial_seed()) & ((1 << 64) - 1)
        self._noise_counter = 0
        self._noise_in_kernel = os.environ.get("REPO_NOISE", "philox") == "philox"
        self.build_models(config, env)
        self.buffer = SequenceReplayBuffer(
            config.replay_size,
            env.observation_space.shape,
            env.action_space.shape,
            obs_type=np.uint8 if config.pixel_obs else np.float32,
        )
        # batches are gathered on the GPU from a device mirror of the ring (common/buffers.py): the host-side
        # gather of 2500 scattered frames costs more than the update it feeds (14.4 vs 10.6 ms)
        if self.device.type == "cuda" and getattr(config, "replay_on_device", True):
            self.buffer.enable_device_mirror(self.device)
        self.free_nats = torch.full((1,), float(config.free_nats), device=self.device)
        # Intra-lane overlaps: w = weight-gradient kernels beside the data-gradient chain, s = reverse scan
        # beside the decoder backward, c = critic update beside the actor backward.  HIP multiplexes streams
        # onto 4 hardware queues (GPU_MAX_HW_QUEUES); measured over {4, 8} queues x {with, without RCCL's own
        # stream} (tools/ovl_sweep.sh) "ws" gives 93.5-95.6 updates/s in all four, while adding "c" swings
        # between 81.6 and 96.6 depending on which streams happen to share a queue.  "C" forks the critic
        # onto the world-model lane's weight-gradient stream (idle at that point of the pipeline) instead
        # of a stream of its own: +0.5-1 updates/s in all four settings, no new stream.
        ovl = os.environ.get("REPO_OVL", "wsC")
        self._side_stream = torch.cuda.Stream(device=self.device) if "s" in ovl else None
        self._wgrad_stream = torch.cuda.Stream(device=self.device) if "w" in ovl else None
        # ... but only for batches that fill the chip: at a strong-scaling shard's size (B = 6 .. 13 sequences) the
        # forks and joins of the weight-gradient stream cost more than the overlap gives (round 4, one box:
        # B = 6: 2.52 -> 2.40 ms, B = 13: 3.15 -> 2.91 without it; B = 25 and 50: equal within 0.5 %)
        self._wgrad_min_batch = 20
        # "c" (experiment) borrows the world-model lane's weight-gradient stream instead of creating a stream
        self._ac_side_stream = (self._wgrad_stream if "C" in ovl else
                                torch.cuda.Stream(device=self.device)) if ("c" in ovl or "C" in ovl) else None
        # update(): the world-model lane and the actor-critic lane run on their own streams so
        # that WM(k+1) overlaps AC(k) (see update()); events order the only true dependencies
        # (measured: giving the world-model lane the high-priority hardware queues is SLOWER, 13.5 vs
        # 11.7 ms/update -- all streams stay at the default priority)
        self._wm_stream = torch.cuda.Stream(device=self.device)
        self._ac_stream = torch.cuda.Stream(device=self.device)
        self._ev_ac_done = None      # AC(k) finished reading the world-model parameters
        self._log_pending = None     # (event, pinned host buffer, meta) of the last enqueued update
        self._pending_extra = None   # (loss sums, gradient norms) a sibling algorithm adds to the update's log
        self._log_host = torch.empty(32, dtype=torch.float32).pin_memory()
        self._act_graphs = {}
        self._act_graph_enabled = os.environ.get("REPO_ACT_GRAPH", "1") == "1"
        self._last_scalars = {}
        self.last_grad_norms = {}
        # fail-safe updates (include/repo_hip.h, repo_clip_adam `skip_if_nonzero`): every update owns one word of this
        # ring -- its copy of the scans' asynchronous status, taken (and, data parallel, MAX-reduced) right before its
        # first optimiser step; all of its steps skip themselves on a non-zero word and `_flush_log` raises.  A ring
        # because the actor-critic lane of update k still reads its word while update k+1 takes the next one.
        self._status_ring = torch.zeros(8, dtype=torch.int32, device=self.device)
        self._ustatus = self._status_ring[7:8]
        self._update_seq = 0
        self._restore_point = None

    # ------------------------------------------------------------------ construction
    def build_models(self, config, env):
        if not config.pixel_obs:
            raise NotImplementedError("only pixel observations are on the MI355X hot path")
        if getattr(config, "disag_model", False) or getattr(config, "inv_dynamics", False):
            raise NotImplementedError("disagreement / inverse-dynamics auxiliaries are out of scope (SURVEY 2.1 #10)")
        obs_size = env.observation_space.shape
        action_size = int(np.prod(env.action_space.shape))
        self.action_size = action_size
        self._npix = int(np.prod(obs_size))  # 3*64*64 (the reference's frames) or 3*128*128 (build-defined)
        self._build_modules(config, env, obs_size, action_size)
        self._build_optimizers(config)

    def _build_modules(self, config, env, obs_size, action_size):
        """The six modules (the multitask agents build the task-conditioned ones instead: dreamer_mt.py)."""
        dev = self.device
        # same construction order as the reference (dreamer.py:57-114) => same default init under a seed
        self.encoder = Encoder(False, obs_size, config.embedding_size, config.cnn_activation_function).to(dev)
        self.transition_model = TransitionModel(
            config.belief_size, config.state_size, action_size, config.hidden_size, config.embedding_size,
            config.dense_activation_function,
        ).to(dev)
        self.obs_model = ObservationModel(
            False, obs_size, config.belief_size, config.state_size, config.embedding_size,
            config.cnn_activation_function,
        ).to(dev)
        self.reward_model = RewardModel(
            config.belief_size, config.state_size, config.hidden_size, config.dense_activation_function
        ).to(dev)
        # quirk kept: dense_activation_function lands in ActorModel's `dist` slot (dreamer.py:99-105)
        self.actor_model = ActorModel(
            config.belief_size, config.state_size, config.hidden_size, action_size, config.dense_activation_function
        ).to(dev)
        self.value_model = ValueModel(
            config.belief_size, config.state_size, config.hidden_size, config.dense_activation_function
        ).to(dev)

    def _build_optimizers(self, config):
        dev = self.device
        self.model_params = (
            list(self.encoder.parameters())
            + list(self.transition_model.parameters())
            + list(self.obs_model.parameters())
            + list(self.reward_model.parameters())
        )
        self.model_optimizer = FlatAdam(self.model_params, lr=config.model_lr)
        # the actor's and the critic's flat gradients are the two halves of ONE buffer: a data-parallel job
        # exchanges them as a single 1.18 MB bucket
        na = FlatAdam.padded_numel(list(self.actor_model.parameters()))
        nv = FlatAdam.padded_numel(list(self.value_model.parameters()))
        self._ac_grad = torch.zeros(na + nv, dtype=torch.float32, device=dev)
        self.actor_optimizer = FlatAdam(self.actor_model.parameters(), lr=config.actor_lr, grad=self._ac_grad[:na])
        self.value_optimizer = FlatAdam(self.value_model.parameters(), lr=config.value_lr, grad=self._ac_grad[na:])
        # model gradient buckets of a data-parallel job, in the order the backward finishes them: [decoder +
        # reward head) is final when the decoder backward joins, [encoder + RSSM) after the encoder backward
        n_head = len(list(self.encoder.parameters())) + len(list(self.transition_model.parameters()))
        self._model_cut = self.model_optimizer.offsets[n_head]
        self._model_works = []
        # REPO_DP_BUCKETS=1: the whole 20.7 MB model gradient as ONE all-reduce after the backward (the round-2
        # exchange; kept for A/B runs on a multi-GPU node and for the bucketed-equals-single test)
        self._dp_two_buckets = os.environ.get("REPO_DP_BUCKETS", "2") != "1"

    def _wgrad_side(self, batch):
        """The weight-gradient side stream for a batch of `batch` sequences, or None (in line)."""
        return self._wgrad_stream if batch >= self._wgrad_min_batch else None

    def _pg(self, module):
        """(params, grads) of a module as detached tensors / flat-gradient views, state_dict order."""
        ps = module.plist()
        return _d(ps), [p.grad for p in ps]

    def toggle_train(self, train=True):
        for m in (self.encoder, self.transition_model, self.obs_model, self.reward_model, self.actor_model,
                  self.value_model):
            m.train(train)

    # ------------------------------------------------------------------ helpers
    def _noise(self, key, shape):
        """An explicit noise tensor (injected by a test, or torch.randn), or None = "draw it in the kernel"."""
        if self.noise_source is not None:
            t = self.noise_source[key]
            assert tuple(t.shape) == tuple(shape), (key, t.shape, shape)
            return t
        if self._noise_in_kernel:
            return None
        return torch.randn(*shape, device=self.device)

    _N_SCANS = 1   # observe scans per update (TIA: 2)
    # added per state dimension to the logged latent entropy: the multitask agents log sum(log std) WITHOUT the Normal
    # entropy's constant 0.5 + 0.5 ln(2 pi) (reference dreamer_mt.py:258 against dreamer.py:327-328)
    _LATENT_ENTROPY_SHIFT = 0.0

    def _noise_stride(self):
        """Upper bound (a power of two) of the normals ONE update draws at this configuration:
        scans * 2*T*B*S + Hm*N*(A+S) + samples*Hm*N*A (21.3 M at B=50, L=50, H=15, A=6 -> 2**25; it grows with B, A and
        the horizon, so it is computed, not a constant).  A resumed run skips this many per optimiser step already
        taken and therefore never replays the noise of the run that wrote the checkpoint."""
        c = self.c
        T, B, Hm = c.chunk_size - 1, c.batch_size, c.horizon - 1
        S, A, N = self.transition_model.state_size, self.action_size, T * B
        per_update = self._N_SCANS * 2 * T * B * S + Hm * N * (A + S) + self.actor_model._samples * Hm * N * A
        return 1 << max(int(per_update) - 1, 1).bit_length()

    def seed_noise(self, seed):
        """Re-key the in-kernel Philox stream of the update's reparameterisation noise and rewind its counter."""
        self._noise_seed = int(seed) & ((1 << 64) - 1)
        self._noise_counter = 0

    def _draw(self, n):
        """Reserve n normals of the agent's Philox stream: (seed, offset) for one op."""
        off = self._noise_counter
        self._noise_counter += int(n)
        return self._noise_seed, off

    def _zero_state(self, B):
        """The initial (belief, state) of a scan -- zeros (repo.py:26-27) -- as constants of the agent: the scans only
        read them, so the two fill launches per update are made once per batch size."""
        z = self.__dict__.setdefault("_zero_states", {})
        if B not in z:
            z[B] = (torch.zeros(B, self.c.belief_size, device=self.device), torch.zeros(B, self.c.state_size, device=self.device))
        return z[B]

    def _global_rows(self, local_rows):
        """Row count of the global batch (sum over data-parallel ranks)."""
        if self.dp is None:
            return local_rows
        return self.dp.global_count(local_rows)

    def _allreduce(self, t):
        if self.dp is not None:
            self.dp.all_reduce(t)

    # ------------------------------------------------------------------ world model
    def _world_model_forward(self, obs, actions, rewards, nonterms):
        """Shared by Dreamer and RePo: encoder, observe scan, decoder+NLL, reward head.
        Returns a dict of everything the backward needs."""
        c = self.c
        L, B = obs.shape[:2]
        T = L - 1
        rows = T * B
        grow = self._global_rows(rows)
        dev = self.device
        D, S = c.belief_size, c.state_size
        st = {"T": T, "B": B, "rows": rows, "grow": grow}
        # frame 0's embedding is never used (embeds[1:], repo.py:41): encode frames 1..L-1 only
        frames = obs[1:].reshape(rows, *obs.shape[2:])
        st["frames"] = frames
        pe, _ = self._pg(self.encoder)
        embeds, st["enc_saved"] = Fn.encoder_fwd(pe, frames)
        pr, _ = self._pg(self.transition_model)
        b0, s0 = self._zero_state(B)
        pd, _ = self._pg(self.obs_model)
        # the decoder's composed first layers (functional.dec_head_compose) depend on the parameters only: made here, under
        # the latency-bound scan, off the decoder's chain
        head = Fn.dec_head_compose(pd) if Fn._dec_compose(rows) else None
        sv = ops.rssm_observe_fwd(
            pr, b0, s0, actions[:-1].contiguous(), nonterms[:-1].reshape(T, B).contiguous(), embeds.view(T, B, -1),
            self._noise("obs_prior", (T, B, S)), self._noise("obs_post", (T, B, S)), self.transition_model.min_std_dev,
            noise=self._draw(2 * T * B * S),
            # the prior head is off the recurrence: evaluated for all steps on the side stream, beside the decoder
            prior_stream=self._side_stream if os.environ.get("REPO_PRIOR_HOIST", "1") == "1" else None,
        )
        st["sv"] = sv
        feat = sv.featx[1:].reshape(rows, D + S)
        st["feat"] = feat
        # decoder + pixel NLL (mean over (T,B) of the per-frame sums)
        st["nll_sum"], st["dec_saved"] = Fn.decoder_fwd_nll(pd, feat, frames, 1.0 / grow, head=head)
        # reward head; predicted from the next state, masked by nonterminal (repo.py:58-61)
        pw, _ = self._pg(self.reward_model)
        r_pred, st["rew_hid"] = ops.mlp_fwd(pw, feat)
        st["rew_sums"], st["drew"] = ops.scalar_nll(
            r_pred.view(-1), rewards[:-1].reshape(-1).contiguous(), nonterms[:-1].reshape(-1).contiguous(), 1.0 / grow
        )
        return st

    def _world_model_backward(self, st, kl_grads, decoder_attached):
        """reward head -> (decoder) -> reverse scan -> encoder; gradients into the flat model buffer."""
        rows = st["rows"]
        sv = st["sv"]
        feat = st["feat"]
        dev = self.device
        wside = self._wgrad_side(st["frames"].shape[0] // sv.featx[1:].shape[0])   # sequences in the batch
        dfeat = torch.empty(rows, feat.shape[1], device=dev)
        pw, gw = self._pg(self.reward_model)
        ops.mlp_bwd(pw, feat, st["rew_hid"], st["drew"].view(rows, 1), dparams=gw, dx=dfeat)
        pd, gd = self._pg(self.obs_model)
        pr, gr = self._pg(self.transition_model)
        pe, ge = self._pg(self.encoder)
        dembeds = torch.empty(rows, self.c.embedding_size, device=dev)
        dpm, dps, dqm, dqs = kl_grads
        if decoder_attached:
            # Dreamer: the decoder's INPUT gradient feeds the reverse scan, so the data-gradient chain runs
            # first; the decoder's weight gradients feed nothing downstream and are issued afterwards,
            # beside the reverse scan (a latency-bound kernel on ~25 CUs) which hides them.
            wgrads = []
            Fn.decoder_bwd(pd, feat, st["dec_saved"], gd, dfeat=dfeat, accumulate_dfeat=True, deferred=wgrads)
            main = torch.cuda.current_stream(dev)
            side = self._side_stream or main
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.rssm_observe_bwd(pr, sv, gr, dfeat=dfeat, dpm=dpm, dps=dps, dqm=dqm, dqs=dqs, dembeds=dembeds,
                                     min_std=self.transition_model.min_std_dev)
            for fn in wgrads:
                fn()
            main.wait_stream(side)
            self._model_bucket_begin(tail=True)   # decoder + reward-head gradients are final
            Fn.encoder_bwd(pe, st["frames"], st["enc_saved"], dembeds, ge, side=wside)
            return
        # RePo: the decoder is a probe on detached latents (repo.py:46-48), so its backward is
        # independent of the RSSM/encoder backward.  The reverse scan is a latency-bound chain
        # that occupies ~50 CUs; running it on a side stream lets the compute-bound decoder
        # backward fill the other ~200 CUs instead of waiting for it.
        main = torch.cuda.current_stream(dev)
        side = self._side_stream or main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.rssm_observe_bwd(pr, sv, gr, dfeat=dfeat, dpm=dpm, dps=dps, dqm=dqm, dqs=dqs, dembeds=dembeds,
                                 min_std=self.transition_model.min_std_dev)
        Fn.decoder_bwd(pd, feat, st["dec_saved"], gd, side=wside)
        main.wait_stream(side)
        self._model_bucket_begin(tail=True)   # decoder + reward-head gradients are final (15.7 MB of 20.7)
        Fn.encoder_bwd(pe, st["frames"], st["enc_saved"], dembeds, ge, side=wside)

    def _model_bucket_begin(self, tail):
        """Data parallel: start the SUM all-reduce of one of the two model-gradient buckets on RCCL's stream,
        ordered behind everything enqueued so far on the current stream (every writer of the bucket has been
        joined into it).  The decoder + reward bucket goes out while the encoder backward still computes."""
        if self.dp is None or not self._dp_two_buckets:
            return
        g, cut = self.model_optimizer.grad, self._model_cut
        if not tail:   # the last bucket: nothing left to overlap it with, exchanged in line
            self._allreduce(g[:cut])
            return
        # the reverse scan's side stream is idle from here to the end of the backward: the bucket rides on it
        self._model_works.append(self.dp.all_reduce_begin(g[cut:], stream=self._side_stream))

    def _steppers(self):
        """Everything of this agent that counts optimiser steps (FlatAdam groups, the dual variables' Adam states)."""
        return [v for v in vars(self).values() if hasattr(v, "step_count") and hasattr(v, "exp_avg")]

    def _take_status(self):
        """Called on the stream that has joined every scan of this update, before its first optimiser step: the
        update's own status word (ops.take_scan_status) becomes the `skip` word of every step of this update."""
        st = self._status_ring[self._update_seq % 8 : self._update_seq % 8 + 1]
        ops.take_status_into(ops.scan_status(self.device), st)
        if self.dp is not None:
            self.dp.all_reduce_status(st)
        self._ustatus = st
        for opt in self._steppers():
            opt.skip = st
        return st

    def _model_step(self):
        self._take_status()
        # the optimiser step WRITES the world-model parameters the previous update's imagination
        # may still be reading on the actor-critic stream
        if self._ev_ac_done is not None:
            torch.cuda.current_stream(self.device).wait_event(self._ev_ac_done)
        if self.dp is not None and self._dp_two_buckets:
            self._model_bucket_begin(tail=False)  # encoder + RSSM
            works, self._model_works = self._model_works, []
            assert len(works) == 1, "model gradient buckets: the decoder bucket was not started by the backward"
            self.dp.all_reduce_end(works)
        elif self.dp is not None:
            self._allreduce(self.model_optimizer.grad)
        self.model_optimizer.clip_and_step(self.c.grad_clip_norm)

    def _prep_batch(self, obs, actions, rewards, nonterms):
        # the first thing every train_dynamics variant calls: where a faulted update is rolled back to (_flush_log)
        self._update_seq += 1
        self._restore_point = (self._update_seq, self._noise_counter, [(o, o.step_count) for o in self._steppers()])
        obs = obs.contiguous()
        assert obs.dtype in (torch.uint8, torch.float32), obs.dtype
        return obs, actions.float().contiguous(), rewards.float().contiguous(), nonterms.float().contiguous()

    def train_dynamics(self, obs, actions, rewards, nonterms):
        """Dreamer world-model step (reference dreamer.py:241-302).  obs (L,B,3,64,64) float32 in
        [-1,1] (reference convention) or uint8; returns detached (beliefs, posterior_states)."""
        c = self.c
        obs, actions, rewards, nonterms = self._prep_batch(obs, actions, rewards, nonterms)
        st = self._world_model_forward(obs, actions, rewards, nonterms)
        sv, grow = st["sv"], st["grow"]
        if sv.prior_ready is not None:
            torch.cuda.current_stream(self.device).wait_stream(sv.prior_ready)
        kl_sum, kl_grads = ops.kl_balance(sv.prior_mean, sv.prior_std, sv.post_mean, sv.post_std, 1, 0.0, None,
                                          float(c.free_nats), 1.0 / grow)
        self._world_model_backward(st, kl_grads, decoder_attached=True)
        self._model_step()
        self._pending_model = (torch.cat([st["nll_sum"], st["rew_sums"], kl_sum, self.model_optimizer.sqnorm]), None,
                               grow)
        D = c.belief_size
        return sv.featx[1:, :, :D], sv.featx[1:, :, D:]

    # ------------------------------------------------------------------ actor critic
    def train_actor_critic(self, beliefs, posterior_states, cond=None):
        """Imagination + actor and critic steps (reference dreamer.py:304-381).
        beliefs (N, D), posterior_states (N, S): detached start states.
        cond (N, C): the multitask agents' task one-hot (dreamer_mt.py:230-301 of the reference) -- every head then
        reads [belief | state | cond] rows and the rollout runs conditioned (repo_rssm_imagine_fwd, cond)."""
        c = self.c
        dev = self.device
        N = beliefs.shape[0]
        Hm = c.horizon - 1
        D, S, A = c.belief_size, c.state_size, self.action_size
        F_ = D + S
        gN = self._global_rows(N)
        pr, _ = self._pg(self.transition_model)
        pa, ga = self._pg(self.actor_model)
        pv, gv = self._pg(self.value_model)
        pw, _ = self._pg(self.reward_model)
        am = self.actor_model
        a_consts = (am._min_std, am._init_std, float(am._mean_scale))
        # -- imagine (world model frozen, actor inputs detached)
        sv = ops.rssm_imagine_fwd(
            pr, pa, beliefs.contiguous(), posterior_states.contiguous(), self._noise("img_act", (Hm, N, A)),
            self._noise("img_prior", (Hm, N, S)), self.transition_model.min_std_dev, *a_consts, spare_slot=True,
            noise=self._draw(Hm * N * (A + S)), horizon=Hm, cond=cond,
        )
        # every row the heads read: all Hm + 1 slots of the rollout, widened by the condition columns if there is one
        Fw = F_
        x_all = sv.featx.reshape((Hm + 1) * N, F_)
        if cond is not None:
            Fw = F_ + cond.shape[1]
            wide = torch.empty((Hm + 1) * N, Fw, device=dev)
            wide[:, :F_] = x_all
            wide.view(Hm + 1, N, Fw)[:, :, F_:] = cond
            x_all = wide
        feats = x_all[N:]
        # the reward of the LAST imagined step is never used (lambda_return reads r[:-1], common/utils.py:61-71; the
        # reference computes it all the same): the head runs on the first Hm - 1 steps' rows, forward and backward
        nr_ = (Hm - 1) * N
        r_pred = torch.empty(Hm * N, 1, device=dev)
        _, r_hid = ops.mlp_fwd(pw, feats[:nr_], out=r_pred[:nr_])
        v_pred, v_hid = ops.mlp_fwd(pv, feats)
        # -- action entropy on the (attached) imagined states (dreamer.py:320-324).  The reference
        #    re-runs the actor on imag[0..Hm-1]; rows of steps 1..Hm-1 are the very inputs the rollout
        #    already pushed through the actor (detaching does not change values), so only the final
        #    state is evaluated here, into the spare step slot of the rollout's saved activations.
        nl = sv.a_hidden.shape[0]
        tail = slice(Hm * N, (Hm + 1) * N)
        ops.mlp_fwd(pa, x_all[Hm * N:], out=sv.a_raw[tail], hid=[sv.a_hidden[l, tail] for l in range(nl)])
        ops.actor_head_fwd(sv.a_raw[tail], *a_consts, mean=sv.a_mean[tail], std=sv.a_std[tail])
        ent_rows = slice(N, (Hm + 1) * N)  # imagined steps 1..Hm
        mean2, std2 = sv.a_mean[ent_rows], sv.a_std[ent_rows]
        eps_ent = self._noise("entropy", (am._samples, Hm * N, A))
        ent_sum, dmean2, dstd2 = ops.tanh_normal_entropy(mean2, std2, eps_ent, gscale=-c.action_ent_coef / (Hm * gN),
                                                         noise=self._draw(am._samples * Hm * N * A), samples=am._samples)
        lat_sum, dpstd = ops.normal_entropy(sv.prior_std, gscale=-c.latent_ent_coef / (Hm * gN),
                                            want_grad=c.latent_ent_coef != 0)
        # -- lambda returns and the actor objective
        gret = -1.0 / ((Hm - 1) * gN)
        returns, dr, dv, ret_sum = ops.lambda_return(r_pred.view(Hm, N), v_pred.view(Hm, N), c.gamma, c.gae_lambda,
                                                     gret)
        # -- backward: heads -> entropy path (input gradient only) -> reverse rollout
        #    The value head is differentiated for TWO losses on the same rows and activations: the actor's objective
        #    through the lambda-returns into the imagined states (dreamer.py:343-359, weights frozen) and the critic's
        #    own loss on detached imag[:-1] against detached returns into its weights (dreamer.py:362-373).  A scalar
        #    head's reverse chain is per row a unit chain times the row's scalar, so ONE chain serves both
        #    (ops.mlp_bwd, dout_w): round 5 ran it twice -- the second time on a forked side stream, with its own weight
        #    packs -- REPO_VALUE_ONE_CHAIN=0 restores that form.
        dfeat = torch.empty(Hm * N, Fw, device=dev)
        nv = (Hm - 1) * N
        main = torch.cuda.current_stream(dev)
        side = self._ac_side_stream or main
        one_chain = os.environ.get("REPO_VALUE_ONE_CHAIN", "1") == "1"
        if one_chain:
            v_sums, dv2 = ops.scalar_nll(v_pred.view(-1)[:nv], returns.view(-1), None, 1.0 / ((Hm - 1) * gN))
            ops.mlp_bwd(pv, feats, v_hid, dv.view(Hm * N, 1), dparams=gv, dx=dfeat, dout_w=dv2.view(nv, 1))
        else:
            ops.mlp_bwd(pv, feats, v_hid, dv.view(Hm * N, 1), dparams=None, dx=dfeat)
        ops.mlp_bwd(pw, feats[:nr_], r_hid, dr.view(Hm * N, 1)[:nr_], dparams=None, dx=dfeat[:nr_], accumulate_dx=True)
        # -- the critic's step (one chain: only the optimiser step is left of it), forked onto a side stream: it runs
        #    while the reverse rollout (which fills only ~77 CUs) and the actor backward proceed on the main stream
        side.wait_stream(main)  # after the value head's backward above read the weights
        with torch.cuda.stream(side):
            if not one_chain:
                v_sums, dv2 = ops.scalar_nll(v_pred.view(-1)[:nv], returns.view(-1), None, 1.0 / ((Hm - 1) * gN))
                ops.mlp_bwd(pv, feats[:nv], [h[:nv] for h in v_hid], dv2.view(nv, 1), dparams=gv, dx=None)
            if self.dp is None:
                self.value_optimizer.clip_and_step(c.grad_clip_norm)
        # gradient at the actor trunk's output, all (Hm+1)*N rows: rollout path on steps 0..Hm-1
        # (written by the reverse rollout), entropy path on steps 1..Hm (added on top)
        d_out = torch.zeros((Hm + 1) * N, 2 * A, device=dev)
        draw2 = ops.actor_head_bwd(mean2, std2, dmean=dmean2, dstd=dstd2, min_std=a_consts[0], mean_scale=a_consts[2])
        ops.mlp_bwd(pa, feats, [sv.a_hidden[l, ent_rows] for l in range(nl)], draw2, dparams=None, dx=dfeat,
                    accumulate_dx=True)
        if cond is not None:   # the condition columns take a gradient nobody reads
            dfeat = dfeat[:, :F_].contiguous()
        ops.rssm_imagine_bwd(pr, sv, dfeat, dprior_std=dpstd, min_std=self.transition_model.min_std_dev,
                             a_min_std=a_consts[0], a_mean_scale=a_consts[2], d_araw=d_out)
        ops.actor_head_bwd(mean2, std2, dmean=dmean2, dstd=dstd2, min_std=a_consts[0], mean_scale=a_consts[2],
                           out=d_out[ent_rows], accumulate=True)
        # -- ONE actor-trunk backward over every row the actor saw: both gradient paths share the
        #    same forward activations, and the chain is linear in the output gradient
        ops.mlp_bwd(pa, x_all, [sv.a_hidden[l] for l in range(nl)], d_out, dparams=ga, accumulate_w=False, dx=None)
        if self.dp is not None:
            # ONE bucket for both optimisers (their gradients are the halves of self._ac_grad); the critic's
            # backward is joined first, both steps follow the exchange
            main.wait_stream(side)
            self._allreduce(self._ac_grad)
            self.value_optimizer.clip_and_step(c.grad_clip_norm)
        self.actor_optimizer.clip_and_step(c.grad_clip_norm)
        main.wait_stream(side)
        self._pending_ac = (ret_sum, ent_sum, lat_sum, v_sums, Hm, gN)
        self._log_update()

    # ------------------------------------------------------------------ logging: one D2H per update
    def _log_update(self):
        """Gather every logged scalar of this update into one device buffer and start ONE
        asynchronous device->host copy; the values are turned into floats (and handed to the
        logger) when they are first needed: `last_scalars`, or the next update's log call."""
        # the PREVIOUS update's log is read here; if it reports a fault, the exception leaves only after this
        # update's own copy has been enqueued (an update that is already in flight keeps its log)
        fault = self._flush_log(defer=True)
        msc, dual, grow = self._pending_model   # [nll, rsq, rmask, kl, model_sqnorm] (+ dual[4])
        ret_sum, ent_sum, lat_sum, v_sums, Hm, gN = self._pending_ac
        cur = torch.cuda.current_stream(self.device)
        msc.record_stream(cur)
        if dual is not None:
            dual.record_stream(cur)
        # a sibling algorithm's own loss sums / gradient norms (TIA): behind the shared ones of each kind
        xs, xn = self._pending_extra if self._pending_extra is not None else (msc[:0], msc[:0])
        self._pending_extra = None
        parts = [msc[:4], ret_sum, ent_sum, lat_sum, v_sums, xs, msc[4:5], self.actor_optimizer.sqnorm,
                 self.value_optimizer.sqnorm, xn]
        if dual is not None:
            parts.append(dual)
        # LAST: this update's status word (bits reinterpreted, the copy is exact; already global under data
        # parallelism), see _take_status
        parts.append(self._ustatus.view(torch.float32))
        buf = torch.cat([p.reshape(-1) for p in parts])
        # the leading entries (losses) are per-rank partial sums; the gradient norms behind them are already global
        self._allreduce_scalars(buf, n_sum=sum(p.numel() for p in parts[:6]))
        n = buf.numel()
        self._log_host[:n].copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(cur)
        self._log_pending = (ev, n, dual.numel() if dual is not None else 0, grow, Hm, gN, xs.numel(), xn.numel(),
                             self._restore_point)
        if fault is not None:
            raise fault

    def _dual_scalars(self, out, dual):
        """The dual step's scalars (RePo: repo.py:99-105; per-task betas: repo_mt.py:100-112)."""
        kl_div, kl_loss, beta_loss, beta = dual
        out["train/kl_loss"] = kl_loss
        out["train/kl_div"] = kl_div
        out["train/beta"] = beta
        out["train/beta_loss"] = beta_loss

    def _extra_scalars(self, out, sums, norms, grow):
        """Hook of the sibling algorithms: rewrite / add logged scalars from their own sums (see _log_update)."""

    def _flush_log(self, defer=False):
        """defer: return a fault's exception instead of raising it (see _log_update)."""
        if self._log_pending is None:
            return None
        ev, n, n_dual, grow, Hm, gN, nxs, nxn, restore = self._log_pending
        self._log_pending = None
        ev.synchronize()
        c = self.c
        try:
            self._raise_update_fault(int(self._log_host[n - 1 : n].view(torch.int32).item()), restore)
        except Exception as fault:  # noqa: BLE001  (RepoHipError)
            if defer:
                return fault
            raise
        h = self._log_host[: n - 1].tolist()
        nll, rsq, rmask, kl, ret, ent, lat, vsq, _vn = h[:9]
        xsums, h = h[9 : 9 + nxs], h[:9] + h[9 + nxs :]
        gm, ga_, gv_ = h[9:12]
        xnorms, h = h[12 : 12 + nxn], h[:12] + h[12 + nxn :]
        npix = self._npix
        out = {}
        out["train/obs_loss"] = nll / grow + 0.5 * LOG_2PI * npix
        out["train/reward_loss"] = (rsq + 0.5 * LOG_2PI * rmask) / grow
        if n_dual:
            self._dual_scalars(out, h[12 : 12 + n_dual])
        else:
            out["train/kl_loss"] = kl / grow
        out["train/model_loss"] = out["train/obs_loss"] + out["train/reward_loss"] + out["train/kl_loss"]
        self._extra_scalars(out, xsums, xnorms, grow)
        action_entropy = ent / (Hm * gN)
        latent_entropy = lat / (Hm * gN) + self._LATENT_ENTROPY_SHIFT * self.c.state_size
        out["train/actor_loss"] = (-ret / ((Hm - 1) * gN) - c.action_ent_coef * action_entropy
                                   - c.latent_ent_coef * latent_entropy)
        out["train/value_loss"] = vsq / ((Hm - 1) * gN) + 0.5 * LOG_2PI
        out["train/action_entropy"] = action_entropy
        out["train/latent_entropy"] = latent_entropy
        self._last_scalars = out
        self.last_grad_norms = {"model": math.sqrt(max(gm, 0.0)), "actor": math.sqrt(max(ga_, 0.0)),
                                "value": math.sqrt(max(gv_, 0.0))}
        for k, v in out.items():
            self.logger.record(k, v)
        return None

    def _raise_update_fault(self, word, restore):
        """word != 0: a scan of that update timed out.  Every optimiser step of the update has skipped itself on the
        device (parameters, moments, dual variables unchanged -- on every rank of a data-parallel job, which all get
        here in the same update).  If no later update has been started the host-side counters are rolled back too
        (Adam step counts for the bias correction, the Philox offset), so that calling the update again -- e.g. under
        REPO_SCAN_CS=0 -- is exactly the update that failed; with a later update already in flight (train_agent's
        pipelining) they stay: that update ran on the unchanged parameters with its bias correction one step ahead."""
        if not word:
            return
        rolled = restore is not None and restore[0] == self._update_seq
        if rolled:
            self._noise_counter = restore[1]
            for opt, n in restore[2]:
                opt.step_count = n
        ops.raise_scan_status(word, "the update's optimiser steps were skipped: parameters, Adam moments and dual "
                              "variables are unchanged" + ("; step counts and the noise offset were rolled back, the "
                              "update can be retried as it was" if rolled else " (a later update was already in flight)"))

    @property
    def last_scalars(self):
        """Logged scalars of the most recent update (waits for it to finish on the device)."""
        self._flush_log()
        return self._last_scalars

    def synchronize(self):
        """Join the update lanes into the current stream (before reading parameters elsewhere)."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self._wm_stream)
        cur.wait_stream(self._ac_stream)

    def _allreduce_scalars(self, buf, n_sum):
        """Loss sums are per-rank partial sums; gradient norms (already global) are not summed."""
        if self.dp is not None:
            self.dp.all_reduce_prefix(buf, n_sum)

    # ------------------------------------------------------------------ update loop
    def update(self, batch, join=True):
        """One iteration of the train_agent loop body on a device batch
        (obs (L,B,3,64,64) uint8|float32, actions (L,B,A), rewards (L,B,1), dones (L,B,1)).

        The two halves run on two streams.  train_actor_critic(k) only READS the world model and
        train_dynamics(k+1) does not touch the actor or the critic, so the only orderings that matter
        are: AC(k) after the model optimiser step of WM(k), and the model optimiser step of WM(k+1)
        after AC(k).  Both are events; everything else of WM(k+1) -- encoder, scan, decoder, all of
        the backward -- overlaps AC(k), whose rollout kernels fill less than a third of the CUs.
        Results are identical to running the halves back to back; nothing here synchronises the
        host (scalars are read lazily through `last_scalars`).  join=True (default) makes the
        caller's stream wait for both lanes, so parameters can be read right after the call;
        train_agent()'s loop passes join=False and joins once at the end, which is what lets
        consecutive updates overlap."""
        obs, actions, rewards, dones = batch
        dev = self.device
        caller = torch.cuda.current_stream(dev)
        wm, ac = self._wm_stream, self._ac_stream
        wm.wait_stream(caller)  # the batch was produced on the caller's stream
        with torch.cuda.stream(wm):
            nonterms = 1.0 - dones.float()
            beliefs, post = self.train_dynamics(obs, actions, rewards, nonterms)
            ev_wm = torch.cuda.Event()
            ev_wm.record(wm)
        for t in (obs, actions, rewards, dones):
            t.record_stream(wm)
        with torch.cuda.stream(ac):
            ac.wait_event(ev_wm)
            beliefs.record_stream(ac)  # views of the scan's feature buffer, allocated on the WM stream
            try:
                self.train_actor_critic(beliefs.flatten(0, 1), post.flatten(0, 1))
            finally:
                # also when the previous update's fault is raised from this one's log call: the next model step
                # must still wait for this update's imagination
                self._ev_ac_done = torch.cuda.Event()
                self._ev_ac_done.record(ac)
        if join:
            self.synchronize()

    def train_agent(self):
        c = self.c
        B, L = c.batch_size, c.chunk_size
        h = self.buffer.prefetch(B, L, self.device)
        for i in range(c.train_steps):
            batch = self.buffer.acquire(h, B, L, self.device)
            cur = h
            self.update(batch, join=False)
            with torch.cuda.stream(self._wm_stream):  # the consumer of the staged batch
                self.buffer.release(cur, B, L, self.device)
            if i + 1 < c.train_steps:
                # host gather + PCIe copy of the next batch overlap the update just enqueued
                h = self.buffer.prefetch(B, L, self.device)
        self.synchronize()
        # hand the LAST update's scalars to the logger too (the reference records every update before its
        # next logger.dump); the acting step that follows synchronises with the device anyway
        self._flush_log()

    # ------------------------------------------------------------------ acting
    def collect_seed_data(self):
        """Random-policy prefill of the replay ring; stops only at an episode boundary once `prefill`
        transitions are stored (reference dreamer.py:159-167)."""
        env, ring = self.env, self.buffer
        obs, mid_episode = env.reset(), True
        while mid_episode or len(ring) < self.c.prefill:
            action = env.action_space.sample()
            following, reward, done, _ = env.step(action)
            ring.push(obs, action, reward, done)
            mid_episode = not done
            obs = env.reset() if done else following

    def init_latent_and_action(self):
        shapes = (self.c.belief_size, self.c.state_size, self.action_size)
        return tuple(torch.zeros(1, n, device=self.device) for n in shapes)

    @torch.no_grad()
    def _act_eager(self, belief, posterior_state, action, obs, explore):
        embed = self.encoder(obs)
        outs = self.transition_model.observe(belief, posterior_state, action.unsqueeze(0), embed.unsqueeze(0))
        belief, posterior_state = outs[0].squeeze(0), outs[4].squeeze(0)
        action = self.actor_model.get_action(belief, posterior_state, det=not explore)
        if explore:   # drawn and clamped unconditionally, as the reference does (dreamer.py:193-195): with
            # action_noise == 0 the draw still advances torch's generator, so seeded runs stay in step
            action = torch.clamp(action + torch.randn_like(action) * self.c.action_noise, -1, 1)
        return belief, posterior_state, action

    def update_latent_and_select_action(self, belief, posterior_state, action, obs, explore=False):
        """One filtering step + policy (reference dreamer.py:175-196).

        This runs once per environment step (500 k times per run) on one frame: ~35 tiny kernels whose
        cost is launch latency, not work.  The whole step is therefore captured ONCE per (explore,
        batch) into a HIP graph over static input/output buffers and replayed (783 -> ~250 us per step
        incl. the action's D2H).  Parameters are updated in place by the optimisers, so the captured
        pointers stay valid; the noise kernels advance torch's Philox offset on every replay.
        REPO_ACT_GRAPH=0 falls back to eager launches."""
        self.synchronize()
        if not self._act_graph_enabled:
            with torch.no_grad():
                return self._act_eager(belief, posterior_state, action, obs, explore)
        key = (bool(explore), int(obs.shape[0]), obs.dtype)
        g = self._act_graphs.get(key)
        if g is None:
            g = self._capture_act_graph(belief, posterior_state, action, obs, bool(explore))
            self._act_graphs[key] = g
        graph, sin, sout, _scratch = g
        for dst, src in zip(sin, (belief, posterior_state, action, obs)):
            dst.copy_(src)
        graph.replay()
        return tuple(t.clone() for t in sout)

    def _capture_act_graph(self, belief, posterior_state, action, obs, explore):
        """-> (graph, static inputs, static outputs, scratch): the graph OWNS the scratch its kernels were captured with
        (ops.capture_graph) -- the process-wide per-stream workspaces never enter a captured region."""
        sin = tuple(t.detach().to(self.device).clone().contiguous() for t in (belief, posterior_state, action, obs))
        graph, sout, scratch = ops.capture_graph(lambda: self._act_eager(*sin, explore), device=self.device)
        return graph, sin, sout, scratch

    def _dump_log(self):
        self.logger.record("train/step", self.step)
        self.logger.dump(step=self.step)

    def train(self):
        """Interleave environment steps with agent updates (reference dreamer.py:403-455): every
        environment step stores one transition; training, evaluation, checkpointing and log dumps fire
        on their own periods of the step counter, in that order."""
        c = self.c
        if c.load_checkpoint:
            self.load_checkpoint()
        if len(self.buffer) == 0:
            self.collect_seed_data()
        periodic = ((c.train_every, self.train_agent), (c.eval_every, self.eval_agent),
                    (c.checkpoint_every, self.save_checkpoint), (c.log_every, self._dump_log))
        driver = EpisodeDriver(self, self.env, explore=True)
        driver.begin()
        while self.step < c.num_steps:
            tr = driver.advance()
            self.buffer.push(tr.obs, tr.action, tr.reward, tr.done)
            if tr.done:
                driver.report("train")
                driver.begin()
            for period, job in periodic:
                if self.step % period == 0:
                    job()
            self.step += 1

    def _reconstruct(self, belief, state):
        return self.obs_model(belief, state)

    def eval_agent(self):
        """One deterministic-policy episode on eval_env; logs return, success and a side-by-side video of
        observed and reconstructed frames (reference dreamer.py:457-490)."""
        self.toggle_train(False)
        driver = EpisodeDriver(self, self.eval_env, explore=False)
        driver.begin()
        pairs = []
        finished = False
        while not finished:
            tr = driver.advance()
            if self.c.pixel_obs:
                with torch.no_grad():
                    recon = self._reconstruct(driver.latent[0], driver.latent[1])
                pairs.append([tr.obs, postprocess(to_np(recon))[0]])
            finished = tr.done
        driver.report("test")
        if pairs:
            clip = np.stack(pairs).transpose(1, 0, 2, 3, 4)  # (T, 2, C, H, W) -> (2, T, C, H, W)
            self.logger.record("test/video", _as_video(clip, fps=30), exclude="stdout")
        self.toggle_train(True)

    # ------------------------------------------------------------------ checkpoints (reference key layout)
    def save_checkpoint(self):
        torch.save(self.get_param_dict(), os.path.join(self.logger.dir, "models.pt"))
        if self.c.save_buffer:
            self.buffer.save(os.path.join(self.logger.dir, "buffer.npz"))

    def get_param_dict(self):
        self.synchronize()

        def sd(m):
            return {k: v.detach().clone() for k, v in m.state_dict().items()}

        return {
            "step": self.step,
            "encoder": sd(self.encoder),
            "transition_model": sd(self.transition_model),
            "obs_model": sd(self.obs_model),
            "reward_model": sd(self.reward_model),
            "actor_model": sd(self.actor_model),
            "value_model": sd(self.value_model),
            "model_optimizer": self.model_optimizer.state_dict(),
            "actor_optimizer": self.actor_optimizer.state_dict(),
            "value_optimizer": self.value_optimizer.state_dict(),
        }

    def load_checkpoint(self, ckpt_dir=None):
        if ckpt_dir is None:
            ckpt_dir = self.logger.dir
        buffer_path = os.path.join(ckpt_dir, "buffer.npz")
        if os.path.exists(buffer_path):
            self.buffer.load(buffer_path)
            print(f"Loaded buffer from {buffer_path}")
        elif self.c.load_offline:
            self.load_offline_data()
        params_path = os.path.join(ckpt_dir, "models.pt")
        if os.path.exists(params_path):
            params = torch.load(params_path, map_location=self.device, weights_only=False)
            self.load_param_dict(params)
            print(f"Loaded parameters from {params_path}")

    def _load_module(self, module, sd):
        """copy_ into the existing (flat-buffer backed) parameters; never rebinds storage."""
        own = module.state_dict()
        assert set(own.keys()) == set(sd.keys()), (sorted(own.keys()), sorted(sd.keys()))
        with torch.no_grad():
            for k, v in own.items():
                v.copy_(sd[k].to(v.device))

    def load_param_dict(self, params):
        self.step = params["step"]
        for name in ("encoder", "transition_model", "obs_model", "reward_model", "actor_model", "value_model"):
            self._load_module(getattr(self, name), params[name])
        self.model_optimizer.load_state_dict(params["model_optimizer"])
        self.actor_optimizer.load_state_dict(params["actor_optimizer"])
        self.value_optimizer.load_state_dict(params["value_optimizer"])
        # the checkpoint keeps the reference's key layout (no noise state in it): resume behind every normal the
        # saved run can have drawn, instead of replaying the first updates' noise at offset 0
        self._noise_counter = max(self._noise_counter, self.model_optimizer.step_count * self._noise_stride())

    def load_offline_data(self):
        """Replace the replay ring by the concatenation of every `buffer*.npz` under c.offline_dir
        (reference dreamer.py:566-596); the work is host-only: common/buffers.py."""
        paths = list(glob.glob(os.path.join(self.c.offline_dir, "buffer*.npz")))
        self.buffer.adopt_offline(paths, self.c.offline_truncate_size)
        for path in paths:
            print(f"Loaded buffer from {path}")
