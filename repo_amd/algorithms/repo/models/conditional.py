"""Task-conditioned modules of the multitask agents.

Reference: ConditionalVisualEncoder (/root/reference/algorithms/repo/models/encoder.py:68-88),
ConditionalVisualObservationModel (models/decoder.py:96-123), ConditionalRewardModel (models/decoder.py:198-213),
ConditionalTransitionModel (models/rssm.py:187-249), ConditionalActorModel / ConditionalValueModel
(models/actor_critic.py:28-55,104-139).  Same constructors, children and state_dict names, so a reference
`models.pt` of a multitask agent loads key for key.

Two forms of conditioning:
 * dense modules CONCATENATE the condition (task one-hot): `cat([belief, state, cond])` for the reward / value / actor
   heads, `cat([actions, cond])` pseudo-actions for the RSSM (the pixel decoder's fc1 does NOT: decoder.py:116).  No kernel of its own: the condition is C more
   K columns of the rows the dense kernels already read;
 * the conv stacks are FiLM-modulated: `film = Linear(cond, 2 * channels)`, each conv output becomes
   `relu((1 + gamma) * conv(x) + beta)` per (frame, channel) -- repo_film_fwd / repo_film_bwd around the conv kernels
   run with the bias-only epilogue (repo_amd/functional_mt.py).

The modules are parameter containers; `forward` evaluates values without an autograd graph (the acting path and
evaluation use it) -- training is scheduled by hand in algorithms/repo/dreamer_mt.py.
"""
import torch
import torch.nn as nn

from .... import ops
from .actor_critic import ActorModel, ValueModel
from .decoder import RewardModel, VisualObservationModel
from .encoder import VisualEncoder
from .rssm import TransitionModel

ENC_CHANNELS = (32, 64, 128, 256)
DEC_CHANNELS = (128, 64, 32)


def film_offsets(channels):
    """Column offsets (gamma_l, beta_l) of each modulated layer in the FiLM layer's output: the reference chunks the
    output in two halves [gammas | betas] and splits each by the layers' channel counts."""
    total, offs, o = sum(channels), [], 0
    for c in channels:
        offs.append((o, total + o))
        o += c
    return offs


class ConditionalVisualEncoder(VisualEncoder):
    def __init__(self, embedding_size, condition_size, activation_function="relu"):
        super().__init__(embedding_size, activation_function)
        self.condition_size = condition_size
        self.film = nn.Linear(condition_size, 2 * sum(ENC_CHANNELS))

    def plist(self):
        return super().plist() + [self.film.weight, self.film.bias]

    @torch.no_grad()
    def forward(self, observation, condition):
        from .... import functional_mt as Fm

        embeds, _ = Fm.cond_encoder_fwd([t.detach() for t in self.plist()], observation.contiguous(),
                                        condition.float().contiguous())
        return embeds


class ConditionalVisualObservationModel(VisualObservationModel):
    def __init__(self, belief_size, state_size, embedding_size, condition_size, activation_function="relu"):
        super().__init__(belief_size, state_size, embedding_size, activation_function)
        self.condition_size = condition_size
        self.film = nn.Linear(condition_size, 2 * sum(DEC_CHANNELS))

    def plist(self):
        return super().plist() + [self.film.weight, self.film.bias]

    @torch.no_grad()
    def forward(self, belief, state, condition):
        from .... import functional_mt as Fm

        feat = torch.cat([belief, state], dim=1).contiguous()
        recon, _ = Fm.cond_decoder_fwd([t.detach() for t in self.plist()], feat, condition.float().contiguous())
        return recon


def ConditionalEncoder(symbolic, observation_size, embedding_size, condition_size, activation_function="relu"):
    if symbolic:
        raise NotImplementedError("symbolic (non-pixel) observations are outside the MI355X hot path")
    if int(observation_size[-1]) != 64:
        raise NotImplementedError("the conditional conv stacks are built for the reference's 64 x 64 frames")
    return ConditionalVisualEncoder(embedding_size, condition_size, activation_function)


def ConditionalObservationModel(symbolic, observation_size, belief_size, state_size, embedding_size, condition_size,
                                activation_function="relu"):
    if symbolic:
        raise NotImplementedError("symbolic (non-pixel) observations are outside the MI355X hot path")
    return ConditionalVisualObservationModel(belief_size, state_size, embedding_size, condition_size, activation_function)


def _wide(belief, state, condition):
    return torch.cat([belief, state, condition.float()], dim=1).contiguous()


class ConditionalRewardModel(RewardModel):
    """RewardModel on cat([belief, cat([state, condition])]) (models/decoder.py:198-213)."""

    def __init__(self, belief_size, state_size, hidden_size, condition_size, activation_function="relu"):
        super().__init__(belief_size, state_size + condition_size, hidden_size, activation_function)
        self.condition_size = condition_size

    @torch.no_grad()
    def forward(self, belief, state, condition):
        out, _ = ops.mlp_fwd([t.detach() for t in self.plist()], _wide(belief, state, condition))
        return out.squeeze(dim=1)


class ConditionalValueModel(ValueModel):
    def __init__(self, belief_size, state_size, hidden_size, condition_size, activation_function="relu"):
        super().__init__(belief_size, state_size + condition_size, hidden_size, activation_function)
        self.condition_size = condition_size

    @torch.no_grad()
    def forward(self, belief, state, condition):
        out, _ = ops.mlp_fwd([t.detach() for t in self.plist()], _wide(belief, state, condition))
        return out.squeeze(dim=1)


class ConditionalActorModel(ActorModel):
    def __init__(self, belief_size, state_size, hidden_size, action_size, condition_size, dist="tanh_normal",
                 activation_function="elu", min_std=0.1, init_std=0.0, mean_scale=5):
        super().__init__(belief_size, state_size + condition_size, hidden_size, action_size, dist, activation_function,
                         min_std, init_std, mean_scale)
        self.condition_size = condition_size

    @torch.no_grad()
    def forward(self, belief, state, condition):
        return super().forward(belief, torch.cat([state, condition.float()], dim=1))

    @torch.no_grad()
    def get_action(self, belief, state, condition, det=False, eps=None):
        # the base class evaluates cat([belief, state']) and slices the action behind state' = [state | condition]
        return super().get_action(belief, torch.cat([state, condition.float()], dim=1), det=det, eps=eps)


class ConditionalTransitionModel(TransitionModel):
    """TransitionModel over pseudo-actions [action | condition] (models/rssm.py:187-249)."""

    def __init__(self, belief_size, state_size, action_size, hidden_size, embedding_size, condition_size,
                 activation_function="relu", min_std_dev=0.1):
        super().__init__(belief_size, state_size, action_size + condition_size, hidden_size, embedding_size,
                         activation_function, min_std_dev)
        self.condition_size = condition_size
        self.real_action_size = action_size

    def observe(self, prev_belief, prev_state, actions, conditions, observations=None, nonterminals=None, noise=None):
        pseudo = torch.cat((actions, conditions.float()), dim=2)
        return super().observe(prev_belief, prev_state, pseudo, observations, nonterminals, noise)

    @torch.no_grad()
    def imagine(self, prev_belief, prev_state, condition, policy, horizon, noise=None):
        N = prev_belief.shape[0]
        dev = prev_belief.device
        A, S, D = self.real_action_size, self.state_size, self.belief_size
        if noise is None:
            noise = (torch.randn(horizon - 1, N, A, device=dev), torch.randn(horizon - 1, N, S, device=dev))
        sv = ops.rssm_imagine_fwd([t.detach() for t in self.plist()], [t.detach() for t in policy.plist()],
                                  prev_belief.contiguous(), prev_state.contiguous(), noise[0], noise[1],
                                  self.min_std_dev, policy._min_std, policy._init_std, float(policy._mean_scale),
                                  cond=condition.float().contiguous())
        return [sv.featx[1:, :, :D], sv.featx[1:, :, D:], sv.prior_mean, sv.prior_std]
