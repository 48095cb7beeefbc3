"""Recurrent state-space model (reference: algorithms/repo/models/rssm.py:8-184)."""
import torch
import torch.nn as nn


class TransitionModel(nn.Module):
    """Linear(s+a -> belief) -> ELU -> GRUCell(belief, belief); prior MLP belief -> hidden -> 2s;
    posterior MLP (belief + embedding) -> hidden -> 2s; std = softplus(.) + min_std_dev.

    Children are parameter containers (reference constructors, reference state_dict names);
    `observe` runs the fused HIP scan, `imagine` the HIP rollout."""

    def __init__(self, belief_size, state_size, action_size, hidden_size, embedding_size,
                 activation_function="relu", min_std_dev=0.1):
        super().__init__()
        if activation_function != "elu":
            raise NotImplementedError("HIP RSSM kernels fuse ELU (dense_activation_function='elu')")
        self.min_std_dev = min_std_dev
        self.belief_size, self.state_size, self.action_size = belief_size, state_size, action_size
        self.hidden_size, self.embedding_size = hidden_size, embedding_size
        self.fc_embed_state_action = nn.Linear(state_size + action_size, belief_size)
        self.rnn = nn.GRUCell(belief_size, belief_size)
        self.fc_embed_belief_prior = nn.Linear(belief_size, hidden_size)
        self.fc_state_prior = nn.Linear(hidden_size, 2 * state_size)
        self.fc_embed_belief_posterior = nn.Linear(belief_size + embedding_size, hidden_size)
        self.fc_state_posterior = nn.Linear(hidden_size, 2 * state_size)

    def plist(self):
        r = self.rnn
        out = [self.fc_embed_state_action.weight, self.fc_embed_state_action.bias, r.weight_ih, r.weight_hh, r.bias_ih,
               r.bias_hh]
        for m in (self.fc_embed_belief_prior, self.fc_state_prior, self.fc_embed_belief_posterior,
                  self.fc_state_posterior):
            out += [m.weight, m.bias]
        return out

    def observe(self, prev_belief, prev_state, actions, observations=None, nonterminals=None, noise=None):
        """-> [beliefs, prior_states, prior_means, prior_std_devs, posterior_states, posterior_means,
        posterior_std_devs], each (T, B, .).  `noise` = (eps_prior, eps_post) of shape (T,B,S)
        replaces the internally drawn standard-normal noise (for parity runs)."""
        from ..autograd import observe_apply

        if observations is None:
            # open-loop rollout under GIVEN actions (reference rssm.py:118: the next step is fed the prior sample):
            # -> [beliefs, prior_states, prior_means, prior_std_devs].  FORWARD VALUES ONLY: the kernel's prior-only
            # mode has no backward, so a caller that expects the reference's differentiable branch is told so
            # instead of silently receiving detached tensors.
            if torch.is_grad_enabled() and any(
                    t is not None and t.requires_grad for t in (prev_belief, prev_state, actions, nonterminals)):
                raise NotImplementedError(
                    "TransitionModel.observe(observations=None) computes forward values only (no backward through "
                    "the open-loop rollout); call it under torch.no_grad() or detach its inputs")
            return self._observe_prior_only(prev_belief, prev_state, actions, nonterminals, noise)
        return observe_apply(self, prev_belief, prev_state, actions, observations, nonterminals, noise)

    @torch.no_grad()
    def _observe_prior_only(self, prev_belief, prev_state, actions, nonterminals, noise):
        from .... import ops

        T, B = actions.shape[:2]
        dev = actions.device
        S, D = self.state_size, self.belief_size
        nt = torch.ones(T, B, device=dev) if nonterminals is None else nonterminals.reshape(T, B).float()
        eps = noise[0] if noise is not None else torch.randn(T, B, S, device=dev)
        sv = ops.rssm_observe_fwd([t.detach() for t in self.plist()], prev_belief.contiguous(), prev_state.contiguous(),
                                  actions.contiguous(), nt.contiguous(), torch.zeros(T, B, self.embedding_size, device=dev),
                                  eps.contiguous(), torch.zeros(T, B, S, device=dev), self.min_std_dev, prior_only=True)
        return [sv.featx[1:, :, :D], sv.prior_state, sv.prior_mean, sv.prior_std]

    @torch.no_grad()
    def imagine(self, prev_belief, prev_state, policy, horizon, noise=None):
        """-> [beliefs, prior_states, prior_means, prior_std_devs], each (horizon-1, N, .).
        Forward values only: the gradient path through the rollout is scheduled by hand in
        Dreamer.train_actor_critic (repo_rssm_imagine_bwd)."""
        from .... import ops

        N = prev_belief.shape[0]
        dev = prev_belief.device
        A, S = self.action_size, self.state_size
        if noise is None:
            noise = (torch.randn(horizon - 1, N, A, device=dev), torch.randn(horizon - 1, N, S, device=dev))
        sv = ops.rssm_imagine_fwd([t.detach() for t in self.plist()], [t.detach() for t in policy.plist()],
                                  prev_belief.contiguous(), prev_state.contiguous(), noise[0], noise[1],
                                  self.min_std_dev, policy._min_std, policy._init_std, float(policy._mean_scale))
        D = self.belief_size
        return [sv.featx[1:, :, :D], sv.featx[1:, :, D:], sv.prior_mean, sv.prior_std]
