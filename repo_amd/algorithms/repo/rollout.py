"""Stepping an environment with the agent's filter + policy (host side).

The reference inlines the same dozen lines in `train()` and `eval_agent()`
(/root/reference/algorithms/repo/dreamer.py:403-455 and :457-490): preprocess the frame, one
`update_latent_and_select_action`, one `env.step`, running return / success totals, re-initialise
the latent on episode end.  Here that is one object both loops drive; each `advance()` costs one
HIP-graph replay of the acting path plus the action's device->host copy.
"""
from collections import namedtuple

import numpy as np

from ...common.utils import preprocess, to_np, to_torch

Transition = namedtuple("Transition", "obs action reward done")


class EpisodeDriver:
    def __init__(self, agent, env, explore):
        self.agent, self.env, self.explore = agent, env, bool(explore)
        self.latent = None      # (belief, posterior_state, action) device tensors
        self.obs = None         # the observation the next action will be chosen from
        self.episode_return = 0
        self.episode_success = 0

    def begin(self):
        """Start an episode: zero latent / previous action, fresh observation, cleared totals."""
        self.latent = self.agent.init_latent_and_action()
        self.obs = self.env.reset()
        self.episode_return = 0
        self.episode_success = 0

    def advance(self):
        """Filter on the current observation, act, step the environment once."""
        seen = self.obs
        frame = to_torch(preprocess(seen[None]), device=self.agent.device)
        self.latent = self.agent.update_latent_and_select_action(*self.latent, frame, self.explore)
        action = to_np(self.latent[2])[0]
        if not np.isfinite(action).all():
            # fail loudly: a non-finite action can only come from a broken acting path (round 5's driver-box garbage,
            # DESIGN section 5a), and pushed into the ring it would poison every later batch
            raise FloatingPointError(f"the acting path returned a non-finite action {action!r} at environment step "
                                     f"{getattr(self.agent, 'step', '?')}")
        self.obs, reward, done, info = self.env.step(action)
        self.episode_return += reward
        self.episode_success += info.get("success", 0)
        return Transition(seen, action, reward, done)

    def report(self, prefix):
        """`<prefix>/return` and `<prefix>/success` of the episode just finished."""
        log = self.agent.logger
        log.record(f"{prefix}/return", self.episode_return)
        log.record(f"{prefix}/success", float(self.episode_success > 0))
