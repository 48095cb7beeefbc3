"""Value head and tanh-Normal actor (reference: algorithms/repo/models/actor_critic.py:9-102,
models/utils.py:112-166)."""
import torch
import torch.nn as nn

from .... import ops
from .decoder import _ScalarHead


class ValueModel(_ScalarHead):
    pass


class ActorModel(nn.Module):
    """230 -> hidden^4 -> 2A ELU MLP; mean = mean_scale*tanh(m/mean_scale),
    std = softplus(s + init_std) + min_std; action ~ tanh(Normal(mean, std)).

    The reference passes `dense_activation_function` positionally into the `dist` slot
    (dreamer.py:99-105), leaving the activation at its default "elu"; the same signature is kept."""

    def __init__(self, belief_size, state_size, hidden_size, action_size, dist="tanh_normal",
                 activation_function="elu", min_std=0.1, init_std=0.0, mean_scale=5):
        super().__init__()
        if activation_function != "elu":
            raise NotImplementedError("HIP MLP kernels fuse ELU")
        self.fc1 = nn.Linear(belief_size + state_size, hidden_size)
        self.fc2 = nn.Linear(hidden_size, hidden_size)
        self.fc3 = nn.Linear(hidden_size, hidden_size)
        self.fc4 = nn.Linear(hidden_size, hidden_size)
        self.fc5 = nn.Linear(hidden_size, 2 * action_size)
        self._dist = dist
        self._min_std = min_std
        self._init_std = init_std
        self._mean_scale = mean_scale
        self._samples = 100  # SampleDist default (models/utils.py:138)

    def plist(self):
        return [t for m in (self.fc1, self.fc2, self.fc3, self.fc4, self.fc5) for t in (m.weight, m.bias)]

    @torch.no_grad()
    def forward(self, belief, state):
        feat = torch.cat([belief, state], dim=1).contiguous()
        raw, _ = ops.mlp_fwd([t.detach() for t in self.plist()], feat)
        mean, std, _ = ops.actor_head_fwd(raw, self._min_std, self._init_std, float(self._mean_scale))
        return mean, std

    @torch.no_grad()
    def get_action(self, belief, state, det=False, eps=None):
        """rsample of the policy (det=False) or SampleDist.mode (det=True): the sample with the
        highest log-probability among `_samples` draws (models/utils.py:149-158)."""
        feat = torch.cat([belief, state], dim=1).contiguous()
        raw, _ = ops.mlp_fwd([t.detach() for t in self.plist()], feat)
        if not det:
            if eps is None:
                eps = torch.randn(raw.shape[0], raw.shape[1] // 2, device=raw.device)
            S = state.shape[1]
            _, _, xsa = ops.actor_head_fwd(raw, self._min_std, self._init_std, float(self._mean_scale), eps=eps,
                                           state=feat[:, belief.shape[1]:])
            return xsa[:, S:].contiguous()
        from ..autograd import tanh_normal_mode

        mean, std, _ = ops.actor_head_fwd(raw, self._min_std, self._init_std, float(self._mean_scale))
        return tanh_normal_mode(mean, std, self._samples, eps)
