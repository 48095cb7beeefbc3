"""Transposed-conv observation model and reward head
(reference: algorithms/repo/models/decoder.py:28-48,178-195)."""
import torch.nn as nn


class VisualObservationModel(nn.Module):
    """Linear(230->1024) -> convT 1024->128 (k5) -> 64 (k5) -> 32 (k6) -> 3 (k6), stride 2, ReLU
    between.  Children hold parameters only; see repo_amd.functional.decoder_*."""

    def __init__(self, belief_size, state_size, embedding_size, activation_function="relu", image_size=64):
        """image_size=128 (build-defined, see VisualEncoder): conv4 becomes 32 -> 16 (k6, 30 -> 64, ReLU) and
        conv5 = 16 -> 3 (k2, 64 -> 128) is the output layer."""
        super().__init__()
        if activation_function != "relu":
            raise NotImplementedError("HIP decoder kernels fuse ReLU (cnn_activation_function='relu')")
        if embedding_size != 1024:
            raise NotImplementedError("embedding_size != 1024 is not on the hot path")
        if image_size not in (64, 128):
            raise NotImplementedError(f"{image_size} x {image_size} frames: only 64 (the reference) and 128 (build-defined) are built")
        self.embedding_size = embedding_size
        self.image_size = image_size
        self.fc1 = nn.Linear(belief_size + state_size, embedding_size)
        self.conv1 = nn.ConvTranspose2d(embedding_size, 128, 5, stride=2)
        self.conv2 = nn.ConvTranspose2d(128, 64, 5, stride=2)
        self.conv3 = nn.ConvTranspose2d(64, 32, 6, stride=2)
        if image_size == 64:
            self.conv4 = nn.ConvTranspose2d(32, 3, 6, stride=2)
        else:
            self.conv4 = nn.ConvTranspose2d(32, 16, 6, stride=2)
            self.conv5 = nn.ConvTranspose2d(16, 3, 2, stride=2)

    def plist(self):
        mods = (self.fc1, self.conv1, self.conv2, self.conv3, self.conv4) + ((self.conv5,) if self.image_size == 128 else ())
        return [t for m in mods for t in (m.weight, m.bias)]

    def forward(self, belief, state):
        from ..autograd import decoder_apply

        return decoder_apply(self, belief, state)


class TIAObservationModel(nn.Module):
    """TIA's decoder (reference models/decoder.py:154-175): the visual decoder with a 6-channel output layer,
    returned as (recon, mask) = out.chunk(2, 1).  64 x 64 frames only (as the reference)."""

    def __init__(self, belief_size, state_size, embedding_size, activation_function="relu"):
        super().__init__()
        if activation_function != "relu":
            raise NotImplementedError("HIP decoder kernels fuse ReLU (cnn_activation_function='relu')")
        if embedding_size != 1024:
            raise NotImplementedError("embedding_size != 1024 is not on the hot path")
        self.embedding_size = embedding_size
        self.fc1 = nn.Linear(belief_size + state_size, embedding_size)
        self.conv1 = nn.ConvTranspose2d(embedding_size, 128, 5, stride=2)
        self.conv2 = nn.ConvTranspose2d(128, 64, 5, stride=2)
        self.conv3 = nn.ConvTranspose2d(64, 32, 6, stride=2)
        self.conv4 = nn.ConvTranspose2d(32, 6, 6, stride=2)

    def plist(self):
        return [t for m in (self.fc1, self.conv1, self.conv2, self.conv3, self.conv4) for t in (m.weight, m.bias)]

    def forward(self, belief, state):
        from ..autograd import decoder_apply

        return decoder_apply(self, belief, state).chunk(2, 1)


def ObservationModel(symbolic, observation_size, belief_size, state_size, embedding_size, activation_function="relu"):
    if symbolic:
        raise NotImplementedError("symbolic (non-pixel) observations are outside the MI355X hot path")
    return VisualObservationModel(belief_size, state_size, embedding_size, activation_function,
                                  image_size=int(observation_size[-1]))


class _ScalarHead(nn.Module):
    """230 -> hidden^3 -> 1 ELU MLP on cat([belief, state])."""

    def __init__(self, belief_size, state_size, hidden_size, activation_function="relu"):
        super().__init__()
        if activation_function != "elu":
            raise NotImplementedError("HIP MLP kernels fuse ELU (dense_activation_function='elu')")
        self.fc1 = nn.Linear(belief_size + state_size, hidden_size)
        self.fc2 = nn.Linear(hidden_size, hidden_size)
        self.fc3 = nn.Linear(hidden_size, hidden_size)
        self.fc4 = nn.Linear(hidden_size, 1)

    def plist(self):
        return [t for m in (self.fc1, self.fc2, self.fc3, self.fc4) for t in (m.weight, m.bias)]

    def forward(self, belief, state):
        from ..autograd import mlp_apply

        return mlp_apply(self, belief, state).squeeze(dim=1)


class RewardModel(_ScalarHead):
    pass
