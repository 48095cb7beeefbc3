"""Conv image encoder (reference: algorithms/repo/models/encoder.py:21-47)."""
import torch
import torch.nn as nn

from .... import functional as Fn


class VisualEncoder(nn.Module):
    """3x64x64 -> 32x31x31 -> 64x14x14 -> 128x6x6 -> 256x2x2 (k4, s2, ReLU), flattened to 1024.

    The nn.Conv2d children are parameter containers only (same constructors, hence the same
    default initialisation and state_dict names as the reference); arithmetic runs in the HIP
    implicit-GEMM kernels.  Accepts float32 frames in [-1,1] or raw uint8 frames."""

    def __init__(self, embedding_size, activation_function="relu"):
        super().__init__()
        if activation_function != "relu":
            raise NotImplementedError("HIP encoder kernels fuse ReLU (cnn_activation_function='relu')")
        if embedding_size != 1024:
            raise NotImplementedError("embedding_size != 1024 (extra fc layer) is not on the hot path")
        self.embedding_size = embedding_size
        self.conv1 = nn.Conv2d(3, 32, 4, stride=2)
        self.conv2 = nn.Conv2d(32, 64, 4, stride=2)
        self.conv3 = nn.Conv2d(64, 128, 4, stride=2)
        self.conv4 = nn.Conv2d(128, 256, 4, stride=2)
        self.fc = nn.Identity()

    def plist(self):
        return [t for c in (self.conv1, self.conv2, self.conv3, self.conv4) for t in (c.weight, c.bias)]

    def forward(self, observation):
        from ..autograd import encoder_apply

        return encoder_apply(self, observation)


def Encoder(symbolic, observation_size, embedding_size, activation_function="relu"):
    if symbolic:
        raise NotImplementedError("symbolic (non-pixel) observations are outside the MI355X hot path")
    return VisualEncoder(embedding_size, activation_function)
