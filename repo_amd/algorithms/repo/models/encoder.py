"""Conv image encoder (reference: algorithms/repo/models/encoder.py:21-47)."""
import torch
import torch.nn as nn

from .... import functional as Fn


class VisualEncoder(nn.Module):
    """3x64x64 -> 32x31x31 -> 64x14x14 -> 128x6x6 -> 256x2x2 (k4, s2, ReLU), flattened to 1024
    (image_size=128: -> 256x6x6, then fc 9216 -> 1024).

    The nn.Conv2d children are parameter containers only (same constructors, hence the same
    default initialisation and state_dict names as the reference); arithmetic runs in the HIP
    implicit-GEMM kernels.  Accepts float32 frames in [-1,1] or raw uint8 frames."""

    def __init__(self, embedding_size, activation_function="relu", image_size=64):
        """image_size=128: the BUILD-DEFINED wider stack of BASELINE config 4 (the reference's encoder hard-codes the
        64 x 64 flatten, encoder.py:39): the same four convs (3x128x128 -> ... -> 256x6x6) and `fc` =
        Linear(9216, embedding_size), applied without an activation like the reference's optional fc (:40)."""
        super().__init__()
        if activation_function != "relu":
            raise NotImplementedError("HIP encoder kernels fuse ReLU (cnn_activation_function='relu')")
        if embedding_size != 1024:
            raise NotImplementedError("embedding_size != 1024 (extra fc layer) is not on the hot path")
        if image_size not in (64, 128):
            raise NotImplementedError(f"{image_size} x {image_size} frames: only 64 (the reference) and 128 (build-defined) are built")
        self.embedding_size = embedding_size
        self.image_size = image_size
        self.conv1 = nn.Conv2d(3, 32, 4, stride=2)
        self.conv2 = nn.Conv2d(32, 64, 4, stride=2)
        self.conv3 = nn.Conv2d(64, 128, 4, stride=2)
        self.conv4 = nn.Conv2d(128, 256, 4, stride=2)
        self.fc = nn.Identity() if image_size == 64 else nn.Linear(256 * 6 * 6, embedding_size)

    def plist(self):
        ps = [t for c in (self.conv1, self.conv2, self.conv3, self.conv4) for t in (c.weight, c.bias)]
        if self.image_size == 128:
            ps += [self.fc.weight, self.fc.bias]
        return ps

    def forward(self, observation):
        from ..autograd import encoder_apply

        return encoder_apply(self, observation)


def Encoder(symbolic, observation_size, embedding_size, activation_function="relu"):
    if symbolic:
        raise NotImplementedError("symbolic (non-pixel) observations are outside the MI355X hot path")
    return VisualEncoder(embedding_size, activation_function, image_size=int(observation_size[-1]))
