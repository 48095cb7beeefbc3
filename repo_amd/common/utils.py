"""Device selection and host<->device helpers with the call surface scripts written against the
reference expect (reference common/utils.py: set_gpu_mode :10-19, get_device :22-24, to_torch :27-30,
to_np :33-34, FreezeParameters :47-58, lambda_return :61-71, preprocess :74-80, postprocess :83-89).
On a ROCm build of PyTorch the "cuda:<id>" device string IS the HIP device.
"""
import contextlib

import numpy as np
import torch


class _Selected:
    """Process-wide device choice.  Unlike the reference, selecting a device does not touch torch's
    default tensor type (deprecated global state): every tensor this package creates names its device."""

    device = None

    @classmethod
    def choose(cls, use_gpu, index):
        cls.device = torch.device("cuda", int(index)) if use_gpu else torch.device("cpu")
        if cls.device.type == "cuda":
            torch.cuda.set_device(cls.device)
        return cls.device


def set_gpu_mode(mode, gpu_id=0):
    _Selected.choose(bool(mode), gpu_id)


def get_device():
    return _Selected.device or _Selected.choose(torch.cuda.is_available(), 0)


def to_torch(x, dtype=None, device=None):
    return torch.as_tensor(x, dtype=dtype, device=get_device() if device is None else device)


def to_np(x):
    return x.detach().cpu().numpy()


class FreezeParameters(contextlib.AbstractContextManager):
    """`with FreezeParameters(params):` -- requires_grad off inside, restored (per parameter) on exit."""

    def __init__(self, params):
        self._saved = [(p, p.requires_grad) for p in params]

    def __enter__(self):
        for p, _ in self._saved:
            p.requires_grad_(False)
        return self

    def __exit__(self, *exc):
        for p, flag in self._saved:
            p.requires_grad_(flag)
        return False


def lambda_return(rewards, values, discounts, bootstrap, lambda_=0.95):
    """TD(lambda) returns on the HIP kernel (repo_lambda_return).  The kernel takes ONE discount
    (dreamer.py:342 builds gamma*ones); a tensor that is not constant raises."""
    from .. import ops

    gamma = float(discounts.reshape(-1)[0])
    if not bool((discounts == gamma).all()):
        raise NotImplementedError("lambda_return kernel supports a constant discount only")
    pad = torch.zeros_like(bootstrap).unsqueeze(0)
    out = ops.lambda_return(torch.cat([rewards, pad]).contiguous(), torch.cat([values, bootstrap.unsqueeze(0)]).contiguous(),
                            gamma, lambda_, want_grads=False)
    return out[0]


def _is_image_batch(a):
    if a.ndim not in (2, 4):
        raise AssertionError("expected a batch: (n, features) or (n, C, H, W)")
    return a.ndim == 4


def preprocess(obs):
    """uint8 frames -> float32 in [-1, 1] on the HOST, rounding exactly like x/255*2-1.  The training path
    does not use this (it ships uint8 frames and normalises inside the first convolution's loader); the
    acting path and API parity do.  Feature batches pass through."""
    if not _is_image_batch(obs):
        return obs
    x = obs.astype(np.float32)
    x /= 255
    x *= 2
    x -= 1.0
    return x


def postprocess(obs):
    """[-1, 1] frames -> uint8 (floor, clipped); feature batches pass through."""
    if not _is_image_batch(obs):
        return obs
    return np.clip(np.floor((obs + 1.0) / 2 * 255), 0, 255).astype(np.uint8)


class Video:
    """(frames, fps) record handed to `logger.record("test/video", ...)` when the host application
    has no `common.logger.Video` of its own (reference common/logger.py:26-35)."""

    def __init__(self, frames, fps):
        self.frames = frames
        self.fps = fps
