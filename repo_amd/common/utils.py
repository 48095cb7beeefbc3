"""Device selection and host<->device helpers.

Mirrors the surface of the reference's common/utils.py (set_gpu_mode :10-19, get_device
:22-24, to_torch :27-30, to_np :33-34, FreezeParameters :47-58, lambda_return :61-71,
preprocess :74-80, postprocess :83-89) so scripts written against it keep working.  On a
ROCm build of PyTorch the "cuda:<id>" device string IS the HIP device.
"""
import numpy as np
import torch

_GPU_ID = 0
_USE_GPU = False
_DEVICE = None


def set_gpu_mode(mode, gpu_id=0):
    """Select the device.  Unlike the reference this does not change torch's default tensor
    type (deprecated global state); every tensor the package creates names its device."""
    global _GPU_ID, _USE_GPU, _DEVICE
    _GPU_ID = gpu_id
    _USE_GPU = bool(mode)
    _DEVICE = torch.device(("cuda:" + str(_GPU_ID)) if _USE_GPU else "cpu")
    if _USE_GPU:
        torch.cuda.set_device(_DEVICE)


def get_device():
    global _DEVICE
    if _DEVICE is None:
        set_gpu_mode(torch.cuda.is_available())
    return _DEVICE


def to_torch(x, dtype=None, device=None):
    if device is None:
        device = get_device()
    return torch.as_tensor(x, dtype=dtype, device=device)


def to_np(x):
    return x.detach().cpu().numpy()


class FreezeParameters:
    """Context manager turning requires_grad off for a parameter list (common/utils.py:47-58)."""

    def __init__(self, params):
        self.params = list(params)
        self.param_states = [p.requires_grad for p in self.params]

    def __enter__(self):
        for p in self.params:
            p.requires_grad = False

    def __exit__(self, exc_type, exc_val, exc_tb):
        for p, s in zip(self.params, self.param_states):
            p.requires_grad = s


def lambda_return(rewards, values, discounts, bootstrap, lambda_=0.95):
    """TD(lambda) returns (common/utils.py:61-71) on the HIP kernel.  The kernel takes a
    constant discount (dreamer.py:342 builds gamma*ones); a non-constant tensor raises."""
    from .. import ops

    gamma = float(discounts.reshape(-1)[0])
    if not bool((discounts == gamma).all()):
        raise NotImplementedError("lambda_return kernel supports a constant discount only")
    r = torch.cat([rewards, torch.zeros_like(bootstrap)[None]], 0).contiguous()
    v = torch.cat([values, bootstrap[None]], 0).contiguous()
    returns, _, _, _ = ops.lambda_return(r, v, gamma, lambda_, want_grads=False)
    return returns


def preprocess(obs):
    """uint8 pixels -> float32 in [-1, 1] on the host (common/utils.py:74-80).  The training
    path does NOT use this: it ships uint8 frames to the device and normalises inside the
    first convolution's loader; this is kept for the acting path and for API parity."""
    ndims = len(obs.shape)
    assert ndims == 2 or ndims == 4, "preprocess accepts a batch of observations"
    if ndims == 4:
        obs = ((obs.astype(np.float32) / 255) * 2) - 1.0
    return obs


def postprocess(obs):
    ndims = len(obs.shape)
    assert ndims == 2 or ndims == 4, "postprocess accepts a batch of observations"
    if ndims == 4:
        obs = np.floor((obs + 1.0) / 2 * 255).clip(0, 255).astype(np.uint8)
    return obs
