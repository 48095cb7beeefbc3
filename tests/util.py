import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "parity_log.txt")


def log(msg):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(msg + "\n")
    print(msg)


def relerr(got, want):
    """max |got-want| / max|want| (normwise, robust to near-zero entries)."""
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    scale = want.abs().max().item() + 1e-30
    return (got - want).abs().max().item() / scale


def l2err(got, want):
    got = got.detach().double().cpu().flatten()
    want = want.detach().double().cpu().flatten()
    return ((got - want).norm() / (want.norm() + 1e-30)).item()


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))
