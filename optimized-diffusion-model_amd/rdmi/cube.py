"""Unit-hypercube helpers: reflect, inside, sample_hk, score_hk.

Host mirror of Reflected-Diffusion/cube.py ("RD/cube.py"); reflect and score_hk run as HIP kernels
(rdmi_reflect / rdmi_score_hk), the one-line helpers are tensor plumbing.
"""
import torch

from . import _native


def unsqueeze_as(x, y, back=True):
    """RD/cube.py:5-14."""
    extra = (1,) * (len(y.shape) - len(x.shape))
    return x.view(*x.shape, *extra) if back else x.view(*extra, *x.shape)


def inside(x):
    """RD/cube.py:17-31: per-sample 'all coordinates in [0, 1]'."""
    x = x.flatten(1)
    return torch.logical_and(x >= 0, x <= 1).all(dim=-1)


def reflect(x):
    """RD/cube.py:34-49: fold x into [0, 1] by reflections (floor-mod 2, then 2 - m above 1)."""
    return _native.reflect(x)


def sample_hk(x, sigma):
    """RD/cube.py:52-70: reflect(x + sigma * N(0, I))."""
    if not torch.is_tensor(sigma):
        sigma = sigma * torch.ones(x.shape[0]).to(x)
    return reflect(torch.randn_like(x) * unsqueeze_as(sigma, x) + x)


def score_hk(x, x_orig, sigma, efs=20, refls=10, min_cutoff=1e-2):
    """RD/cube.py:149-193: score of the reflected heat kernel; eigenfunction series for
    t = sigma^2/2 > min_cutoff, image sum otherwise (chosen per sample, in-kernel)."""
    if not torch.is_tensor(sigma):
        sigma = sigma * torch.ones(x.shape[0]).to(x)
    return _native.score_hk(x, x_orig, sigma, efs, refls, min_cutoff)
