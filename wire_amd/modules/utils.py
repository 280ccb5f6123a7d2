"""Harness helpers on the hot path's contract (reference modules/utils.py).

Only the functions the training/eval loop of wire_image_denoise.py and
wire_occupancy.py touch are provided: coordinate grids, PSNR, parameter count,
normalisation, the noise model and logging.  Plotting / montage / table helpers
are out of scope.
"""
from datetime import datetime

import numpy as np
import torch


def log(msg):                                   # modules/utils.py:291-292
    print(f"{datetime.now()} - {msg}")


def normalize(x, fullnormalize=False):          # modules/utils.py:21-38
    if x.sum() == 0:
        return x
    xmax = x.max()
    xmin = x.min() if fullnormalize else 0
    return (x - xmin) / (xmax - xmin)


def psnr(x, xhat):
    """10 log10(max(x) / mse) -- max(x), not max(x)^2 (modules/utils.py:67-82)."""
    err = np.asarray(x) - np.asarray(xhat)
    return 10 * np.log10(np.max(x) / np.mean(err ** 2))


def measure(x, noise_snr=40, tau=100):
    """Photon + readout noise, same np.random call order as modules/utils.py:85-112."""
    meas = np.copy(x)
    noise = np.random.randn(meas.size).reshape(meas.shape) * noise_snr
    if tau != float('Inf'):
        meas = meas * tau
        pos = x > 0
        meas[pos] = np.random.poisson(meas[pos])
        meas[~pos] = -np.random.poisson(-meas[~pos])
        return (meas + noise) / tau
    return meas + noise


def count_parameters(model):                    # modules/utils.py:159-160
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def get_coords(H, W, T=None):
    """2-D / 3-D grids in [-1, 1] (modules/utils.py:163-176): np.linspace in
    fp64, 'xy' meshgrid, flattened row-major, cast to fp32."""
    axes = [np.linspace(-1, 1, W), np.linspace(-1, 1, H)]
    if T is not None:
        axes.append(np.linspace(-1, 1, T))
    grids = np.meshgrid(*axes)
    coords = np.hstack([g.reshape(-1, 1) for g in grids])
    return torch.tensor(coords.astype(np.float32))


def axis_tables(H, W, T=None, style="numpy"):
    """Per-axis coordinate tables for the on-device generator
    (wire_coords_from_index).  style='numpy' reproduces get_coords above;
    style='torch' reproduces torch.linspace(-1, 1, n) of
    wire_image_denoise.py:63-64 bit for bit."""
    def ax(n):
        if style == "torch":
            return torch.linspace(-1, 1, n)
        return torch.tensor(np.linspace(-1, 1, n).astype(np.float32))
    return ax(W), ax(H), (ax(T) if T is not None else None)
