"""Harness helpers on the hot path's contract (reference modules/utils.py).

Only the functions the training/eval loop of wire_image_denoise.py and
wire_occupancy.py touch are provided: coordinate grids, PSNR, parameter count,
normalisation, the noise model and logging.  Plotting / montage / table helpers
are out of scope -- except ``get_layer_outputs`` + ``build_montage``, the per-layer
visualisation query SURVEY section 8 (f)3 names.
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


def build_montage(images):
    """Tile ``images`` [n, H, W] (each min-max normalised, ``normalize(.., True)``) row by row into a
    ceil(sqrt(n))-row grid; unused cells stay 0 (modules/utils.py:131-156)."""
    images = np.asarray(images)
    n, H, W = images.shape
    nrows = int(np.ceil(np.sqrt(n)))
    ncols = int(np.ceil(n / nrows))
    out = np.zeros((H * nrows, W * ncols), dtype=np.float32)
    for k in range(n):
        r, c = divmod(k, ncols)
        out[r * H:(r + 1) * H, c * W:(c + 1) * W] = normalize(images[k], True)
    return out


@torch.no_grad()
def get_layer_outputs(model, coords, imsize, nfilters_vis=16, get_imag=False):
    """Activation images after each layer, for visualisation (modules/utils.py:229-288).  Every layer runs through its
    own HIP entry point -- ``model.net[idx](x)`` -> wire_gabor_fwd / wire_real_layer_fwd / wire_gabor2d_fwd, the kernels
    of the fused path -- on the previous layer's (complex) output; the rest is the reference's host-side bookkeeping:
    the first ``nfilters_vis`` filters ('all': every filter), real or imaginary part, sign chosen so that the larger
    excursion is positive, filters ordered by standard deviation, each min-max normalised with a white frame, tiled by
    ``build_montage``.  Returns one montage per activation layer."""
    H, W = imsize
    if getattr(model, "pos_encode", False):
        coords = model.positional_encoding(coords)
    montages = []
    x = coords
    for idx in range(len(model.net) - 1):
        x = model.net[idx](x)
        imgs = x.reshape(1, H, W, -1)[0]
        if nfilters_vis != 'all':
            imgs = imgs[..., :nfilters_vis]
        atoms = imgs.detach().cpu().numpy()
        atoms = atoms.imag if get_imag else atoms.real
        lo = atoms.min(0, keepdims=True).min(1, keepdims=True)
        hi = atoms.max(0, keepdims=True).max(1, keepdims=True)
        atoms = (1 - 2 * (abs(lo) > abs(hi))) * atoms
        atoms = atoms[..., np.argsort(atoms.std((0, 1)))]
        lo = atoms.min(0, keepdims=True).min(1, keepdims=True)
        hi = atoms.max(0, keepdims=True).max(1, keepdims=True)
        atoms = (atoms - lo) / np.maximum(1e-14, hi - lo)
        atoms[:, [0, -1], :] = 1
        atoms[[0, -1], :, :] = 1
        montages.append(build_montage(np.transpose(atoms, [2, 0, 1])))
    return montages
