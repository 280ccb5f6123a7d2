#!/usr/bin/env python3
"""Golden vector for ``utils.get_layer_outputs`` (modules/utils.py:229-288; build container only):

    python3 tests/golden/make_layer_outputs_golden.py        # writes tests/golden/layer_outputs.npz

The reference's own function on the reference's own ``wire.INR`` (2 hidden layers x 64 features -> K = 45, omega0 = 5,
sigma0 = 5, torch.manual_seed(0)) over a 20 x 24 grid, real and imaginary montages of the first 9 filters per layer.
Only data is written (coordinates, montages, state_dict checksums).  ``cv2`` is stubbed as in make_golden.py."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, REF)
from modules import utils, wire  # noqa: E402


def checksum(a):
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = np.stack([a.real, a.imag], -1)
    a = a.astype(np.float64).ravel()
    w = np.cos(np.arange(a.size) * 0.37) + 0.5
    return np.array([a.sum(), np.abs(a).sum(), (a * w).sum()], np.float64)


H, W = 20, 24
torch.manual_seed(0)
model = wire.INR(2, 64, 0, 2, 3, True, 5.0, 5.0, 5.0)
coords = utils.get_coords(H, W)[None].float()
out = {"H": np.int64(H), "W": np.int64(W), "coords": coords.numpy()}
for tag, imag in (("re", False), ("im", True)):
    for i, m in enumerate(utils.get_layer_outputs(model, coords, (H, W), nfilters_vis=9, get_imag=imag)):
        out[f"montage_{tag}_{i}"] = np.asarray(m, np.float32)
for k, v in model.state_dict().items():
    if "omega_0" not in k and "scale_0" not in k:
        out["sd_checksum__" + k] = checksum(v.numpy())
np.savez_compressed(os.path.join(OUT, "layer_outputs.npz"), **out)
print("wrote layer_outputs.npz:", sorted(k for k in out if k.startswith("montage")))
