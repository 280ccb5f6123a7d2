#!/usr/bin/env python3
"""The reference's training loop unchanged (wire_image_denoise.py:141-178) through the drop-in modules: epoch time, and the
same loop with the reference's host work taken out step by step (what of the gap to FusedTrainer is the loop's own)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

dev = torch.device("cuda:0")
wire_kw = dict(first_omega_0=bench.OMEGA0, hidden_omega_0=bench.OMEGA0, scale=bench.SIGMA0)
r = bench.reference_loop(dev, wire_kw, epochs=8)
print(f"loop unchanged:                      {r['ms_per_epoch']:.3f} ms / epoch  {r['samples_per_s'] / 1e6:.2f} M samples/s")

r = bench.reference_loop(dev, wire_kw, epochs=8, device_resident=True)
print(f"same calls, everything on the device: {r['ms_per_epoch']:.3f} ms / epoch  {r['samples_per_s'] / 1e6:.2f} M samples/s")

from wire_amd.modules import models
H = W = bench.SIDE
torch.manual_seed(0)
model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=bench.HIDDEN_FEATURES, hidden_layers=4,
                       **wire_kw).cuda()
x = torch.linspace(-1, 1, W); y = torch.linspace(-1, 1, H)
X, Y = torch.meshgrid(x, y, indexing="xy")
coords = torch.hstack((X.reshape(-1, 1), Y.reshape(-1, 1)))[None, ...].cuda()
gt = torch.rand(1, H * W, 3, device="cuda")
optim = torch.optim.Adam(lr=5e-3, params=model.parameters())
indices = torch.randperm(H * W, device="cuda")
b_coords = coords[:, indices, ...]
# forward + backward only
for epoch in range(10):
    if epoch == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    pix = model(b_coords)
    loss = ((pix - gt[:, indices, :]) ** 2).mean()
    optim.zero_grad()
    loss.backward()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 8
print(f"model(b_coords) + loss + backward only: {dt * 1e3:.3f} ms")
