#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs through the same FusedTrainer path
(configs 4 and 5: wire2d 1024x1024, siren / gauss / relu / relu+posenc at 4x256, 512x512).
Prints one JSON line per config: samples/s fwd+bwd+Adam and the fraction of the fp32-MFMA
roofline on the SURVEY 8(d) algorithmic flop counts."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wire_amd.modules import models
from wire_amd.trainer import FusedTrainer

PEAK = 157.3e12
dev = torch.device("cuda:0")


def flops(kind, K, L, D, O):
    if kind == "wire":
        return 24 * K * K * L + 4 * D * K + 12 * K * O
    if kind == "wire2d":
        return 48 * K * K * L + 8 * D * K + 12 * K * O
    return 6 * K * K * L + 4 * D * K + 6 * K * O


def run(kind, side, hf, L=4, D=2, O=3, steps=8, **kw):
    torch.manual_seed(0)
    model = models.get_INR(nonlin=kind, in_features=D, out_features=O, hidden_features=hf, hidden_layers=L, **kw).to(dev)
    K = model._arch["width"]
    n = side * side
    target = torch.rand(n, O)
    tr = FusedTrainer(model, (side, side), target, lr=1e-3)
    for e in range(2):
        tr.step_hashed(e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(steps):
        loss = tr.step_hashed(2 + e)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    Din = D if not kw.get("pos_encode") else model.positional_encoding.out_dim
    F = flops(kind, K, L, Din, O)
    print(json.dumps({"config": f"{kind}{'+posenc' if kw.get('pos_encode') else ''} {L}x{hf} (K={K}) {side}x{side}",
                      "samples_per_s": n / dt, "ms_per_step": dt * 1e3, "alg_flop_per_sample": F,
                      "frac_of_fp32_mfma_peak": n / dt * F / PEAK, "loss": float(loss.item())}))
    del tr, model
    torch.cuda.empty_cache()


def run_occupancy(side, hf, L, batch, steps=8, **kw):
    """BASELINE config 3 shape (wire_occupancy.py): D = 3, O = 1, minibatches of `batch` points = consecutive slices
    of the epoch's shuffle of a side^3 volume (the position-keyed shuffle: no side^3 index vector), synthetic sphere
    occupancy.  One GPU's share of the 8-GPU job."""
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=3, out_features=1, hidden_features=hf, hidden_layers=L, **kw).to(dev)
    K = model._arch["width"]
    n = side ** 3
    ax = torch.linspace(-1, 1, side, device=dev)
    target = torch.empty(n, 1, device=dev)
    for i0 in range(0, side, 64):           # sphere indicator, built on the device slab by slab
        zz, yy, xx = torch.meshgrid(ax[i0:i0 + 64], ax, ax, indexing="ij")
        target[i0 * side * side:(i0 + zz.shape[0]) * side * side, 0] = ((xx * xx + yy * yy + zz * zz) < 0.5).float().reshape(-1)
    tr = FusedTrainer(model, (side, side, side), target, lr=1e-3, coords_style="numpy")
    for b in range(2):
        tr.step_hashed(0, first=b * batch, count=batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(steps):
        loss = tr.step_hashed(0, first=(2 + b) * batch, count=batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    F = flops("wire", K, L, 3, 1)
    print(json.dumps({"config": f"wire occupancy {L}x{hf} (K={K}) D=3 O=1, {side}^3 volume, batch {batch} (hashed shuffle)",
                      "samples_per_s": batch / dt, "ms_per_step": dt * 1e3, "alg_flop_per_sample": F,
                      "frac_of_fp32_mfma_peak": batch / dt * F / PEAK, "loss": float(loss.item())}))
    del tr, model
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run("wire", 512, 363, first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0)
    run("wire2d", 1024, 256, first_omega_0=10.0, hidden_omega_0=10.0, scale=10.0)
    run("siren", 512, 256, first_omega_0=30.0, hidden_omega_0=30.0)
    run("gauss", 512, 256, scale=10.0)
    run("relu", 512, 256)
    run("relu", 512, 256, pos_encode=True, sidelength=512)
    # config 3: occupancy, one GPU's share (256 k points per step) -- BASELINE width and the reference's own 3 x 300
    run_occupancy(512, 363, 4, 262144, first_omega_0=20.0, hidden_omega_0=20.0, scale=10.0)
    run_occupancy(512, 300, 3, 200000, first_omega_0=20.0, hidden_omega_0=20.0, scale=10.0)
