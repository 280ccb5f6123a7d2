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
    for _ in range(2):
        tr.step(torch.randperm(n, device=dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(torch.randperm(n, device=dev))
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
    """BASELINE config 3 shape (wire_occupancy.py): D = 3, O = 1, random index minibatches of `batch` points out
    of a side^3 volume, synthetic sphere occupancy."""
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=3, out_features=1, hidden_features=hf, hidden_layers=L, **kw).to(dev)
    K = model._arch["width"]
    n = side ** 3
    ax = torch.linspace(-1, 1, side)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    target = ((xx * xx + yy * yy + zz * zz) < 0.5).float().reshape(-1, 1)
    tr = FusedTrainer(model, (side, side, side), target, lr=1e-3, coords_style="numpy")
    mk = lambda: torch.randint(0, n, (batch,), device=dev, dtype=torch.int64)
    for _ in range(2):
        tr.step(mk())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(mk())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    F = flops("wire", K, L, 3, 1)
    print(json.dumps({"config": f"wire occupancy {L}x{hf} (K={K}) D=3 O=1, {side}^3 volume, batch {batch}",
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
    run_occupancy(256, 363, 4, 262144, first_omega_0=20.0, hidden_omega_0=20.0, scale=10.0)
    run_occupancy(256, 300, 3, 200000, first_omega_0=20.0, hidden_omega_0=20.0, scale=10.0)
