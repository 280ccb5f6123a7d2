#!/usr/bin/env python3
"""Longer quality check than the unit test: fit a 128x128 synthetic RGB image with the BASELINE
architecture (wire, hidden_features=256 -> K=181, 4 hidden layers, omega0=20, sigma0=30 -- the
representation regime of wire_image_denoise.py:40) for --steps full-batch Adam steps with the
denoise script's LambdaLR, on the MI355X path and on the CPU restatement (oracle/torch_ref.py),
same seed, same data.  Prints both PSNR trajectories (utils.psnr: max(x)/mse)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--side", type=int, default=128)
ap.add_argument("--control64", action="store_true",
                help="also run the CPU restatement in fp64: the fp32-vs-fp64 gap of the reference's own "
                     "arithmetic is the yardstick for the HIP-vs-CPU gap (trajectories are chaotic)")
args = ap.parse_args()

from oracle import torch_ref, wire_oracle as wo
from wire_amd.modules import models, utils
from wire_amd.trainer import FusedTrainer

H = W = args.side
yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
img = np.stack([0.5 + 0.35 * np.sin(9 * xx + 4 * yy) * np.cos(5 * yy),
                0.5 + 0.4 * np.cos(7 * xx * yy + 1.0),
                0.5 + 0.3 * np.sin(11 * yy) * np.cos(3 * xx) + 0.1 * np.sign(np.sin(6 * xx))], -1).astype(np.float32)
target = torch.tensor(img.reshape(-1, 3))
L, hf, om, sc, lr = 4, 256, 20.0, 30.0, 5e-3
torch.manual_seed(0)
model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=hf, hidden_layers=L,
                       first_omega_0=om, hidden_omega_0=om, scale=sc)
p_cpu = {k: v.detach().clone() for k, v in model.state_dict().items() if "omega" not in k and "scale_0" not in k}
model = model.to("cuda")
tr = FusedTrainer(model, (H, W), target, lr=lr, niters=args.steps)
marks = sorted(set([args.steps // 8, args.steps // 4, args.steps // 2, args.steps]))
hip = {}
t0 = time.time()
for it in range(1, args.steps + 1):
    tr.step()
    tr.scheduler_step()
    if it in marks:
        hip[it] = float(tr.psnr(tr.render()).item())
torch.cuda.synchronize()
t_hip = time.time() - t0

coords = torch.tensor(wo.image_coords(H, W))[None]
params = {k: v.clone().requires_grad_(True) for k, v in p_cpu.items()}
opt = torch.optim.Adam(lr=lr, params=list(params.values()))
sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: 0.1 ** min(x / args.steps, 1))
ref = {}
t0 = time.time()
for it in range(1, args.steps + 1):
    y = torch_ref.wire_forward(params, coords, L, om, om, sc)
    loss = ((y - target[None]) ** 2).mean()
    opt.zero_grad(); loss.backward(); opt.step(); sched.step()
    if it in marks:
        with torch.no_grad():
            rec = torch_ref.wire_forward(params, coords, L, om, om, sc)[0].numpy()
        ref[it] = float(utils.psnr(img.reshape(-1, 3), rec))
        print(f"step {it}: HIP {hip[it]:.3f} dB  CPU {ref[it]:.3f} dB  diff {hip[it] - ref[it]:+.3f}", flush=True)
t_cpu = time.time() - t0
ref64 = {}
if args.control64:
    p64 = {k: (v.to(torch.cdouble) if v.is_complex() else v.double()).clone().requires_grad_(True) for k, v in p_cpu.items()}
    opt = torch.optim.Adam(lr=lr, params=list(p64.values()))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: 0.1 ** min(x / args.steps, 1))
    c64, t64 = coords.double(), target[None].double()
    for it in range(1, args.steps + 1):
        y = torch_ref.wire_forward(p64, c64, L, om, om, sc)
        loss = ((y - t64) ** 2).mean()
        opt.zero_grad(); loss.backward(); opt.step(); sched.step()
        if it in marks:
            with torch.no_grad():
                rec = torch_ref.wire_forward(p64, c64, L, om, om, sc)[0].numpy()
            ref64[it] = float(utils.psnr(img.reshape(-1, 3).astype(np.float64), rec))
            print(f"step {it}: CPU fp64 {ref64[it]:.3f} dB  (CPU fp32 {ref[it]:.3f}, HIP {hip[it]:.3f})", flush=True)
print(json.dumps({"psnr_cpu_fp64": ref64}))
print(json.dumps({"image": f"{H}x{W}", "net": f"wire 4x{hf} (K=181)", "steps": args.steps, "psnr_hip": hip,
                  "psnr_cpu": ref, "seconds_hip": t_hip, "seconds_cpu": t_cpu}))
