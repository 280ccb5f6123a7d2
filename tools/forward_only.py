#!/usr/bin/env python3
"""Forward-only dense query (FusedTrainer.render; modules/volutils.py:124-133, wire_multi_sr.py:215-217) of the headline net
over the 512 x 512 grid: samples/s.  Under rocprofv3 --kernel-trace --stats it shows the per-kernel split."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wire_amd.modules import models
from wire_amd.trainer import FusedTrainer

dev = torch.device("cuda:0")
torch.manual_seed(0)
hf = int(sys.argv[1]) if len(sys.argv) > 1 else 363
model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=hf, hidden_layers=4,
                       first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0).to(dev)
tr = FusedTrainer(model, (512, 512), torch.zeros(512 * 512, 3), lr=5e-3)
for _ in range(2):
    tr.render()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(5):
        tr.render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"render 512x512 (hidden_features={hf}): {dt * 1e3:.3f} ms  {512 * 512 / dt / 1e6:.1f} M samples/s")
