#!/usr/bin/env python3
"""Timing probes of the data-gradient chain (wire_fused.hip: fused_bwd_kernel; results wrong, harness build only):

    make -C wire_amd/csrc clean && make -C wire_amd/csrc -j4 EXTRA=-DWIRE_FX_ABLATE      # then, on the GPU box:
    python3 tools/fused_bwd_ablate.py

siren 4 x 256 on 512 x 512, the training step with the chain as shipped (0) against editions without its lin_{l-1} loads (1),
without its g_lin stores (2), without the activation derivative (4), without its MFMAs (8), without the weight stream's waits
and barriers (16), without the weight-fragment LDS reads (32), without the splits (64) and combinations -- interleaved rounds in one process; the DIFFERENCE of two step times is the difference
of the chain kernel (everything else in the step is the same work on different numbers)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wire_amd import _lib
from wire_amd.modules import models
from wire_amd.trainer import FusedTrainer

NAMES = {0: "as shipped", 1: "no lin loads", 2: "no g_lin stores", 3: "no loads, no stores", 4: "no activation derivative",
         7: "no loads, stores, derivative", 8: "no MFMAs", 16: "no weight waits / barriers", 23: "1 + 2 + 4 + 16",
         32: "no weight-fragment LDS reads", 64: "no splits", 55: "23 + no fragment reads", 119: "55 + no splits",
         127: "everything off"}
L = _lib.lib()
if L.wire_tune_set(b"fxb_ablate", 0) != 0:
    sys.exit("this library was not built with EXTRA=-DWIRE_FX_ABLATE")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = models.get_INR(nonlin="siren", in_features=2, out_features=3, hidden_features=256, hidden_layers=4,
                       first_omega_0=30.0, hidden_omega_0=30.0).to(dev)
tr = FusedTrainer(model, (512, 512), torch.rand(512 * 512, 3), lr=0.0)
best = {k: 1e9 for k in NAMES}
for rnd in range(3):
    for k in NAMES:
        _lib.check(L.wire_tune_set(b"fxb_ablate", k))
        for i in range(3):
            tr.step_hashed(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(15):
            tr.step_hashed(10 + i)
        torch.cuda.synchronize()
        best[k] = min(best[k], (time.perf_counter() - t0) / 15)
_lib.check(L.wire_tune_set(b"fxb_ablate", 0))
for k, v in best.items():
    print(f"ablate {k:2d}  {NAMES[k]:34s} step {v * 1e3:7.3f} ms   chain {(v - best[0]) * 1e3:+7.3f} ms")
