#!/usr/bin/env python3
"""Timing probes of the whole-net forward kernel (wire_fused.hip; results wrong, harness build only):

    make -C wire_amd/csrc clean && make -C wire_amd/csrc -j4 EXTRA=-DWIRE_FX_ABLATE      # then, on the GPU box:
    python3 tools/fused_ablate.py

siren 4 x 256 on 512 x 512, forward-only render: the kernel as shipped (0) against editions without the producer's vector
work (1), without the weight-fragment LDS reads (2), without the weight stream (4), without the stage barrier (8) and
combinations -- interleaved rounds in one process, best of three."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wire_amd import _lib
from wire_amd.modules import models
from wire_amd.trainer import FusedTrainer

NAMES = {0: "as shipped", 1: "no producer (vector work)", 2: "no fragment reads", 3: "neither (MFMAs + stream + barrier)",
         4: "no weight stream", 8: "no barrier", 12: "no stream, no barrier", 15: "MFMAs only"}
L = _lib.lib()
if L.wire_tune_set(b"fx_ablate", 0) != 0:
    sys.exit("this library was not built with EXTRA=-DWIRE_FX_ABLATE")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = models.get_INR(nonlin="siren", in_features=2, out_features=3, hidden_features=256, hidden_layers=4,
                       first_omega_0=30.0, hidden_omega_0=30.0).to(dev)
tr = FusedTrainer(model, (512, 512), torch.zeros(512 * 512, 3), lr=5e-3)
best = {k: 1e9 for k in NAMES}
for rnd in range(3):
    for k in NAMES:
        _lib.check(L.wire_tune_set(b"fx_ablate", k))
        tr.render()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.render()
        torch.cuda.synchronize()
        best[k] = min(best[k], (time.perf_counter() - t0) / 10)
_lib.check(L.wire_tune_set(b"fx_ablate", 0))
for k, v in best.items():
    print(f"ablate {k:2d}  {NAMES[k]:40s} {v * 1e3:7.3f} ms per render (incl. ~0.03 ms of packing / coordinate launches)")
