#!/usr/bin/env python3
"""Interleaved A/B of one tuning knob on the headline training step (and the forward-only render) in ONE process on one
box: blocks of timed steps alternate between the two settings, so clock / temperature drift hits both alike.
    python3 tools/knob_ab.py split_out [hidden_features] [nonlin] [values, e.g. 2,1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wire_amd import _lib
from wire_amd.modules import models
from wire_amd.trainer import FusedTrainer

knob = sys.argv[1].encode() if len(sys.argv) > 1 else b"split_out"
hf = int(sys.argv[2]) if len(sys.argv) > 2 else 363
nonlin = sys.argv[3] if len(sys.argv) > 3 else "wire"
vals = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (1, 0)
dev = torch.device("cuda:0")
torch.manual_seed(0)
kw = dict(first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0) if nonlin == "wire" else \
    dict(first_omega_0=30.0, hidden_omega_0=30.0, scale=10.0)
model = models.get_INR(nonlin=nonlin, in_features=2, out_features=3, hidden_features=hf, hidden_layers=4, **kw).to(dev)
tr = FusedTrainer(model, (512, 512), torch.rand(512 * 512, 3), lr=5e-3, niters=2000)
L = _lib.lib()
tot = {v: [] for v in vals}
rnd = {v: [] for v in vals}
default = L.wire_tune_get(knob)
for rep in range(6):
    for v in vals:
        _lib.check(L.wire_tune_set(knob, v))
        for i in range(3):
            tr.step_hashed(rep * 100 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(15):
            tr.step_hashed(rep * 100 + 10 + i)
        torch.cuda.synchronize()
        tot[v].append((time.perf_counter() - t0) / 15 * 1e3)
        for _ in range(2):
            tr.render()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tr.render()
        torch.cuda.synchronize()
        rnd[v].append((time.perf_counter() - t0) / 5 * 1e3)
_lib.check(L.wire_tune_set(knob, default))
for v in vals:
    print(f"{knob.decode()} = {v} ({nonlin}, hidden_features {hf}): step mean {sum(tot[v]) / len(tot[v]):.3f} ms  min {min(tot[v]):.3f} ms"
          f"   |  render mean {sum(rnd[v]) / len(rnd[v]):.3f} ms  min {min(rnd[v]):.3f} ms   "
          f"[{' '.join(f'{x:.3f}' for x in tot[v])}]")
