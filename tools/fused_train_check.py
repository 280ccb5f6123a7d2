#!/usr/bin/env python3
"""Debug aid: one FusedTrainer step with the fused training forward (knob "fused_train" = 1) against the layer-by-layer
forward (= 0) on the same weights -- the stored lin_l / out_l of the act buffer, loss and flat gradient."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wire_amd import _lib
from wire_amd.modules import models
from wire_amd.trainer import FusedTrainer

kind = sys.argv[1] if len(sys.argv) > 1 else "siren"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 96
kw = {"siren": dict(hidden_features=256, first_omega_0=30.0, hidden_omega_0=30.0), "gauss": dict(hidden_features=256, scale=10.0),
      "relu": dict(hidden_features=256), "wire": dict(hidden_features=182, first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0),
      "wire90": dict(hidden_features=128, first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0)}[kind]
L = _lib.lib()
res = {}
for knob in (0, 1):
    _lib.check(L.wire_tune_set(b"fused_train", knob))
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire" if kind.startswith("wire") else kind, in_features=2, out_features=3, hidden_layers=4, **kw).to("cuda")
    N = side * side
    g = torch.Generator().manual_seed(1)
    tr = FusedTrainer(model, (side, side), torch.rand(N, 3, generator=g), lr=0.0, keep_rec=True)
    print("step with fused_train =", knob, flush=True)
    loss = tr.step(torch.randperm(N, generator=g).to("cuda"))
    torch.cuda.synchronize()
    print("  done, loss", float(loss), flush=True)
    res[knob] = (float(loss), tr.flat_grad.clone(), tr.rec.clone())
_lib.check(L.wire_tune_set(b"fused_train", 1))
g0, g1 = res[0][1], res[1][1]
print("loss", res[0][0], res[1][0])
print("grad max |diff| / max |g|:", float((g0 - g1).abs().max() / g0.abs().max()))
print("rec  max |diff| / max |y|:", float((res[0][2] - res[1][2]).abs().max() / res[0][2].abs().max()))
