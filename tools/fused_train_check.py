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
for knob in (0, 1, 2):                                   # 0: layer by layer; 1: fused training forward; 2: + data-gradient chain
    _lib.check(L.wire_tune_set(b"fused_train", 1 if knob else 0))
    _lib.check(L.wire_tune_set(b"fused_bwd", 1 if knob == 2 else 0))
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire" if kind.startswith("wire") else kind, in_features=2, out_features=3, hidden_layers=4, **kw).to("cuda")
    N = side * side
    g = torch.Generator().manual_seed(1)
    tr = FusedTrainer(model, (side, side), torch.rand(N, 3, generator=g), lr=0.0, keep_rec=True)
    print("step with edition", knob, flush=True)
    loss = tr.step(torch.randperm(N, generator=g).to("cuda"))
    torch.cuda.synchronize()
    print("  done, loss", float(loss), flush=True)
    res[knob] = (float(loss), tr.flat_grad.clone(), tr.rec.clone())
_lib.check(L.wire_tune_set(b"fused_train", 1))
_lib.check(L.wire_tune_set(b"fused_bwd", 1))
g0 = res[0][1]
for k in (1, 2):
    print(f"edition {k} vs layer by layer: loss", res[0][0], res[k][0],
          " grad max |diff| / max |g|:", float((g0 - res[k][1]).abs().max() / g0.abs().max()),
          " rec:", float((res[0][2] - res[k][2]).abs().max() / res[0][2].abs().max()))
print("edition 2 vs edition 1: grad max |diff| / max |g|:", float((res[1][1] - res[2][1]).abs().max() / g0.abs().max()),
      " elements that differ:", int((res[1][1] != res[2][1]).sum()), "of", g0.numel())
