#!/usr/bin/env python3
"""Forward-only dense query (FusedTrainer.render; modules/volutils.py:124-133, wire_multi_sr.py:215-217) over the 512 x 512
grid: samples/s of every net of BASELINE.json's sweep, with the whole-net kernel (wire_fused.hip, knob "fused_fwd" = 1) and
layer by layer ("fused_fwd" = 0) in alternating rounds of one process.  Under rocprofv3 --kernel-trace --stats it shows
the per-kernel split.

    python3 tools/forward_only.py [name ...]        names: wire_k256 wire_k181 siren gauss relu wire_k128 wire_k90
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wire_amd import _lib
from wire_amd.modules import models
from wire_amd.trainer import FusedTrainer

NETS = {
    "wire_k256": dict(nonlin="wire", hidden_features=363, first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0),
    "wire_k181": dict(nonlin="wire", hidden_features=256, first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0),
    "wire_k128": dict(nonlin="wire", hidden_features=182, first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0),
    "wire_k90": dict(nonlin="wire", hidden_features=128, first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0),
    "siren": dict(nonlin="siren", hidden_features=256, first_omega_0=30.0, hidden_omega_0=30.0),
    "gauss": dict(nonlin="gauss", hidden_features=256, scale=10.0),
    "relu": dict(nonlin="relu", hidden_features=256),
}


def main():
    names = sys.argv[1:] or list(NETS)
    dev = torch.device("cuda:0")
    L = _lib.lib()
    side = int(os.environ.get("SIDE", "512"))
    reps = int(os.environ.get("REPS", "10"))
    for name in names:
        torch.manual_seed(0)
        model = models.get_INR(in_features=2, out_features=3, hidden_layers=4, **NETS[name]).to(dev)
        tr = FusedTrainer(model, (side, side), torch.zeros(side * side, 3), lr=5e-3)
        best = {0: 1e9, 1: 1e9}
        for rnd in range(3):
            for knob in (1, 0):
                _lib.check(L.wire_tune_set(b"fused_fwd", knob))
                tr.render()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    tr.render()
                torch.cuda.synchronize()
                best[knob] = min(best[knob], (time.perf_counter() - t0) / reps)
        _lib.check(L.wire_tune_set(b"fused_fwd", 1))
        n = side * side
        print(json.dumps({"net": name, "K": model._arch["width"], "rows": n,
                          "fused_ms": round(best[1] * 1e3, 4), "layerwise_ms": round(best[0] * 1e3, 4),
                          "fused_Msamples_s": round(n / best[1] / 1e6, 1),
                          "layerwise_Msamples_s": round(n / best[0] / 1e6, 1),
                          "speedup": round(best[0] / best[1], 3)}), flush=True)
        del tr, model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
