#!/usr/bin/env python3
"""Random-shape stress of the default path against the fp64 oracle: wire nets of random width / depth / D / O on
random row counts (forward and every parameter gradient).  python tools/fuzz_shapes.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from oracle import wire_oracle as wo
from wire_amd.modules import models



def main(cases: int = 40, seed: int = 0) -> float:
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst = 0.0
    for c in range(cases):
        hf = int(rng.choice([47, 91, 128, 181, 256, 300, 363, 400]))
        L = int(rng.integers(1, 5))
        D = int(rng.integers(1, 4))
        O = int(rng.integers(1, 5))
        n = int(rng.choice([1, 7, 64, 200, 1000, 4096, 5000, 9001]))
        om, sc = float(rng.choice([5.0, 10.0, 20.0])), float(rng.choice([5.0, 10.0, 30.0]))
        torch.manual_seed(c)
        model = models.get_INR(nonlin="wire", in_features=D, out_features=O, hidden_features=hf, hidden_layers=L,
                               first_omega_0=om, hidden_omega_0=om, scale=sc).to(dev)
        coords = rng.uniform(-1, 1, (n, D)).astype(np.float32)
        target = rng.uniform(0, 1, (n, O)).astype(np.float32)
        y = model(torch.tensor(coords, device=dev))
        loss = ((y - torch.tensor(target, device=dev)) ** 2).mean()
        model.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        P = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if "omega_0" not in k and "scale_0" not in k}
        res = {}
        for dbl in (True, False):
            p = wo.cast_params(P, dbl)
            dt = np.float64 if dbl else np.float32
            yy, cache = wo.wire_forward(p, coords.astype(dt), L, dt(om), dt(om), dt(sc), keep=True)
            _, gy = wo.mse_loss_and_grad(yy, target.astype(dt))
            res[dbl] = (yy, wo.wire_backward(p, cache, gy, L, dt(om), dt(om), dt(sc)))
        y64, g64 = res[True]
        y32, g32 = res[False]
        scale = max(np.abs(y64).max(), 1e-3)
        e_ref = np.abs(y32 - y64).max() / scale
        gref_big = {}
        if n < 256:
            # a handful of rows is no statistic: take scale and yardstick from 1024 rows through the same weights
            cb = rng.uniform(-1, 1, (1024, D)).astype(np.float32)
            tb = rng.uniform(0, 1, (1024, O)).astype(np.float32)
            big = {}
            for dbl in (True, False):
                p = wo.cast_params(P, dbl)
                dt = np.float64 if dbl else np.float32
                yy, cache = wo.wire_forward(p, cb.astype(dt), L, dt(om), dt(om), dt(sc), keep=True)
                _, gy = wo.mse_loss_and_grad(yy, tb.astype(dt))
                big[dbl] = (yy, wo.wire_backward(p, cache, gy, L, dt(om), dt(om), dt(sc)))
            scale = max(scale, np.abs(big[True][0]).max())
            e_ref = max(np.abs(y32 - y64).max(), np.abs(big[False][0] - big[True][0]).max()) / scale
            gref_big = {k: np.abs(big[False][1][k] - big[True][1][k]).max() / max(np.abs(big[True][1][k]).max(), 1e-12)
                        for k in big[True][1]}
        e = np.abs(y.detach().cpu().numpy() - y64).max() / scale
        if e_ref > 2e-3:
            # omega0 / s0 large and deep: the reference arithmetic itself is off by > 0.2 % in fp32 (SURVEY section 7,
            # amplification exp(omega0^2 / 4 s0^2) per layer) -- nothing to compare against
            print(f"case {c:3d} hf={hf:3d} L={L} w0={om:4.0f} s0={sc:4.0f}: fp32 reference error {e_ref:.1e} -- ill-conditioned, skipped")
            continue
        bad = e > 4 * e_ref + 1e-5
        ratio = e / (e_ref + 1e-7)
        for k, prm in model.named_parameters():
            if prm.grad is None:
                continue
            g = prm.grad.cpu().numpy()
            s = max(np.abs(g64[k]).max(), 1e-12)
            er, eg = np.abs(g32[k] - g64[k]).max() / s, np.abs(g - g64[k]).max() / s
            er = max(er, gref_big.get(k, 0.0))
            ratio = max(ratio, eg / (er + 0.05 * e_ref + 1e-7))
            bad = bad or eg > 4 * (er + 0.05 * e_ref) + 1e-5
        worst = max(worst, ratio)
        print(f"case {c:3d} hf={hf:3d} K={model._arch['width']:3d} L={L} D={D} O={O} n={n:5d} w0={om:4.0f} s0={sc:4.0f}  "
              f"fwd err {e:.2e} (ref {e_ref:.2e})  worst err/ref {ratio:5.2f}  {'FAIL' if bad else 'ok'}")
        if bad:
            raise AssertionError(f'case {c} outside 4 x the fp32 reference error')
    print(f"all {cases} cases ok; worst error ratio to the fp32 reference arithmetic {worst:.2f}")
    return worst


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
