#!/usr/bin/env python3
"""Where do the weight-gradient ratios of the wire2d 1024 x 1024 step come from?  

profiles/r03_parity_ratios.txt: step[cfg4_wire2d_4x256_1024x1024] grad net.2.linear.weight 1.14e-6 against the numpy fp32
yardstick's 2.65e-7 = 4.3 x (passes through the protocol's 1e-6 floor only).  DESIGN 4.3 blamed the summation order -- the
weight-gradient kernel adds the rows of a split SEQUENTIALLY into its fp32 accumulators (wire2d at 1 048 576 rows: 128
splits of 8 192 rows, three accumulations per 32 rows), numpy's sgemm blocks the reduction -- without a measurement.
This tool measures it on the GPU (autograd of modules/wire2d.py:56-67 at the bench size of BASELINE.json configs[3]):

  * the same step with the sequential chain bounded to 8 192 (default) / 4 096 / 2 048 / 1 024 / 512 rows
    (knob "x2_tn_rows": more row splits, the slabs summed by wgrad_reduce_kernel in four interleaved groups);
  * the same step on the other GEMM families -- 3 x bf16 split, exact-fp32 MFMA ("4m": a k-ordered fmaf chain per split);
  * the numpy fp32 yardstick with 16 384-row chunks (what the tests use) and with 1 024-row chunks.

    python3 tools/wgrad_order_probe.py [side]          (default 1024; prints one table, about 3 minutes)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import family_ctx, oracle_grads_chunked, params_np, relmax  # noqa: E402
from oracle import wire_oracle as wo  # noqa: E402
from wire_amd import _lib  # noqa: E402
from wire_amd.modules import models  # noqa: E402
from wire_amd.trainer import FusedTrainer  # noqa: E402

DEV = "cuda"


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    kw = dict(nonlin="wire2d", hidden_features=256, first_omega_0=10.0, hidden_omega_0=10.0, scale=10.0)
    N = side * side
    g = torch.Generator().manual_seed(11)
    target = torch.rand(N, 3, generator=g)
    perm = torch.randperm(N, generator=g)
    L = _lib.lib()

    def gpu_step(fam, tn_rows):
        torch.manual_seed(0)
        model = models.get_INR(in_features=2, out_features=3, hidden_layers=4, **kw).to(DEV)
        _lib.check(L.wire_tune_set(b"x2_tn_rows", tn_rows))
        try:
            with family_ctx(fam):
                tr = FusedTrainer(model, (side, side), target, lr=0.0)
                tr.step(perm.to(DEV))
                torch.cuda.synchronize()
                flat = tr.flat_grad.cpu().numpy().copy()
                offs = list(tr.offsets)
        finally:
            _lib.check(L.wire_tune_set(b"x2_tn_rows", 0))
        names = [k for k in model.state_dict().keys() if "omega_0" not in k and "scale_0" not in k]
        P = params_np(model)
        del tr, model
        torch.cuda.empty_cache()
        return flat, offs, names, P

    flat0, offs, names, P = gpu_step("x2", 0)
    coords = wo.image_coords(side, side)[perm.numpy()]
    tgt = target.numpy()[perm.numpy()]
    _, _, g64 = oracle_grads_chunked("wire2d", P, coords, tgt, 4, 10.0, 10.0, 10.0, True)
    _, _, g32 = oracle_grads_chunked("wire2d", P, coords, tgt, 4, 10.0, 10.0, 10.0, False)
    _, _, g32s = oracle_grads_chunked("wire2d", P, coords, tgt, 4, 10.0, 10.0, 10.0, False, chunk=1024)
    hidden = [n for n in names if n.endswith("weight") and n.split(".")[1] in ("1", "2", "3", "4")]

    def errs(flat):
        out = {}
        for name, off in zip(names, offs):
            if name in hidden:
                ref = wo.as_real_pairs(g64[name]).astype(np.float64).ravel()
                out[name] = relmax(flat[off:off + ref.size], ref)
        return out

    def ref_errs(gx):
        return {n: relmax(wo.as_real_pairs(gx[n]).astype(np.float64).ravel(),
                          wo.as_real_pairs(g64[n]).astype(np.float64).ravel()) for n in hidden}

    rows = [("numpy fp32 yardstick, 16384-row chunks (tests)", ref_errs(g32)),
            ("numpy fp32, 1024-row chunks", ref_errs(g32s)),
            ("GPU 2 x fp16, default splits (chain = n / 128)", errs(flat0))]
    for tn_rows in (4096, 2048, 1024, 512):
        if tn_rows * 128 >= N:
            continue
        rows.append((f"GPU 2 x fp16, chain <= {tn_rows} rows", errs(gpu_step("x2", tn_rows)[0])))
    rows.append(("GPU 3 x bf16 family", errs(gpu_step("x3", 0)[0])))
    rows.append(("GPU exact-fp32 MFMA family (4m)", errs(gpu_step("4m", 0)[0])))
    print(f"wire2d 4x256, {side} x {side} = {N} rows: max |g - g64| / max |g64| of the hidden weight gradients")
    print(f"{'':52s}" + "".join(f"{n.replace('net.', '').replace('.weight', ''):>18s}" for n in hidden))
    base = rows[0][1]
    for label, e in rows:
        print(f"{label:52s}" + "".join(f"{e[n]:10.2e} ({e[n] / base[n]:4.1f}x)" for n in hidden))


if __name__ == "__main__":
    main()
