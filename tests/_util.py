"""Shared helpers of the test-suite (fixture loading, model rebuild, metrics)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SMALL = ["small_wire_d2", "small_wire_d3", "small_wire_hi", "small_wire2d", "small_siren",
         "small_gauss", "small_relu", "small_posenc"]
FULL = ["full_cfg1_wire_2x128", "full_cfg2_wire_4x256_api", "full_cfg2_wire_4x363_lit",
        "full_cfg2_wire_4x256_def", "full_cfg3_wire_3x300_d3", "full_cfg3_wire_4x363_d3",
        "full_denoise_wire_2x300", "full_cfg4_wire2d_4x256", "full_cfg5_siren_4x256",
        "full_cfg5_gauss_4x256", "full_cfg5_relu_4x256", "full_cfg5_posenc_4x256"]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def meta(rec):
    return dict(kind=str(rec["meta_kind"]), D=int(rec["meta_D"]), hf=int(rec["meta_hidden_features"]),
                L=int(rec["meta_L"]), O=int(rec["meta_O"]), om1=float(rec["meta_first_omega0"]),
                om=float(rec["meta_hidden_omega0"]), sc=float(rec["meta_scale0"]),
                seed=int(rec["meta_seed"]), pos=bool(int(rec["meta_pos_encode"])),
                side=int(rec["meta_sidelength"]), lr=float(rec["meta_lr"]),
                niters=int(rec["meta_niters"]))


def checksum(a):
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = np.stack([a.real, a.imag], -1)
    a = a.astype(np.float64).ravel()
    w = np.cos(np.arange(a.size) * 0.37) + 0.5
    return np.array([a.sum(), np.abs(a).sum(), (a * w).sum()], np.float64)


def build_model(rec, device="cpu"):
    """wire_amd model built the way the golden generator built the reference's:
    torch.manual_seed(seed) then the constructor (same RNG consumption)."""
    from wire_amd.modules import models
    m = meta(rec)
    torch.manual_seed(m["seed"])
    model = models.get_INR(nonlin=m["kind"], in_features=m["D"], out_features=m["O"],
                           hidden_features=m["hf"], hidden_layers=m["L"], first_omega_0=m["om1"],
                           hidden_omega_0=m["om"], scale=m["sc"], pos_encode=m["pos"],
                           sidelength=m["side"])
    return model.to(device)


def params_np(model):
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()
            if "omega_0" not in k and "scale_0" not in k}


def oracle_run(rec, params, double):
    """Forward+backward of the numpy oracle on the fixture's inputs."""
    from oracle import wire_oracle as wo
    m = meta(rec)
    p = wo.cast_params(params, double)
    rdt = np.float64 if double else np.float32
    coords = rec["coords"].astype(rdt)
    target = rec["target"].astype(rdt)
    om1, om, sc = rdt(m["om1"]), rdt(m["om"]), rdt(m["sc"])
    kind = m["kind"]
    if kind == "wire":
        y, cache = wo.wire_forward(p, coords, m["L"], om1, om, sc, keep=True)
        loss, gy = wo.mse_loss_and_grad(y, target)
        grads = wo.wire_backward(p, cache, gy, m["L"], om1, om, sc)
    elif kind == "wire2d":
        y, cache = wo.wire2d_forward(p, coords, m["L"], om1, om, sc, keep=True)
        loss, gy = wo.mse_loss_and_grad(y, target)
        grads = wo.wire2d_backward(p, cache, gy, m["L"], om1, om, sc)
    else:
        nf = wo.posenc_num_frequencies(m["D"], m["side"]) if m["pos"] else None
        y, cache = wo.realnet_forward(kind, p, coords, m["L"], om1, om, sc, nf, keep=True)
        loss, gy = wo.mse_loss_and_grad(y, target)
        grads = wo.realnet_backward(kind, p, cache, gy, m["L"], om1, om, sc)
    return y, float(loss), grads, cache


def relmax(a, b):
    """max |a - b| / max |b|"""
    a, b = np.asarray(a), np.asarray(b)
    d = np.abs(a - b).max()
    s = np.abs(b).max()
    return float(d / s) if s > 0 else float(d)
