#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REFERENCE (build container only).

    python3 tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference, Annatk26/wire @ 2024_08_07) is pure Python; it
is imported from its own files, run on CPU (torch CPU, fp32 and an fp64 twin)
and only DATA -- seeds, inputs, outputs, gradients, checksums -- is written.
Nothing of the reference's source travels.  ``cv2`` is not installed; the
reference's ``modules/utils.py`` imports it at module scope only for helpers
that are off the hot path, so an empty stub module is registered first
(SURVEY.md section 8(c)).

Two fixture classes:
  small_*.npz : tiny nets, complete state_dict + every intermediate.
  full_*.npz  : BASELINE.json-size nets.  Weights are NOT stored (MBs); the
                test re-creates them with torch.manual_seed(seed) + nn.Linear
                in the reference's construction order and checks the stored
                checksums before using them.
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, REF)
from modules import gauss, relu, siren, utils, wire, wire2d  # noqa: E402

torch.set_num_threads(8)


def sd_numpy(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def to_double(model):
    for p in model.parameters():
        p.data = p.data.to(torch.cdouble if p.is_complex() else torch.double)
    return model


def checksum(a: np.ndarray):
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = np.stack([a.real, a.imag], -1)
    a = a.astype(np.float64).ravel()
    w = np.cos(np.arange(a.size) * 0.37) + 0.5
    return np.array([a.sum(), np.abs(a).sum(), (a * w).sum()], np.float64)


def build(kind, D, hf, L, O, om1, om, sc, pos_encode=False, sidelength=512):
    if kind == "wire":
        return wire.INR(D, hf, 0, L, O, True, om1, om, sc)
    mod = {"wire2d": wire2d, "siren": siren, "gauss": gauss, "relu": relu}[kind]
    return mod.INR(D, hf, L, O, True, om1, om, sc, pos_encode, sidelength)


def grid_rows(D, n):
    if D == 2:
        W = H = 512
        x = torch.linspace(-1, 1, W)
        y = torch.linspace(-1, 1, H)
        X, Y = torch.meshgrid(x, y, indexing="xy")
        c = torch.hstack((X.reshape(-1, 1), Y.reshape(-1, 1)))
        # take a strided sample so both coordinates vary
        idx = (torch.arange(n) * 1021) % (H * W)
        return c[idx].numpy(), idx.numpy()
    c = utils.get_coords(16, 24, 20)
    idx = (torch.arange(n) * 37) % c.shape[0]
    return c[idx].numpy(), idx.numpy()


def run_case(name, kind, D, hf, L, O, om1, om, sc, N, seed, full, pos_encode=False,
             sidelength=512, lr=5e-3, niters=2000):
    torch.manual_seed(seed)
    model = build(kind, D, hf, L, O, om1, om, sc, pos_encode, sidelength)
    sd0 = sd_numpy(model)
    g = torch.Generator().manual_seed(1000 + seed)
    coords_r = (torch.rand(N // 2, D, generator=g) * 2 - 1)
    cg, gidx = grid_rows(D, N - N // 2)
    coords = torch.cat([coords_r, torch.tensor(cg)], 0)[None]        # [1,N,D]
    target = torch.rand(1, N, O, generator=g)

    rec = {"meta_kind": kind, "meta_D": D, "meta_hidden_features": hf, "meta_L": L,
           "meta_O": O, "meta_first_omega0": float(om1), "meta_hidden_omega0": float(om),
           "meta_scale0": float(sc), "meta_seed": seed, "meta_pos_encode": int(pos_encode),
           "meta_sidelength": sidelength, "meta_torch": torch.__version__,
           "meta_lr": lr, "meta_niters": niters,
           "coords": coords.numpy(), "target": target.numpy(), "grid_idx": gidx}
    nparams = sum(p.numel() for p in model.parameters() if p.requires_grad)
    rec["meta_nparams"] = nparams

    # ---- fp32 forward, per-layer, loss, grads
    x = coords
    if getattr(model, "pos_encode", False):
        x = model.positional_encoding(coords)
        rec["posenc_out"] = x.detach().numpy()
    layer_out = []
    h = x
    for i in range(len(model.net) - 1):
        h = model.net[i](h)
        layer_out.append(h.detach().numpy())
    y = model(coords)
    loss = ((y - target) ** 2).mean()
    model.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
    rec["y"] = y.detach().numpy()
    rec["loss"] = np.float64(loss.item())

    # ---- fp64 twin
    torch.manual_seed(seed)
    m64 = to_double(build(kind, D, hf, L, O, om1, om, sc, pos_encode, sidelength))
    c64, t64 = coords.double(), target.double()
    y64 = m64(c64)
    l64 = ((y64 - t64) ** 2).mean()
    l64.backward()
    grads64 = {k: p.grad.detach().numpy().copy() for k, p in m64.named_parameters() if p.grad is not None}
    rec["y64"] = y64.detach().numpy()
    rec["loss64"] = np.float64(l64.item())

    # ---- 3 Adam steps, LambdaLR per wire_image_denoise.py:123-128
    opt = torch.optim.Adam(lr=lr, params=model.parameters())
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 0.1 ** min(e / niters, 1))
    losses = []
    for _ in range(3):
        yy = model(coords)
        ll = ((yy - target) ** 2).mean()
        opt.zero_grad()
        ll.backward()
        opt.step()
        sched.step()
        losses.append(ll.item())
    rec["adam_losses"] = np.array(losses, np.float64)
    sd3 = sd_numpy(model)

    if not full:
        for k, v in sd0.items():
            rec["p:" + k] = v
        for i, a in enumerate(layer_out):
            rec[f"act{i}"] = a
        for k, v in grads.items():
            rec["g:" + k] = v
        for k, v in grads64.items():
            rec["g64:" + k] = v
        for k, v in sd3.items():
            if "omega_0" in k or "scale_0" in k:
                continue
            rec["p3:" + k] = v
    else:
        for k, v in sd0.items():
            rec["pck:" + k] = checksum(v)
        for i, a in enumerate(layer_out):
            rec[f"act{i}_head"] = a[0, :16, :8]
            rec[f"act{i}_ck"] = checksum(a)
        for k, v in grads.items():
            rec["gck:" + k] = checksum(v)
            rec["ghead:" + k] = v[:8, :8] if v.ndim == 2 else v[:16]
            rec["g64head:" + k] = grads64[k][:8, :8] if v.ndim == 2 else grads64[k][:16]
            rec["g64norm:" + k] = np.float64(np.abs(grads64[k]).max())
            # small tensors are kept whole
            if v.size <= 1024:
                rec["g:" + k] = v
                rec["g64:" + k] = grads64[k]
        for k, v in sd3.items():
            if "omega_0" in k or "scale_0" in k:
                continue
            rec["p3ck:" + k] = checksum(v)
            rec["p3head:" + k] = v[:8, :8] if v.ndim == 2 else v[:16]
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: nparams={nparams} loss={rec['loss']:.6g} "
          f"max|y-y64|/max|y64|={np.abs(rec['y'] - rec['y64']).max() / np.abs(rec['y64']).max():.3g} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def misc():
    rec = {}
    rng = np.random.default_rng(5)
    x = rng.random((17, 19, 3)).astype(np.float32)
    xh = (x + 0.05 * rng.standard_normal(x.shape)).astype(np.float32)
    rec["psnr_x"], rec["psnr_xhat"] = x, xh
    rec["psnr_val"] = np.float64(utils.psnr(x, xh))
    pe = relu.PosEncoding(2, sidelength=512)
    rec["posenc2_nf"] = pe.num_frequencies
    rec["posenc2_in"] = np.array([[[0.25, -0.5]]], np.float32)
    rec["posenc2_out"] = pe(torch.tensor(rec["posenc2_in"])).numpy()
    pe3 = relu.PosEncoding(3, sidelength=512)
    rec["posenc3_nf"] = pe3.num_frequencies
    rec["posenc3_in"] = np.array([[[0.25, -0.5, 0.8]]], np.float32)
    rec["posenc3_out"] = pe3(torch.tensor(rec["posenc3_in"])).numpy()
    rec["coords3d_6_5_4"] = utils.get_coords(6, 5, 4).numpy()
    rec["coords2d_7_9"] = utils.get_coords(7, 9).numpy()
    # image-script coordinates (wire_image_denoise.py:63-66), H=5, W=7 and the
    # 512 / 1024 linspace tables the device coordinate generator must match
    for n in (5, 7, 512, 678, 1020, 1024):
        rec[f"linspace_{n}"] = torch.linspace(-1, 1, n).numpy()
    xs, ys = torch.linspace(-1, 1, 7), torch.linspace(-1, 1, 5)
    X, Y = torch.meshgrid(xs, ys, indexing="xy")
    rec["coords_img_5_7"] = torch.hstack((X.reshape(-1, 1), Y.reshape(-1, 1))).numpy()
    # count_parameters pin: 2x300 denoise net = 91 587 (Agg_results.md:3)
    torch.manual_seed(0)
    m = wire.INR(2, 300, 0, 2, 3, True, 7.0, 7.0, 8.0)
    rec["nparams_2x300"] = utils.count_parameters(m)
    rec["sd_keys_2x300"] = np.array(list(m.state_dict().keys()))
    # IoU known answer (modules/volutils.py is not importable: mcubes/open3d
    # absent) -> not generated; the oracle's iou() is "parity unpinned".
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **rec)
    print("misc: nparams_2x300 =", rec["nparams_2x300"], "psnr =", rec["psnr_val"])


def misc2():
    """Round-2 known answers: volutils.get_IoU (numpy path; mcubes / open3d / skimage are absent and only needed by
    the mesh helpers, so empty stubs are registered as for cv2), ComplexGaborLayer(trainable=True) gradients,
    outermost_linear=False nets, a wire net without hidden layers."""
    for name in ("mcubes", "open3d", "skimage", "skimage.metrics"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skimage.metrics"].structural_similarity = None
    from modules import volutils
    rec = {}
    rng = np.random.default_rng(11)
    # ---- IoU (modules/volutils.py:74-91): preds binarised IN PLACE at thres, then |and| / |or|
    pred = rng.random(5000).astype(np.float32)
    gt = (rng.random(5000) > 0.6).astype(np.float32)
    work = pred.copy()
    rec["iou_pred"], rec["iou_gt"], rec["iou_thres"] = pred, gt, np.float32(0.5)
    rec["iou_val"] = np.float64(volutils.get_IoU(work, gt, 0.5))
    rec["iou_pred_after"] = work                               # the in-place binarisation the caller observes
    rec["iou_val_nothres"] = np.float64(volutils.get_IoU((pred > 0.7).astype(np.float32), gt, None))
    rec["iou_batch_val"] = np.float64(volutils.get_IoU_batch(torch.tensor(pred.copy()), torch.tensor(gt), 0.5, 1024)) \
        if False else np.float64(-1)                            # torch path calls .cuda(): not runnable here
    # ---- ComplexGaborLayer(trainable=True): gradients of omega_0 / scale_0 (modules/wire.py:80-81)
    for tag, is_first, fin in (("hid", False, 24), ("first", True, 3)):
        torch.manual_seed(21)
        layer = wire.ComplexGaborLayer(fin, 40, is_first=is_first, omega0=9.0, sigma0=4.0, trainable=True)
        n = 300
        if is_first:
            x = torch.tensor(rng.uniform(-1, 1, (n, fin)).astype(np.float32))
        else:
            x = torch.tensor((0.4 * (rng.standard_normal((n, fin)) + 1j * rng.standard_normal((n, fin)))).astype(np.complex64))
        g = torch.tensor((rng.standard_normal((n, 40)) + 1j * rng.standard_normal((n, 40))).astype(np.complex64))
        for dbl in (False, True):
            lay = wire.ComplexGaborLayer(fin, 40, is_first=is_first, omega0=9.0, sigma0=4.0, trainable=True)
            lay.load_state_dict(layer.state_dict())
            xx = x.clone()
            gg = g.clone()
            if dbl:
                to_double(lay)
                xx = xx.to(torch.double if is_first else torch.cdouble)
                gg = gg.to(torch.cdouble)
            xx.requires_grad_(not is_first)
            out = lay(xx)
            (out.real * gg.real + out.imag * gg.imag).sum().backward()
            sfx = "64" if dbl else ""
            rec[f"tr_{tag}_out{sfx}"] = out.detach().numpy()
            rec[f"tr_{tag}_g_omega{sfx}"] = lay.omega_0.grad.numpy()
            rec[f"tr_{tag}_g_scale{sfx}"] = lay.scale_0.grad.numpy()
            rec[f"tr_{tag}_g_W{sfx}"] = lay.linear.weight.grad.numpy()
            rec[f"tr_{tag}_g_b{sfx}"] = lay.linear.bias.grad.numpy()
            if not is_first:
                rec[f"tr_{tag}_g_x{sfx}"] = xx.grad.numpy()
        rec[f"tr_{tag}_x"], rec[f"tr_{tag}_g"] = x.numpy(), g.numpy()
        for k, v in layer.state_dict().items():
            rec[f"tr_{tag}_p:{k}"] = v.numpy()
    # ---- outermost_linear=False (modules/siren.py:81-84, gauss.py:63-66, relu.py:116-119) and wire with L = 0
    coords = torch.tensor(rng.uniform(-1, 1, (1, 200, 2)).astype(np.float32))
    target = torch.tensor(rng.uniform(0, 1, (1, 200, 3)).astype(np.float32))
    rec["ol_coords"], rec["ol_target"] = coords.numpy(), target.numpy()
    cases = [("siren", lambda: siren.INR(2, 48, 2, 3, False, 30.0, 30.0, 10.0)),
             ("gauss", lambda: gauss.INR(2, 48, 2, 3, False, 30.0, 30.0, 10.0)),
             ("relu", lambda: relu.INR(2, 48, 2, 3, False, 30.0, 30.0, 10.0)),
             ("wireL0", lambda: wire.INR(2, 64, 0, 0, 3, True, 7.0, 7.0, 6.0))]
    for tag, mk in cases:
        torch.manual_seed(31)
        model = mk()
        for k, v in model.state_dict().items():
            rec[f"ol_{tag}_p:{k}"] = v.numpy().copy()
        for dbl in (False, True):
            m2 = mk()
            m2.load_state_dict(model.state_dict())
            c, t = coords, target
            if dbl:
                to_double(m2)
                c, t = coords.double(), target.double()
            y = m2(c)
            loss = ((y - t) ** 2).mean()
            loss.backward()
            sfx = "64" if dbl else ""
            rec[f"ol_{tag}_y{sfx}"] = y.detach().numpy()
            for k, prm in m2.named_parameters():
                if prm.grad is not None:
                    rec[f"ol_{tag}_g{sfx}:{k}"] = prm.grad.numpy()
    rec["meta_torch"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, "misc2.npz"), **rec)
    print("misc2: iou =", rec["iou_val"], "trainable g_omega (hid) =", rec["tr_hid_g_omega"], rec["tr_hid_g_omega64"])


def misc3():
    """ComplexGaborLayer2D(trainable=True) (modules/wire2d.py:21-67): gradients of omega_0 / scale_0 and of both
    Linears, hidden and first layer, fp32 and fp64."""
    rec = {}
    rng = np.random.default_rng(13)
    for tag, is_first, fin in (("hid", False, 20), ("first", True, 2)):
        torch.manual_seed(41)
        layer = wire2d.ComplexGaborLayer2D(fin, 36, is_first=is_first, omega0=6.0, sigma0=3.0, trainable=True)
        n = 256
        if is_first:
            x = torch.tensor(rng.uniform(-1, 1, (n, fin)).astype(np.float32))
        else:
            x = torch.tensor((0.3 * (rng.standard_normal((n, fin)) + 1j * rng.standard_normal((n, fin)))).astype(np.complex64))
        g = torch.tensor((rng.standard_normal((n, 36)) + 1j * rng.standard_normal((n, 36))).astype(np.complex64))
        for dbl in (False, True):
            lay = wire2d.ComplexGaborLayer2D(fin, 36, is_first=is_first, omega0=6.0, sigma0=3.0, trainable=True)
            lay.load_state_dict(layer.state_dict())
            xx, gg = x.clone(), g.clone()
            if dbl:
                to_double(lay)
                xx = xx.to(torch.double if is_first else torch.cdouble)
                gg = gg.to(torch.cdouble)
            xx.requires_grad_(not is_first)
            out = lay(xx)
            (out.real * gg.real + out.imag * gg.imag).sum().backward()
            sfx = "64" if dbl else ""
            rec[f"tr2d_{tag}_out{sfx}"] = out.detach().numpy()
            for nm, prm in (("g_omega", lay.omega_0), ("g_scale", lay.scale_0), ("g_W", lay.linear.weight),
                            ("g_b", lay.linear.bias), ("g_V", lay.scale_orth.weight), ("g_c", lay.scale_orth.bias)):
                rec[f"tr2d_{tag}_{nm}{sfx}"] = prm.grad.numpy()
            if not is_first:
                rec[f"tr2d_{tag}_g_x{sfx}"] = xx.grad.numpy()
        rec[f"tr2d_{tag}_x"], rec[f"tr2d_{tag}_g"] = x.numpy(), g.numpy()
        for k, v in layer.state_dict().items():
            rec[f"tr2d_{tag}_p:{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "misc3.npz"), **rec)
    print("misc3: g_omega", rec["tr2d_hid_g_omega"], rec["tr2d_hid_g_omega64"], "g_scale", rec["tr2d_hid_g_scale64"])


def ct_pair():
    """The reference holds one input/output pair of its CT forward operator (lin_inverse.radon, which needs kornia):
    the phantom ``gt`` and its ``sinogram`` over np.linspace(0, 180, 100) degrees, saved by wire_ct.py:160-163 into
    multiscale_results/ct/Original/WIRE_s12_o8_LR5e3_E2000_1/info.mat.  Data only: copied into a fixture."""
    from scipy.io import loadmat
    d = loadmat(os.path.join(REF, "multiscale_results/ct/Original/WIRE_s12_o8_LR5e3_E2000_1/info.mat"))
    r = d[[k for k in d if not k.startswith("__")][0]][0, 0]
    gt, sino = r["gt"].astype(np.float32), r["sinogram"].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ct_pair.npz"), gt=gt, sinogram=sino,
                        thetas=np.linspace(0, 180, sino.shape[0], dtype=np.float32))
    print("ct_pair:", gt.shape, sino.shape)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "misc2":
        misc2()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "misc3":
        misc3()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ct_pair":
        ct_pair()
        sys.exit(0)
    N = 512
    # ---- small, everything stored
    run_case("small_wire_d2", "wire", 2, 64, 2, 3, 7.0, 7.0, 6.0, N, 0, False)
    run_case("small_wire_d3", "wire", 3, 40, 3, 1, 20.0, 20.0, 10.0, N, 1, False)
    run_case("small_wire_hi", "wire", 2, 48, 2, 3, 30.0, 30.0, 10.0, N, 2, False)
    run_case("small_wire2d", "wire2d", 2, 64, 2, 3, 10.0, 10.0, 10.0, N, 3, False)
    run_case("small_siren", "siren", 2, 64, 2, 3, 30.0, 30.0, 10.0, N, 4, False)
    run_case("small_gauss", "gauss", 2, 64, 2, 3, 30.0, 30.0, 10.0, N, 5, False)
    run_case("small_relu", "relu", 2, 64, 2, 3, 30.0, 30.0, 10.0, N, 6, False)
    run_case("small_posenc", "relu", 2, 64, 2, 3, 30.0, 30.0, 10.0, N, 7, False,
             pos_encode=True, sidelength=512)
    # ---- BASELINE.json configs, checksummed
    NF = 256
    run_case("full_cfg1_wire_2x128", "wire", 2, 128, 2, 3, 7.0, 7.0, 6.0, NF, 0, True)
    run_case("full_cfg2_wire_4x256_api", "wire", 2, 256, 4, 3, 20.0, 20.0, 30.0, NF, 0, True)
    run_case("full_cfg2_wire_4x363_lit", "wire", 2, 363, 4, 3, 20.0, 20.0, 30.0, NF, 0, True)
    run_case("full_cfg2_wire_4x256_def", "wire", 2, 256, 4, 3, 30.0, 30.0, 10.0, NF, 0, True)
    run_case("full_cfg3_wire_3x300_d3", "wire", 3, 300, 3, 1, 20.0, 20.0, 10.0, NF, 0, True)
    run_case("full_cfg3_wire_4x363_d3", "wire", 3, 363, 4, 1, 20.0, 20.0, 10.0, NF, 0, True)
    run_case("full_denoise_wire_2x300", "wire", 2, 300, 2, 3, 7.0, 7.0, 8.0, NF, 0, True)
    run_case("full_cfg4_wire2d_4x256", "wire2d", 2, 256, 4, 3, 10.0, 10.0, 10.0, NF, 0, True)
    run_case("full_cfg5_siren_4x256", "siren", 2, 256, 4, 3, 30.0, 30.0, 10.0, NF, 0, True)
    run_case("full_cfg5_gauss_4x256", "gauss", 2, 256, 4, 3, 30.0, 30.0, 10.0, NF, 0, True)
    run_case("full_cfg5_relu_4x256", "relu", 2, 256, 4, 3, 30.0, 30.0, 10.0, NF, 0, True)
    run_case("full_cfg5_posenc_4x256", "relu", 2, 256, 4, 3, 30.0, 30.0, 10.0, NF, 0, True,
             pos_encode=True, sidelength=512)
    misc()
    misc2()
    misc3()
    ct_pair()
