"""The oracle vs. vectors generated from the reference itself (CPU, no GPU).

Pins oracle/wire_oracle.py (numpy closed form) and oracle/torch_ref.py (eager
restatement) against tests/golden/*.npz, which tests/golden/make_golden.py
produced by importing /root/reference.  Tolerances: fp64 runs must agree with
the reference's fp64 twin to ~1e-12 (same formula, different op order); fp32
runs are compared with the reference's fp32 run at a few ulp scaled by the
reference's own fp32-vs-fp64 error (SURVEY.md section 7 "Precision").
"""
import numpy as np
import pytest
import torch

from _util import FULL, SMALL, checksum, load_golden, meta, oracle_run, params_np, build_model, relmax
from oracle import torch_ref, wire_oracle as wo


def small_params(rec):
    return {k[2:]: v for k, v in rec.items() if k.startswith("p:") and "omega_0" not in k
            and "scale_0" not in k}


@pytest.mark.parametrize("name", SMALL)
def test_small_fp64_matches_reference_twin(name):
    rec = load_golden(name)
    y, loss, grads, _ = oracle_run(rec, small_params(rec), double=True)
    assert relmax(y, rec["y64"]) < 1e-11
    assert abs(loss - float(rec["loss64"])) < 1e-12 * max(1.0, abs(loss))
    for k, g in grads.items():
        assert relmax(g, rec["g64:" + k]) < 1e-10, k


@pytest.mark.parametrize("name", SMALL)
def test_small_fp32_matches_reference_fp32(name):
    rec = load_golden(name)
    y, loss, grads, cache = oracle_run(rec, small_params(rec), double=False)
    ref_err = relmax(rec["y"], rec["y64"])          # the reference's own round-off
    tol = 4 * ref_err + 2e-6
    assert relmax(y, rec["y"]) < tol
    assert relmax(y, rec["y64"]) < tol
    for i, a in enumerate(cache["out"]):
        assert relmax(a, rec[f"act{i}"]) < tol, f"layer {i}"
    for k, g in grads.items():
        gref_err = relmax(rec["g:" + k], rec["g64:" + k])
        assert relmax(g, rec["g64:" + k]) < 4 * gref_err + 2e-6, k


@pytest.mark.parametrize("name", FULL)
def test_full_init_and_outputs(name):
    """BASELINE.json-size nets: wire_amd's constructors reproduce the
    reference's initial state_dict (checksums), and the oracle on those weights
    reproduces the reference's outputs and gradients."""
    rec = load_golden(name)
    model = build_model(rec)
    P = params_np(model)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(rec["meta_nparams"])
    for k, v in P.items():
        np.testing.assert_allclose(checksum(v), rec["pck:" + k], rtol=1e-12, atol=1e-12, err_msg=k)
    y64, loss64, g64, _ = oracle_run(rec, P, double=True)
    assert relmax(y64, rec["y64"]) < 1e-10
    y32, loss32, g32, cache = oracle_run(rec, P, double=False)
    ref_err = relmax(rec["y"], rec["y64"])
    assert relmax(y32, rec["y64"]) < 4 * ref_err + 2e-6
    for i, a in enumerate(cache["out"]):
        assert relmax(a[0, :16, :8], rec[f"act{i}_head"]) < 4 * ref_err + 1e-5
    for k, g in g64.items():
        scale = float(rec["g64norm:" + k])
        head = g[:8, :8] if g.ndim == 2 else g[:16]
        assert np.abs(head - rec["g64head:" + k]).max() < 1e-9 * max(scale, 1e-30), k


def test_adam_restatement_small():
    """3 Adam steps with LambdaLR (wire_image_denoise.py:123-128) through the
    numpy oracle land on the reference's parameters."""
    rec = load_golden("small_wire_d2")
    m = meta(rec)
    P = wo.cast_params(small_params(rec), False)
    state = {k: (np.zeros_like(wo.as_real_pairs(v)), np.zeros_like(wo.as_real_pairs(v))) for k, v in P.items()}
    losses = []
    for step in range(1, 4):
        y, cache = wo.wire_forward(P, rec["coords"], m["L"], np.float32(m["om1"]), np.float32(m["om"]),
                                   np.float32(m["sc"]), keep=True)
        loss, gy = wo.mse_loss_and_grad(y, rec["target"])
        losses.append(loss)
        grads = wo.wire_backward(P, cache, gy.astype(np.float32), m["L"], np.float32(m["om1"]),
                                 np.float32(m["om"]), np.float32(m["sc"]))
        lr = wo.lambda_lr(m["lr"], step - 1, m["niters"])
        for k in P:
            pr = wo.as_real_pairs(P[k]).astype(np.float32)
            gr = wo.as_real_pairs(grads[k]).astype(np.float32)
            pn, mm, vv = wo.adam_step(pr, gr, state[k][0], state[k][1], step, lr)
            state[k] = (mm, vv)
            pn = pn.astype(np.float32)
            P[k] = (pn[..., 0] + 1j * pn[..., 1]).astype(np.complex64) if np.iscomplexobj(P[k]) else pn
    np.testing.assert_allclose(losses, rec["adam_losses"], rtol=2e-4)
    for k in P:
        # Adam's first steps are +-lr regardless of gradient scale, so compare
        # against the step size, not the parameter magnitude
        assert np.abs(P[k] - rec["p3:" + k]).max() < 0.05 * m["lr"], k


def test_torch_ref_matches_reference():
    rec = load_golden("small_wire_hi")
    m = meta(rec)
    p = {k: torch.tensor(v) for k, v in small_params(rec).items()}
    y, acts = torch_ref.wire_forward(p, torch.tensor(rec["coords"]), m["L"], m["om1"], m["om"],
                                     m["sc"], keep=True)
    assert relmax(y.numpy(), rec["y"]) < 1e-5
    for i, a in enumerate(acts):
        assert relmax(a.numpy(), rec[f"act{i}"]) < 1e-5
    losses, p3 = torch_ref.train_steps(p, torch.tensor(rec["coords"]), torch.tensor(rec["target"]),
                                       m["L"], m["om1"], m["om"], m["sc"], m["lr"], 3, m["niters"])
    np.testing.assert_allclose(losses, rec["adam_losses"], rtol=1e-5)
    for k, v in p3.items():
        assert np.abs(v.numpy() - rec["p3:" + k]).max() < 1e-5


def test_torch_ref_init_matches_reference_checksums():
    rec = load_golden("full_cfg2_wire_4x256_api")
    m = meta(rec)
    p = torch_ref.init_wire_params(m["D"], m["hf"], m["L"], m["O"], seed=m["seed"])
    for k, v in p.items():
        np.testing.assert_allclose(checksum(v.numpy()), rec["pck:" + k], rtol=1e-12, err_msg=k)


def test_misc_known_answers():
    rec = load_golden("misc")
    assert abs(wo.psnr(rec["psnr_x"], rec["psnr_xhat"]) - float(rec["psnr_val"])) < 1e-4
    assert wo.posenc_num_frequencies(2, 512) == int(rec["posenc2_nf"]) == 7
    assert wo.posenc_num_frequencies(3, 512) == int(rec["posenc3_nf"]) == 10
    np.testing.assert_allclose(wo.posenc(rec["posenc2_in"], 7), rec["posenc2_out"], atol=2e-6)
    np.testing.assert_allclose(wo.posenc(rec["posenc3_in"], 10), rec["posenc3_out"], atol=3e-5)
    np.testing.assert_array_equal(wo.volume_coords(6, 5, 4), rec["coords3d_6_5_4"])
    np.testing.assert_array_equal(wo.image_coords(5, 7), rec["coords_img_5_7"])
    for n in (5, 7, 512, 678, 1020, 1024):
        np.testing.assert_array_equal(wo.linspace_f32(n), rec[f"linspace_{n}"])
    assert int(rec["nparams_2x300"]) == 91587
    assert wo.wire_flops_per_sample(256, 4, 2, 3) == 6302720
    assert wo.wire_flops_per_sample(181, 4, 2, 3) == 3153020


@pytest.mark.parametrize("H,W,scale", [(12, 12, 3), (13, 10, 4), (8, 9, 1)])
def test_avgpool_loss_restatement_matches_torch(H, W, scale):
    """The super-resolution loss (wire_SISR.py:151-161) is torch.nn.AvgPool2d + MSE: the oracle's numpy
    restatement against that operator and its autograd, fp64."""
    import torch
    rng = np.random.default_rng(H * 100 + W)
    O = 3
    y = rng.standard_normal((H * W, O))
    H2, W2 = H // scale, W // scale
    gt = rng.standard_normal((H2 * W2, O))
    loss, g, rec = wo.avgpool_mse_loss_and_grad(y, H, W, scale, gt)
    yt = torch.tensor(y, requires_grad=True)
    pooled = torch.nn.AvgPool2d(scale)(yt.reshape(H, W, O).permute(2, 0, 1)[None, ...])
    lt = ((torch.tensor(gt)[None, ...] - pooled.reshape(1, O, -1).permute(0, 2, 1)) ** 2).mean()
    lt.backward()
    assert abs(loss - float(lt.detach())) <= 1e-14 * max(1.0, abs(float(lt.detach())))
    np.testing.assert_allclose(g, yt.grad.numpy(), rtol=0, atol=1e-15)
    np.testing.assert_allclose(rec, pooled.reshape(O, -1).T.detach().numpy(), rtol=0, atol=1e-14)


def test_iou_pinned_by_reference_known_answer():
    """oracle.iou == modules/volutils.get_IoU on the reference-generated vector (tests/golden/misc2.npz), and the
    reference's in-place binarisation of ``preds`` is what the fixture recorded."""
    m2 = load_golden("misc2")
    val = wo.iou(m2["iou_pred"], m2["iou_gt"], float(m2["iou_thres"]))
    assert abs(val - float(m2["iou_val"])) < 1e-12
    after = np.where(m2["iou_pred"] < m2["iou_thres"], 0.0, 1.0).astype(np.float32)
    np.testing.assert_array_equal(after, m2["iou_pred_after"])
    assert abs(wo.iou((m2["iou_pred"] > 0.7).astype(np.float32), m2["iou_gt"], None) - float(m2["iou_val_nothres"])) < 1e-12


def test_trainable_gabor_gradients_closed_form():
    """d out / d omega_0 = j lin out, d out / d scale_0 = -2 s0 |lin|^2 out (ComplexGaborLayer(trainable=True),
    modules/wire.py:80-81) against the reference's autograd (fp64 fixture)."""
    m2 = load_golden("misc2")
    for tag in ("hid", "first"):
        W = m2[f"tr_{tag}_p:linear.weight"]
        b = m2[f"tr_{tag}_p:linear.bias"]
        x, g = m2[f"tr_{tag}_x"], m2[f"tr_{tag}_g"].astype(np.complex128)
        lin = x.astype(np.complex128 if np.iscomplexobj(W) else np.float64) @ W.astype(lin_dtype(W)).T + b
        out = wo.gabor_act(lin, 9.0, 4.0)
        assert relmax(out, m2[f"tr_{tag}_out64"]) < 1e-12
        c = np.conj(out) * g
        g_om = np.sum(lin.real * c.imag - lin.imag * c.real)
        g_sc = -2.0 * 4.0 * np.sum(np.abs(lin) ** 2 * c.real)
        assert abs(g_om - float(m2[f"tr_{tag}_g_omega64"][0])) <= 1e-10 * abs(g_om)
        assert abs(g_sc - float(m2[f"tr_{tag}_g_scale64"][0])) <= 1e-10 * abs(g_sc)


def lin_dtype(W):
    return np.complex128 if np.iscomplexobj(W) else np.float64


def test_relu_forced_decisions_reduce_to_the_plain_oracle():
    """oracle.realnet_backward(relu_masks=...) (the identical-decisions comparison of the relu step test): with the
    oracle's own decisions (lin > 0) it IS the plain backward; with one decision flipped only that element's
    contribution changes -- by exactly g_out z^T for the weight gradient of its layer."""
    from _util import build_model, load_golden, meta, params_np
    from oracle import wire_oracle as wo
    rec = load_golden("small_relu")
    m = meta(rec)
    P = wo.cast_params(params_np(build_model(rec)), True)
    c = rec["coords"].reshape(-1, m["D"]).astype(np.float64)
    t = rec["target"].reshape(-1, m["O"]).astype(np.float64)
    y, cache = wo.realnet_forward("relu", P, c, m["L"], m["om1"], m["om"], m["sc"], None, keep=True)
    _, gy = wo.mse_loss_and_grad(y, t)
    g0 = wo.realnet_backward("relu", P, cache, gy, m["L"], m["om1"], m["om"], m["sc"])
    masks = [lin > 0 for lin in cache["lin"]]
    g1 = wo.realnet_backward("relu", P, cache, gy, m["L"], m["om1"], m["om"], m["sc"], relu_masks=masks)
    for k in g0:
        np.testing.assert_array_equal(g0[k], g1[k])
    masks[m["L"]] = masks[m["L"]].copy()
    masks[m["L"]][3, 5] ^= True
    g2 = wo.realnet_backward("relu", P, cache, gy, m["L"], m["om1"], m["om"], m["sc"], relu_masks=masks)
    Wf = P[f"net.{m['L'] + 1}.weight"]
    g_out = (gy @ Wf)[3, 5]
    sign = 1.0 if masks[m["L"]][3, 5] else -1.0
    d = g2[f"net.{m['L']}.linear.weight"] - g0[f"net.{m['L']}.linear.weight"]
    expect = np.zeros_like(d)
    expect[5, :] = sign * g_out * cache["out"][m["L"] - 1][3]
    np.testing.assert_allclose(d, expect, atol=1e-15)
