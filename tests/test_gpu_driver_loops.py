"""The reference's DRIVER LOOPS, restated call for call, on the drop-in modules (SURVEY.md section 8 row a10:
"callers unchanged").

wire_image_denoise.py:104-178 and wire_occupancy.py:107-172 are not runnable as files (hard-coded paths, missing
data); their loops are restated here with the same calls a maintainer's script makes after switching the import
from ``modules`` to ``wire_amd.modules``: ``models.get_INR(nonlin=..., keywords...)``, ``model.cuda()``,
``torch.optim.Adam(params=model.parameters())`` on the complex64 parameters, ``LambdaLR``, HOST ``torch.randperm``,
``b_coords = coords[:, b_indices, ...].cuda()``, ``rec[:, b_indices, :] = pixelvalues``, ``loss.backward()``,
``optim.step()``, the best-image bookkeeping, ``utils.psnr`` / ``volutils.get_IoU``.  The same loop on the CPU
restatement of the reference (oracle/torch_ref.driver_loop) is the yardstick: loss trajectory, final PSNR / IoU.

Also here: the constructor paths the reference offers on this path -- ``ComplexGaborLayer(trainable=True)``,
``outermost_linear=False``, ``hidden_layers=0`` -- against vectors generated from the reference
(tests/golden/misc2.npz), the device IoU against the reference's known answer, and the device-side best-image
tracking against the host logic.
"""
import copy

import numpy as np
import pytest
import torch
from torch.optim.lr_scheduler import LambdaLR

from _util import load_golden, relmax, within_ref
from oracle import torch_ref, wire_oracle as wo

pytestmark = pytest.mark.gpu


def _smooth_image(H, W):
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    img = np.stack([0.5 + 0.4 * np.sin(3 * xx + 2 * yy), 0.5 + 0.4 * np.cos(4 * xx * yy),
                    0.5 + 0.3 * np.sin(5 * yy) * np.cos(2 * xx)], -1)
    return img.astype(np.float32)


def test_image_denoise_driver_loop():
    """wire_image_denoise.py:104-178 on a 32 x 32 image, 2 x 64 WIRE, 3 minibatches per epoch (ragged tail)."""
    from wire_amd.modules import models, utils
    H = W = 32
    niters, maxpoints, learning_rate = 12, 400, 5e-3
    omega0, sigma0, hidden_layers, hidden_features = 7.0, 6.0, 2, 64
    im = _smooth_image(H, W)
    rng = np.random.default_rng(0)
    im_noisy = (im + 0.05 * rng.standard_normal(im.shape)).astype(np.float32)
    # -- wire_image_denoise.py:63-69
    x = torch.linspace(-1, 1, W)
    y = torch.linspace(-1, 1, H)
    X, Y = torch.meshgrid(x, y, indexing='xy')
    coords = torch.hstack((X.reshape(-1, 1), Y.reshape(-1, 1)))[None, ...]
    gt = torch.tensor(im).cuda().reshape(H * W, 3)[None, ...]
    gt_noisy = torch.tensor(im_noisy).cuda().reshape(H * W, 3)[None, ...]
    # -- :106-128
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=hidden_features,
                           hidden_layers=hidden_layers, first_omega_0=omega0, hidden_omega_0=omega0, scale=sigma0,
                           scale_tensor=[], pos_encode=False, sidelength=H)
    p_cpu = {k: v.detach().clone() for k, v in model.state_dict().items() if "omega_0" not in k and "scale_0" not in k}
    model.cuda()
    optim = torch.optim.Adam(lr=learning_rate * min(1, maxpoints / (H * W)), params=model.parameters())
    scheduler = LambdaLR(optim, lambda x: 0.1 ** min(x / niters, 1))
    mse_array = torch.zeros(niters, device="cuda")
    best_mse = torch.tensor(float("inf"))
    best_img = None
    rec = torch.zeros_like(gt)
    perms, losses = [], []
    g = torch.Generator().manual_seed(3)
    for epoch in range(niters):
        indices = torch.randperm(H * W, generator=g)
        perms.append(indices)
        for b_idx in range(0, H * W, maxpoints):
            b_indices = indices[b_idx:min(H * W, b_idx + maxpoints)]
            b_coords = coords[:, b_indices, ...].cuda()
            b_indices = b_indices.cuda()
            pixelvalues = model(b_coords)
            with torch.no_grad():
                rec[:, b_indices, :] = pixelvalues
            loss = ((pixelvalues - gt_noisy[:, b_indices, :]) ** 2).mean()
            optim.zero_grad()
            loss.backward()
            optim.step()
            losses.append(loss.item())
        with torch.no_grad():
            mse_array[epoch] = ((gt - rec) ** 2).mean().item()
        scheduler.step()
        imrec = rec[0, ...].reshape(H, W, 3).detach().cpu().numpy()
        if (mse_array[epoch] < best_mse) or (epoch == 0):
            best_mse = mse_array[epoch]
            best_img = imrec
    assert abs(model.net[0].scale_0.item() - sigma0) < 1e-6 and abs(model.net[0].omega_0.item() - omega0) < 1e-6
    # -- the same loop through the CPU restatement of the reference
    ref_losses, ref_rec, ref_best, _ = torch_ref.driver_loop(
        p_cpu, coords[0], torch.tensor(im_noisy).reshape(-1, 3), hidden_layers, omega0, omega0, sigma0,
        learning_rate * min(1, maxpoints / (H * W)), niters, maxpoints, 0.1, perms)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-3)
    # best-image bookkeeping of the reference is on (gt - rec): redo it for the CPU run with gt, not gt_noisy
    psnr_hip = utils.psnr(im, best_img)
    psnr_ref = utils.psnr(im, ref_rec.reshape(H, W, 3).numpy())
    print(f"image driver: PSNR HIP {psnr_hip:.3f} dB (best of {niters} epochs) vs CPU last epoch {psnr_ref:.3f} dB")
    assert abs(utils.psnr(im, imrec) - psnr_ref) < 0.1
    assert psnr_hip >= utils.psnr(im, imrec) - 1e-6


def test_occupancy_driver_loop():
    """wire_occupancy.py:107-172 on a 12 x 10 x 9 volume: D = 3, O = 1, ``model(b_coords[None, ...]).squeeze()[:,
    None]``, ``torch.nn.MSELoss``, ``loss.item()`` per minibatch, ``volutils.get_IoU`` per epoch (with its in-place
    binarisation of ``im_estim``), best volume by the last minibatch loss."""
    from wire_amd.modules import models, utils, volutils
    H, W, T = 12, 10, 9
    niters, maxpoints, learning_rate, mcubes_thres = 10, 400, 5e-3, 0.5
    omega0, sigma0, hidden_layers, hidden_features = 10.0, 8.0, 2, 64
    cz, cy, cx = np.meshgrid(np.linspace(-1, 1, T), np.linspace(-1, 1, W), np.linspace(-1, 1, H), indexing="ij")
    im = ((cx ** 2 + cy ** 2 + cz ** 2) < 0.6).astype(np.float32).transpose(2, 1, 0)     # [H, W, T] sphere
    imten = torch.tensor(im).cuda().reshape(H * W * T, 1)
    coords = utils.get_coords(H, W, T)
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=3, out_features=1, hidden_features=hidden_features,
                           hidden_layers=hidden_layers, first_omega_0=omega0, hidden_omega_0=omega0, scale=sigma0,
                           pos_encode=False, sidelength=max(H, W, T))
    p_cpu = {k: v.detach().clone() for k, v in model.state_dict().items() if "omega_0" not in k and "scale_0" not in k}
    model = model.cuda()
    optim = torch.optim.Adam(lr=learning_rate, params=model.parameters())
    scheduler = LambdaLR(optim, lambda x: 0.2 ** min(x / niters, 1))
    criterion = torch.nn.MSELoss()
    mse_array = np.zeros(niters)
    best_mse = float("inf")
    best_img = None
    im_estim = torch.zeros((H * W * T, 1), device="cuda")
    perms, losses = [], []
    g = torch.Generator().manual_seed(4)
    for idx in range(niters):
        indices = torch.randperm(H * W * T, generator=g)
        perms.append(indices)
        for b_idx in range(0, H * W * T, maxpoints):
            b_indices = indices[b_idx:min(H * W * T, b_idx + maxpoints)]
            b_coords = coords[b_indices, ...].cuda()
            b_indices = b_indices.cuda()
            pixelvalues = model(b_coords[None, ...]).squeeze()[:, None]
            with torch.no_grad():
                im_estim[b_indices, :] = pixelvalues
            loss = criterion(pixelvalues, imten[b_indices, :])
            optim.zero_grad()
            loss.backward()
            optim.step()
            lossval = loss.item()
            losses.append(lossval)
        mse_array[idx] = volutils.get_IoU(im_estim, imten, mcubes_thres)
        scheduler.step()
        if lossval < best_mse:
            best_mse = lossval
            best_img = copy.deepcopy(im_estim)
    assert set(np.unique(best_img.cpu().numpy())) <= {0.0, 1.0}      # the reference's in-place binarisation
    assert utils.count_parameters(model) == sum(p.numel() for p in model.parameters() if p.requires_grad)
    ref_losses, ref_rec, _, _ = torch_ref.driver_loop(
        p_cpu, coords, torch.tensor(im).reshape(-1, 1), hidden_layers, omega0, omega0, sigma0, learning_rate, niters,
        maxpoints, 0.2, perms, squeeze_occupancy=True)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-3, atol=1e-6)
    iou_ref = wo.iou(ref_rec.numpy(), im.reshape(-1, 1), mcubes_thres)
    print(f"occupancy driver: IoU HIP {mse_array[-1]:.4f} vs CPU restatement {iou_ref:.4f}")
    assert abs(mse_array[-1] - iou_ref) < 0.01


def test_device_iou_matches_reference_known_answer():
    """volutils.get_IoU on the device == the reference's value on its own vector (tests/golden/misc2.npz), including
    the in-place binarisation the caller observes; FusedTrainer.iou leaves its argument untouched."""
    from wire_amd.modules import volutils
    m2 = load_golden("misc2")
    pred = torch.tensor(m2["iou_pred"], device="cuda")
    gt = torch.tensor(m2["iou_gt"], device="cuda")
    val = float(volutils.get_IoU(pred, gt, float(m2["iou_thres"])))
    assert abs(val - float(m2["iou_val"])) < 1e-6
    np.testing.assert_array_equal(pred.cpu().numpy(), m2["iou_pred_after"])
    val2 = float(volutils.get_IoU((torch.tensor(m2["iou_pred"], device="cuda") > 0.7).float(), gt, None))
    assert abs(val2 - float(m2["iou_val_nothres"])) < 1e-6
    assert abs(float(volutils.get_IoU_batch(torch.tensor(m2["iou_pred"], device="cuda"), gt, 0.5, 1024)) -
               float(m2["iou_val"])) < 1e-6


def test_best_image_tracking_on_device():
    """FusedTrainer.update_best == the host logic of wire_image_denoise.py:176-178 over a sequence of metrics
    (ties, a first epoch that is not the best, improvements and regressions), with no .item()."""
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=32, hidden_layers=1).cuda()
    tr = FusedTrainer(model, (9, 7), torch.zeros(63, 3))
    rng = np.random.default_rng(2)
    metrics = [0.5, 0.7, 0.4, 0.4, 0.9, 0.1, 0.3]
    best_mse, best_img = float("inf"), None
    for epoch, mval in enumerate(metrics):
        img = rng.random((63, 3)).astype(np.float32)
        if (mval < best_mse) or (epoch == 0):
            best_mse, best_img = mval, img
        tr.update_best(torch.tensor([mval], device="cuda"), torch.tensor(img, device="cuda"), force=(epoch == 0))
    torch.cuda.synchronize()
    assert abs(float(tr.best_metric.item()) - best_mse) < 1e-7
    np.testing.assert_array_equal(tr.best_img.cpu().numpy(), best_img)


@pytest.mark.parametrize("tag,is_first", [("hid", False), ("first", True)])
def test_trainable_omega_scale_gradients(tag, is_first):
    """ComplexGaborLayer(trainable=True) (modules/wire.py:80-81): forward and every gradient -- omega_0, scale_0,
    weight, bias, input -- against the reference's fp64 autograd, yardstick = the reference's fp32 run."""
    from wire_amd.modules.wire import ComplexGaborLayer
    m2 = load_golden("misc2")
    fin = m2[f"tr_{tag}_x"].shape[1]
    layer = ComplexGaborLayer(fin, 40, is_first=is_first, omega0=9.0, sigma0=4.0, trainable=True)
    layer.load_state_dict({k.split(":", 1)[1]: torch.tensor(v) for k, v in m2.items() if k.startswith(f"tr_{tag}_p:")})
    layer = layer.cuda()
    assert layer.omega_0.requires_grad and layer.scale_0.requires_grad
    x = torch.tensor(m2[f"tr_{tag}_x"], device="cuda", requires_grad=not is_first)
    out = layer(x)
    out.backward(torch.tensor(m2[f"tr_{tag}_g"], device="cuda"))
    torch.cuda.synchronize()
    assert relmax(out.detach().cpu().numpy(), m2[f"tr_{tag}_out64"]) <= 1e-5
    got = {"g_omega": layer.omega_0.grad, "g_scale": layer.scale_0.grad, "g_W": layer.linear.weight.grad,
           "g_b": layer.linear.bias.grad}
    if not is_first:
        got["g_x"] = x.grad
    for k, v in got.items():
        ref64, ref32 = m2[f"tr_{tag}_{k}64"], m2[f"tr_{tag}_{k}"]
        if k in ("g_omega", "g_scale"):
            # one number each: a sum of 12 000 terms that cancels partly; the yardstick is the reference's fp32 error
            # relative to the sum of magnitudes it would have without cancellation (2e-5 backward bar on that scale)
            scale = max(abs(float(ref64[0])), 1.0)
            assert abs(float(v.item()) - float(ref64[0])) <= 2 * abs(float(ref32[0]) - float(ref64[0])) + 2e-5 * scale, k
        else:
            within_ref(relmax(v.cpu().numpy(), ref64), relmax(ref32, ref64), f"trainable {tag} {k}", floor=2e-6)


@pytest.mark.parametrize("tag,kind", [("siren", "siren"), ("gauss", "gauss"), ("relu", "relu"), ("wireL0", "wire")])
def test_constructor_paths_against_reference(tag, kind):
    """outermost_linear=False (modules/siren.py:81-84, gauss.py:63-66, relu.py:116-119: the last module is an
    activation layer) and a wire net with hidden_layers=0: state_dict keys, output and every gradient against the
    reference's fp64 run (tests/golden/misc2.npz)."""
    from wire_amd.modules import models
    m2 = load_golden("misc2")
    if kind == "wire":
        model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=64, hidden_layers=0,
                               first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0)
    else:
        model = models.get_INR(nonlin=kind, in_features=2, out_features=3, hidden_features=48, hidden_layers=2,
                               outermost_linear=False, first_omega_0=30.0, hidden_omega_0=30.0, scale=10.0)
    sd = {k.split(":", 1)[1]: torch.tensor(v) for k, v in m2.items() if k.startswith(f"ol_{tag}_p:")}
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model = model.cuda()
    coords = torch.tensor(m2["ol_coords"], device="cuda")
    target = torch.tensor(m2["ol_target"], device="cuda")
    y = model(coords)
    loss = ((y - target) ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    err_ref = relmax(m2[f"ol_{tag}_y"], m2[f"ol_{tag}_y64"])
    within_ref(relmax(y.detach().cpu().numpy(), m2[f"ol_{tag}_y64"]), err_ref, f"ctor {tag} y")
    for k, prm in model.named_parameters():
        if prm.grad is None:
            continue
        g64, g32 = m2[f"ol_{tag}_g64:{k}"], m2[f"ol_{tag}_g:{k}"]
        if g64.size <= 3:
            assert np.abs(prm.grad.cpu().numpy() - g64).max() <= (2 * err_ref + 1e-6) * max(np.abs(g64).max(), 0.1), k
        else:
            within_ref(relmax(prm.grad.cpu().numpy(), g64), relmax(g32, g64), f"ctor {tag} grad {k}", floor=2e-6)


def test_wide_net_unfused_final_stage():
    """O = 4 with K = 512 (P = 1024): the fused final stage would need 65 616 B of dynamic LDS (> the 64 KB a launch
    gets), so wire_train_fwd_bwd takes the unfused sequence; a step against the fp64 oracle."""
    from _util import params_np, wire_oracle_grads_chunked
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=2, out_features=4, hidden_features=725, hidden_layers=1,
                           first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0).cuda()
    assert model._arch["width"] == 512
    H, W = 24, 20
    g = torch.Generator().manual_seed(1)
    target = torch.rand(H * W, 4, generator=g)
    tr = FusedTrainer(model, (H, W), target, lr=0.0, keep_rec=True)
    loss = tr.step()
    torch.cuda.synchronize()
    P = params_np(model)
    coords = wo.image_coords(H, W)
    y64, l64, g64 = wire_oracle_grads_chunked(P, coords, target.numpy(), 1, 7.0, 7.0, 6.0, double=True)
    y32, l32, g32 = wire_oracle_grads_chunked(P, coords, target.numpy(), 1, 7.0, 7.0, 6.0, double=False)
    within_ref(relmax(tr.rec.cpu().numpy(), y64), relmax(y32, y64), "wide net y")
    assert abs(float(loss.item()) - l64) <= 1e-5 * l64
    flat = tr.flat_grad.cpu().numpy()
    names = [k for k in model.state_dict().keys() if "omega_0" not in k and "scale_0" not in k]
    for name, off in zip(names[:-1], tr.offsets[:-1]):
        ref = wo.as_real_pairs(g64[name]).astype(np.float64).ravel()
        ref32 = wo.as_real_pairs(g32[name]).astype(np.float64).ravel()
        within_ref(relmax(flat[off:off + ref.size], ref), relmax(ref32, ref), f"wide net grad {name}", floor=2e-6)


def test_second_backward_with_retain_graph():
    """The autograd node keeps its saved buffers: backward(retain_graph=True) twice accumulates 2 x the gradient."""
    from wire_amd.modules import models
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=48, hidden_layers=2,
                           first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0).cuda()
    x = torch.rand(1, 100, 2, device="cuda") * 2 - 1
    loss = (model(x) ** 2).mean()
    loss.backward(retain_graph=True)
    g1 = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    loss.backward()
    for a, p in zip(g1, [p for p in model.parameters() if p.grad is not None]):
        assert torch.allclose(p.grad, 2 * a, rtol=1e-6, atol=0)


@pytest.mark.parametrize("tag,is_first", [("hid", False), ("first", True)])
def test_trainable_omega_scale_gradients_2d(tag, is_first):
    """ComplexGaborLayer2D(trainable=True) (modules/wire2d.py:42-43): forward and every gradient against the
    reference's fp64 autograd (tests/golden/misc3.npz)."""
    from wire_amd.modules.wire2d import ComplexGaborLayer2D
    m3 = load_golden("misc3")
    fin = m3[f"tr2d_{tag}_x"].shape[1]
    layer = ComplexGaborLayer2D(fin, 36, is_first=is_first, omega0=6.0, sigma0=3.0, trainable=True)
    layer.load_state_dict({k.split(":", 1)[1]: torch.tensor(v) for k, v in m3.items() if k.startswith(f"tr2d_{tag}_p:")})
    layer = layer.cuda()
    x = torch.tensor(m3[f"tr2d_{tag}_x"], device="cuda", requires_grad=not is_first)
    out = layer(x)
    out.backward(torch.tensor(m3[f"tr2d_{tag}_g"], device="cuda"))
    torch.cuda.synchronize()
    assert relmax(out.detach().cpu().numpy(), m3[f"tr2d_{tag}_out64"]) <= 1e-5
    got = {"g_omega": layer.omega_0.grad, "g_scale": layer.scale_0.grad, "g_W": layer.linear.weight.grad,
           "g_b": layer.linear.bias.grad, "g_V": layer.scale_orth.weight.grad, "g_c": layer.scale_orth.bias.grad}
    if not is_first:
        got["g_x"] = x.grad
    for k, v in got.items():
        ref64, ref32 = m3[f"tr2d_{tag}_{k}64"], m3[f"tr2d_{tag}_{k}"]
        if k in ("g_omega", "g_scale"):
            scale = max(abs(float(ref64[0])), 1.0)
            assert abs(float(v.item()) - float(ref64[0])) <= 2 * abs(float(ref32[0]) - float(ref64[0])) + 2e-5 * scale, k
        else:
            within_ref(relmax(v.cpu().numpy(), ref64), relmax(ref32, ref64), f"trainable2d {tag} {k}", floor=2e-6)


def test_mesh_export_occupancy_query():
    """The dense query of export_mesh (modules/volutils.py:113-133: batches of coordinates -> torch.sigmoid(model(c)) ->
    the occupancy cube handed to marching cubes), through the drop-in module and through FusedTrainer.render."""
    from wire_amd.modules import models, utils, volutils
    from wire_amd.trainer import FusedTrainer
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=3, out_features=1, hidden_features=48, hidden_layers=2,
                           first_omega_0=10.0, hidden_omega_0=10.0, scale=8.0).cuda()
    res = 11
    coords = utils.get_coords(res, res, res)
    cube = volutils.query_occupancy(coords, res, model, batchsize=500)
    assert cube.shape == (res, res, res) and cube.dtype == np.float32
    P64 = wo.cast_params({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()
                          if "omega_0" not in k and "scale_0" not in k}, True)
    y64 = wo.wire_forward(P64, coords.numpy().astype(np.float64), 2, 10.0, 10.0, 8.0)
    ref = 1.0 / (1.0 + np.exp(-y64))
    assert np.abs(cube.reshape(-1, 1) - ref).max() <= 2e-5
    tr = FusedTrainer(model, (res, res, res), torch.zeros(res ** 3, 1), coords_style="numpy")
    cube2 = tr.render(tile=400, sigmoid=True).cpu().numpy()
    assert np.abs(cube2 - ref).max() <= 2e-5


def test_index_range_check_is_opt_in(monkeypatch):
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=32, hidden_layers=1).cuda()
    tr = FusedTrainer(model, (8, 8), torch.zeros(64, 3))
    monkeypatch.setenv("WIRE_CHECK_INDICES", "1")
    with pytest.raises(ValueError):
        tr.step(torch.tensor([0, 5, 64], device="cuda"))           # 64 is outside the 8 x 8 grid
    tr.step(torch.tensor([0, 5, 63], device="cuda"))


def test_get_layer_outputs_matches_reference_montages():
    """``utils.get_layer_outputs`` (modules/utils.py:229-288): per-layer activation montages through the
    per-layer HIP entry points against the montages the reference's own function produced for the same net
    (tests/golden/make_layer_outputs_golden.py).  The montage is min-max normalised per filter, so the comparison is
    absolute on [0, 1]."""
    import os
    from _util import GOLDEN, checksum, params_np
    from wire_amd.modules import models, utils
    z = np.load(os.path.join(GOLDEN, "layer_outputs.npz"), allow_pickle=False)
    H, W = int(z["H"]), int(z["W"])
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=64, hidden_layers=2,
                           first_omega_0=5.0, hidden_omega_0=5.0, scale=5.0)
    for k, v in params_np(model).items():
        np.testing.assert_allclose(checksum(v), z["sd_checksum__" + k], rtol=1e-12, atol=1e-12)
    model = model.to("cuda")
    coords = torch.tensor(z["coords"]).to("cuda")
    for tag, imag in (("re", False), ("im", True)):
        got = utils.get_layer_outputs(model, coords, (H, W), nfilters_vis=9, get_imag=imag)
        assert len(got) == 3
        for i, m in enumerate(got):
            ref = z[f"montage_{tag}_{i}"]
            assert m.shape == ref.shape
            assert np.abs(m - ref).max() <= 2e-4, f"{tag} layer {i}: {np.abs(m - ref).max():.2e}"


def test_grad_ready_hook_announces_every_tensor_once_in_backward_order():
    """wire_train_fwd_bwd_hooked (include/wire_hip.h): the callback runs on the host, inside the call, once per group of
    parameter tensors whose gradient is final on the stream -- the final linear layer first (modules/wire.py:156-157), the
    hidden ComplexGaborLayers from the last to the first, the first layer last (the order the autograd backward of
    modules/wire.py:161-167 produces them) -- and the gradients equal those of the plain call."""
    import ctypes as C
    from wire_amd import _lib
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    torch.manual_seed(0)
    model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=91, hidden_layers=3,
                           first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0).cuda()
    g = torch.Generator().manual_seed(1)
    tr = FusedTrainer(model, (40, 33), torch.rand(40 * 33, 3, generator=g), lr=0.0)
    perm = torch.randperm(40 * 33, generator=g).cuda()
    tr.step(perm)
    torch.cuda.synchronize()
    plain = tr.flat_grad.clone()
    calls = []
    cb = _lib.GRAD_READY_FN(lambda user, first, n: calls.append((first, n)))
    L, d = _lib.lib(), C.byref(tr.desc)
    stream = torch.cuda.current_stream().cuda_stream
    n = 40 * 33
    tr.gbuf[0].zero_()
    _lib.check(L.wire_coords_from_index(stream, perm.data_ptr(), 0, n, tr.tx.data_ptr(), tr.grid[1], tr.ty.data_ptr(),
                                        tr.grid[0], None, 1, tr.coords.data_ptr()), "coords")
    _lib.check(L.wire_train_fwd_bwd_hooked(
        stream, d, tr.packed.data_ptr(), tr.coords.data_ptr(), n, tr.target.data_ptr(), perm.data_ptr(), 0, 1.0,
        tr.y.data_ptr(), tr.gy.data_ptr(), tr.gbuf[0].data_ptr() + 4 * tr.count, None, tr.partial.data_ptr(),
        tr.act.data_ptr(), tr.act_bytes, tr.scratch.data_ptr(), tr.scr_bytes, tr.grad_ptrs[0], cb, None), "hooked")
    torch.cuda.synchronize()
    nt = len(tr.offsets)                                   # 2 tensors per layer: 1 first + 3 hidden + 1 final = 10
    assert nt == 10
    assert calls == [(8, 2), (6, 2), (4, 2), (2, 2), (0, 2)]
    assert calls == list(tr._ready_order())
    assert torch.equal(tr.flat_grad, plain)
