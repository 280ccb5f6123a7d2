#!/usr/bin/env python3
"""Golden vector of the real-image quality gate (BASELINE.json configs[0]; build container only).

    python3 tests/golden/make_psnr_golden.py [name [niters]]     # writes tests/golden/<name>.npz; names: SCHEDULES below
                                                                 # (psnr_parrot_cfg1, 100 epochs: about an hour of CPU)

Runs the loop of the reference's wire_image_denoise.py:104-178 with the REFERENCE's own model (``modules.wire.INR``
imported from /root/reference, CPU, fp32) on the RGB channels of the image the reference ships,
``data_noisy/parrot_noisy_T30.0_snr2.png`` (678 x 1020; the clean ``data/parrot.png`` is git-ignored upstream, so the noisy
image is the fit target and PSNR is measured against it):

    model = get_INR('wire', in_features=2, out_features=3, hidden_features=128, hidden_layers=2,      # -> K = 90
                    first_omega_0=7, hidden_omega_0=7, scale=6)                 wire_image_denoise.py:40,83,106-118
    Adam(lr = 5e-3 * min(1, maxpoints / (H W))), maxpoints = 256 * 256             :50,123-125
    LambdaLR(0.1 ** min(epoch / niters, 1))                                         :128
    per epoch: indices = torch.randperm(H W); 11 minibatches of <= 65 536 rows; rec[b] = model(coords[b]);
               loss = ((pix - gt[b]) ** 2).mean(); zero_grad / backward / step      :142-157
    after the last epoch: utils.psnr(gt, rec)                                       modules/utils.py:67-82

for a fixed short schedule (NITERS epochs; the schedule's ``niters`` is NITERS too, so the learning rate decays over the
run).  torch.manual_seed(0) is set once, before the model is built: the parameters AND the per-epoch permutations follow
from it, so the GPU test regenerates both with the same calls instead of storing 50 MB of indices.

Stored (data only): the decoded image as uint8 RGB, per-step losses, per-epoch MSE of ``rec`` against the target, the
final PSNR, checksums of the initial state_dict and of the final ``rec`` -- and the same trajectory from an fp64 twin of
the loop (the oracle's eager restatement, oracle/torch_ref.py, float64): |fp32 - fp64| is the yardstick for "how far may
two correct fp32 implementations drift apart on this schedule".
"""
import os
import sys
import time
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
MAXPOINTS = 256 * 256
LR = 5e-3
NOISY = "data_noisy/parrot_noisy_T30.0_snr2.png"
PUBLISHED = "multiscale_results/denoise/T30.0_SNR2/Final/WIRE_s8_o7_LR5e3_E2000_2/Output_img.png"
# name -> (niters, hidden_features, hidden_layers, omega0, sigma0, image)
SCHEDULES = {
    # BASELINE.json configs[0]: 2 hidden x 128 features (K = 90), omega0 = 7, sigma0 = 6; round 4: 100 epochs = 1100 steps.
    # The fit target is the NOISY image (the clean data/parrot.png is git-ignored upstream), so the PSNR saturates at the
    # noise floor of that target (about 17 dB).
    "psnr_parrot_cfg1": (100, 128, 2, 7.0, 6.0, NOISY),
    # the net of the reference's published denoise result (29.70 dB, 91 587 parameters: .../WIRE_s8_o7_LR5e3_E2000_2/
    # metrics_table.md:3): 2 hidden x 300 features (K = 212), omega0 = 7, sigma0 = 8, lr 5e-3, on a shorter schedule.  Fit
    # target: the RECONSTRUCTION that run stored (Output_img.png, 678 x 1020, the one clean parrot image the reference
    # holds) -- an image this very architecture produced, so the fit reaches the >= 25 dB regime the published number
    # lives in within tens of epochs.
    "psnr_parrot_pub2x300": (30, 300, 2, 7.0, 8.0, PUBLISHED),
}
NAME = sys.argv[1] if len(sys.argv) > 1 else "psnr_parrot_cfg1"
NITERS, HIDDEN_FEATURES, LAYERS, OMEGA0, SIGMA0, IMAGE = SCHEDULES[NAME]
if len(sys.argv) > 2:
    NITERS = int(sys.argv[2])

sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
from modules import utils, wire  # noqa: E402  (the reference's own modules)

torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))


def checksum(a):
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = np.stack([a.real, a.imag], -1)
    a = a.astype(np.float64).ravel()
    w = np.cos(np.arange(a.size) * 0.37) + 0.5
    return np.array([a.sum(), np.abs(a).sum(), (a * w).sum()], np.float64)


def load_image():
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(REF, IMAGE)))
    assert img.dtype == np.uint8 and img.ndim == 3
    return np.ascontiguousarray(img[..., :3])                  # RGB of RGBA


def main():
    u8 = load_image()
    H, W, _ = u8.shape
    im = np.divide(u8, 255, dtype=np.float32)                   # what plt.imread returns for an 8-bit PNG
    x = torch.linspace(-1, 1, W)
    y = torch.linspace(-1, 1, H)
    X, Y = torch.meshgrid(x, y, indexing="xy")
    coords = torch.hstack((X.reshape(-1, 1), Y.reshape(-1, 1)))[None, ...]
    gt = torch.tensor(im).reshape(H * W, 3)[None, ...]

    torch.manual_seed(0)
    model = wire.INR(2, HIDDEN_FEATURES, 0, LAYERS, 3, True, OMEGA0, OMEGA0, SIGMA0)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    lr0 = LR * min(1, MAXPOINTS / (H * W))
    optim = torch.optim.Adam(lr=lr0, params=model.parameters())
    scheduler = torch.optim.lr_scheduler.LambdaLR(optim, lambda e: 0.1 ** min(e / NITERS, 1))
    rec = torch.zeros_like(gt)
    losses, mse_epoch, perms = [], [], []
    t0 = time.time()
    for epoch in range(NITERS):
        indices = torch.randperm(H * W)
        perms.append(indices.clone())
        for b_idx in range(0, H * W, MAXPOINTS):
            b_indices = indices[b_idx:min(H * W, b_idx + MAXPOINTS)]
            pixelvalues = model(coords[:, b_indices, ...])
            with torch.no_grad():
                rec[:, b_indices, :] = pixelvalues
            loss = ((pixelvalues - gt[:, b_indices, :]) ** 2).mean()
            optim.zero_grad()
            loss.backward()
            optim.step()
            losses.append(float(loss.item()))
        with torch.no_grad():
            mse_epoch.append(float(((gt - rec) ** 2).mean().item()))
        scheduler.step()
        print(f"fp32 reference epoch {epoch}: mse {mse_epoch[-1]:.6f}  ({time.time() - t0:.0f} s)", flush=True)
    psnr32 = float(utils.psnr(im, rec[0].reshape(H, W, 3).numpy()))

    # ---- fp64 twin of the same loop (oracle restatement; same initial parameters, same permutations)
    from oracle import torch_ref
    p64 = {k: (v.to(torch.cdouble) if v.is_complex() else v.to(torch.double)).clone().requires_grad_(True)
           for k, v in sd0.items() if "omega_0" not in k and "scale_0" not in k}
    opt64 = torch.optim.Adam(lr=lr0, params=list(p64.values()))
    sch64 = torch.optim.lr_scheduler.LambdaLR(opt64, lambda e: 0.1 ** min(e / NITERS, 1))
    c64, gt64 = coords.double(), gt.double()
    rec64 = torch.zeros_like(gt64)
    losses64, mse64 = [], []
    for epoch in range(NITERS):
        indices = perms[epoch]
        for b_idx in range(0, H * W, MAXPOINTS):
            b = indices[b_idx:min(H * W, b_idx + MAXPOINTS)]
            pix = torch_ref.wire_forward(p64, c64[:, b, :], LAYERS, OMEGA0, OMEGA0, SIGMA0)
            with torch.no_grad():
                rec64[:, b, :] = pix
            loss = ((pix - gt64[:, b, :]) ** 2).mean()
            opt64.zero_grad()
            loss.backward()
            opt64.step()
            losses64.append(float(loss.item()))
        with torch.no_grad():
            mse64.append(float(((gt64 - rec64) ** 2).mean().item()))
        sch64.step()
        print(f"fp64 twin epoch {epoch}: mse {mse64[-1]:.6f}  ({time.time() - t0:.0f} s)", flush=True)
    psnr64 = float(utils.psnr(im.astype(np.float64), rec64[0].reshape(H, W, 3).numpy()))

    out = dict(
        image_u8=u8, niters=np.int64(NITERS), maxpoints=np.int64(MAXPOINTS), hidden_features=np.int64(HIDDEN_FEATURES),
        hidden_layers=np.int64(LAYERS), omega0=np.float64(OMEGA0), sigma0=np.float64(SIGMA0), lr=np.float64(LR),
        seed=np.int64(0), torch_version=np.array(torch.__version__), image_path=np.array(IMAGE),
        losses=np.array(losses, np.float64), mse_epoch=np.array(mse_epoch, np.float64), psnr=np.float64(psnr32),
        losses64=np.array(losses64, np.float64), mse_epoch64=np.array(mse64, np.float64), psnr64=np.float64(psnr64),
        perm_first8=np.stack([q[:8].numpy() for q in perms]), rec_checksum=checksum(rec.numpy()),
    )
    for k, v in sd0.items():
        if "omega_0" not in k and "scale_0" not in k:
            out["sd0_checksum__" + k] = checksum(v.numpy())
    np.savez_compressed(os.path.join(OUT, NAME + ".npz"), **out)
    print(f"PSNR fp32 reference {psnr32:.4f} dB, fp64 twin {psnr64:.4f} dB; wrote {NAME}.npz")


if __name__ == "__main__":
    main()
