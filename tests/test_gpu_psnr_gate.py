"""Quality gate of BASELINE.json's north_star ("PSNR within 0.1 dB of reference") on the images the reference ships.

tests/golden/psnr_parrot_*.npz (tests/golden/make_psnr_golden.py, build container) hold an 8-bit RGB image of the reference
and the trajectory of the REFERENCE's own model (``modules.wire.INR`` imported from the reference, CPU fp32) through the
loop of wire_image_denoise.py:104-178 -- Adam lr = 5e-3 min(1, maxpoints / HW), LambdaLR 0.1^(epoch / niters), maxpoints =
65 536 (11 minibatches per epoch of the 678 x 1020 image), torch.manual_seed(0) once before the model is built,
``torch.randperm(H W)`` per epoch -- plus the same loop in fp64 (the yardstick for what two correct fp32 implementations may
differ by):

 * psnr_parrot_cfg1: BASELINE.json configs[0] -- 2 hidden layers x 128 features (K = 90), omega0 = 7, sigma0 = 6 -- on
   ``data_noisy/parrot_noisy_T30.0_snr2.png`` for 100 epochs = 1100 optimizer steps (round 3: 10 epochs).  The fit target is
   the noisy image itself (the clean one is git-ignored upstream), so the PSNR saturates at that target's noise floor,
   17.26 dB;
 * psnr_parrot_pub2x300: the net of the reference's published denoise result (2 x 300 -> K = 212, omega0 = 7, sigma0 = 8,
   91 587 parameters, 29.70 dB: multiscale_results/denoise/T30.0_SNR2/Final/WIRE_s8_o7_LR5e3_E2000_2/metrics_table.md:3)
   fitted for 30 epochs = 330 steps to the reconstruction that run stored (``Output_img.png`` beside the table: the one
   clean parrot image the reference holds) -- the regime of the published number, about 25 dB after this schedule.

Here the same loop runs on the MI355X through ``FusedTrainer.step(indices)`` with the regenerated permutations: the
per-minibatch losses must follow the reference's, the per-epoch MSE of ``rec`` must follow it, and the final
``utils.psnr`` (modules/utils.py:67-82) must agree within 0.1 dB.
"""
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, checksum, params_np
from oracle import wire_oracle as wo

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("fixture", ["psnr_parrot_cfg1", "psnr_parrot_pub2x300"])
def test_parrot_denoise_schedule_psnr_within_0p1_db_of_reference(fixture):
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    z = np.load(os.path.join(GOLDEN, fixture + ".npz"), allow_pickle=False)
    u8 = z["image_u8"]
    H, W, _ = u8.shape
    assert (H, W) == (678, 1020)
    im = np.divide(u8, 255, dtype=np.float32)
    niters, maxpoints = int(z["niters"]), int(z["maxpoints"])
    torch.manual_seed(int(z["seed"]))
    model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=int(z["hidden_features"]),
                           hidden_layers=int(z["hidden_layers"]), first_omega_0=float(z["omega0"]),
                           hidden_omega_0=float(z["omega0"]), scale=float(z["sigma0"]))
    # same construction order -> the reference's initial parameters, bit for bit
    for k, v in params_np(model).items():
        np.testing.assert_allclose(checksum(v), z["sd0_checksum__" + k], rtol=1e-12, atol=1e-12)
    model = model.to(DEV)
    lr0 = float(z["lr"]) * min(1, maxpoints / (H * W))
    tr = FusedTrainer(model, (H, W), torch.tensor(im).reshape(H * W, 3), lr=lr0, niters=niters, keep_rec=True)
    ref_loss, ref_mse = z["losses"], z["mse_epoch"]
    l64, mse64 = z["losses64"], z["mse_epoch64"]
    losses, mse_epoch = [], []
    tgt = tr.target
    for epoch in range(niters):
        indices = torch.randperm(H * W)                      # CPU generator, as wire_image_denoise.py:142
        assert np.array_equal(indices[:8].numpy(), z["perm_first8"][epoch])
        idx = indices.to(DEV)
        for b_idx in range(0, H * W, maxpoints):
            losses.append(tr.step(idx[b_idx:min(H * W, b_idx + maxpoints)].contiguous()))
        mse_epoch.append(((tgt - tr.rec) ** 2).mean())      # device scalars; one sync at the end
        tr.scheduler_step()
    torch.cuda.synchronize()
    losses = np.array([float(x.item()) for x in losses])
    mse_epoch = np.array([float(x.item()) for x in mse_epoch])
    rec = tr.rec.cpu().numpy().reshape(H, W, 3)
    psnr = wo.psnr(im, rec)
    # the reference's own fp32-vs-fp64 drift on this schedule is the yardstick (+ a floor for the last fp32 bits)
    drift_ref = np.abs(ref_loss - l64) / l64
    drift = np.abs(losses - l64) / l64
    print(f"{fixture} ({niters} epochs, {len(losses)} steps): PSNR build {psnr:.4f} dB  reference {float(z['psnr']):.4f} dB  (fp64 twin {float(z['psnr64']):.4f});  "
          f"loss drift vs fp64: build max {drift.max():.2e}, reference max {drift_ref.max():.2e}")
    assert abs(psnr - float(z["psnr"])) < 0.1
    assert np.all(np.abs(losses - ref_loss) <= 1e-3 * ref_loss)            # the trajectories stay together ...
    assert drift.max() <= 4 * drift_ref.max() + 2e-5                        # ... and within the fp32 yardstick of fp64
    assert np.all(np.abs(mse_epoch - ref_mse) <= 1e-3 * ref_mse)
    assert np.all(np.abs(mse_epoch - mse64) <= 4 * np.abs(ref_mse - mse64) + 1e-4 * mse64)
