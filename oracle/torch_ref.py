"""Eager-PyTorch CPU restatement of the WIRE hot path -- TEST INFRASTRUCTURE.

Same role and same rules as ``oracle/wire_oracle.py`` (only tests, smoke() and
bench.py's cpu_baseline leg may import this).  It expresses the reference's
forward with the same ATen ops the reference dispatches on CPU (complex
``F.linear`` + ``torch.exp``; modules/wire.py:88-93, :161-167) but as plain
functions over a dict of tensors, so autograd produces the reference's
backward and ``torch.optim.Adam`` its update.  It is what bench.py times as the
"reference CPU path" (``cpu_baseline.kind = "port"``): the reference's own
files cannot travel to the GPU box.

Pinned by tests/test_oracle_golden.py against vectors generated from the
reference modules (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

TParams = Dict[str, torch.Tensor]


def gabor(lin: torch.Tensor, omega0: float, scale0: float) -> torch.Tensor:
    # modules/wire.py:90-93
    omega = omega0 * lin
    scale = scale0 * lin
    return torch.exp(1j * omega - scale.abs().square())


def wire_forward(p: TParams, coords: torch.Tensor, hidden_layers: int,
                 first_omega0: float, hidden_omega0: float, scale0: float,
                 keep: bool = False):
    acts = []
    h = gabor(F.linear(coords, p["net.0.linear.weight"], p["net.0.linear.bias"]),
              first_omega0, scale0)
    acts.append(h)
    for l in range(1, hidden_layers + 1):
        h = gabor(F.linear(h, p[f"net.{l}.linear.weight"], p[f"net.{l}.linear.bias"]),
                  hidden_omega0, scale0)
        acts.append(h)
    y = F.linear(h, p[f"net.{hidden_layers + 1}.weight"],
                 p[f"net.{hidden_layers + 1}.bias"]).real
    return (y, acts) if keep else y


def gabor2d(lin, sy, omega0, scale0):
    # modules/wire2d.py:56-67
    freq = torch.exp(1j * omega0 * lin)
    arg = lin.abs().square() + sy.abs().square()
    return freq * torch.exp(-scale0 * scale0 * arg)


def wire2d_forward(p: TParams, coords, hidden_layers, first_omega0,
                   hidden_omega0, scale0):
    h = coords
    for l in range(hidden_layers + 1):
        lin = F.linear(h, p[f"net.{l}.linear.weight"], p[f"net.{l}.linear.bias"])
        sy = F.linear(h, p[f"net.{l}.scale_orth.weight"], p[f"net.{l}.scale_orth.bias"])
        h = gabor2d(lin, sy, first_omega0 if l == 0 else hidden_omega0, scale0)
    return F.linear(h, p[f"net.{hidden_layers + 1}.weight"],
                    p[f"net.{hidden_layers + 1}.bias"]).real


def posenc(coords: torch.Tensor, num_frequencies: int) -> torch.Tensor:
    # modules/relu.py:62-75
    cols = [coords]
    for i in range(num_frequencies):
        for j in range(coords.shape[-1]):
            c = coords[..., j]
            cols.append(torch.sin((2 ** i) * math.pi * c).unsqueeze(-1))
            cols.append(torch.cos((2 ** i) * math.pi * c).unsqueeze(-1))
    return torch.cat(cols, dim=-1)


def realnet_forward(kind: str, p: TParams, coords, hidden_layers, first_omega0,
                    hidden_omega0, scale0, num_frequencies=None):
    h = coords if num_frequencies is None else posenc(coords, num_frequencies)
    for l in range(hidden_layers + 1):
        lin = F.linear(h, p[f"net.{l}.linear.weight"], p[f"net.{l}.linear.bias"])
        om = first_omega0 if l == 0 else hidden_omega0
        if kind == "siren":
            h = torch.sin(om * lin)                  # modules/siren.py:48-49
        elif kind == "gauss":
            h = torch.exp(-(scale0 * lin) ** 2)      # modules/gauss.py:27-28
        elif kind == "relu":
            h = F.relu(lin)                          # modules/relu.py:28-29
        else:
            raise ValueError(kind)
    return F.linear(h, p[f"net.{hidden_layers + 1}.weight"],
                    p[f"net.{hidden_layers + 1}.bias"])


def init_wire_params(D: int, hidden_features: int, L: int, O: int,
                     seed: int = 0) -> TParams:
    """nn.Linear default init in the reference's construction order
    (modules/wire.py:127-157) under torch.manual_seed(seed): reproduces the
    reference's initial state_dict bit-for-bit (checked by the golden test)."""
    K = int(hidden_features / math.sqrt(2))
    torch.manual_seed(seed)
    p: TParams = {}
    lin = torch.nn.Linear(D, K, dtype=torch.float)
    p["net.0.linear.weight"], p["net.0.linear.bias"] = lin.weight.detach(), lin.bias.detach()
    for l in range(1, L + 1):
        lin = torch.nn.Linear(K, K, dtype=torch.cfloat)
        p[f"net.{l}.linear.weight"], p[f"net.{l}.linear.bias"] = lin.weight.detach(), lin.bias.detach()
    lin = torch.nn.Linear(K, O, dtype=torch.cfloat)
    p[f"net.{L + 1}.weight"], p[f"net.{L + 1}.bias"] = lin.weight.detach(), lin.bias.detach()
    return p


def train_steps(p: TParams, coords: torch.Tensor, target: torch.Tensor,
                hidden_layers: int, first_omega0: float, hidden_omega0: float,
                scale0: float, lr: float, steps: int, niters: int = 2000,
                gamma: float = 0.1):
    """Full-batch restatement of the loop at wire_image_denoise.py:141-169:
    MSE -> zero_grad/backward/Adam.step, LambdaLR stepped per epoch.
    Returns (losses, params-after)."""
    params = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(lr=lr, params=list(params.values()))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: gamma ** min(x / niters, 1))
    losses = []
    for _ in range(steps):
        y = wire_forward(params, coords, hidden_layers, first_omega0,
                         hidden_omega0, scale0)
        loss = ((y - target) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        losses.append(float(loss.detach()))
    return losses, {k: v.detach() for k, v in params.items()}


def driver_loop(p: TParams, coords: torch.Tensor, gt: torch.Tensor, hidden_layers: int, first_omega0: float,
                hidden_omega0: float, scale0: float, lr: float, niters: int, maxpoints: int, gamma: float,
                perms, squeeze_occupancy: bool = False):
    """The reference drivers' training loop on CPU (wire_image_denoise.py:141-178 / wire_occupancy.py:136-172):
    per epoch ``indices = perms[epoch]`` (the driver's ``torch.randperm``), minibatches of ``maxpoints`` rows,
    ``rec[b_indices] = pixelvalues``, MSE, zero_grad / backward / Adam.step, LambdaLR(gamma ** min(x / niters, 1))
    stepped per epoch, best-so-far bookkeeping on the epoch's reconstruction error (image driver) or on the last
    minibatch loss (occupancy driver).  coords [N, D], gt [N, O].  Returns (per-step losses, rec, best_img,
    params-after)."""
    params = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(lr=lr, params=list(params.values()))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: gamma ** min(x / niters, 1))
    N = coords.shape[0]
    rec = torch.zeros_like(gt)
    losses, best, best_img = [], float("inf"), None
    for epoch in range(niters):
        indices = perms[epoch]
        for b in range(0, N, maxpoints):
            bi = indices[b:min(N, b + maxpoints)]
            y = wire_forward(params, coords[bi][None], hidden_layers, first_omega0, hidden_omega0, scale0)[0]
            with torch.no_grad():
                rec[bi] = y
            loss = ((y - gt[bi]) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        sched.step()
        with torch.no_grad():
            metric = losses[-1] if squeeze_occupancy else float(((gt - rec) ** 2).mean())
        if metric < best or (epoch == 0 and not squeeze_occupancy):
            best, best_img = metric, rec.clone()
    return losses, rec, best_img, {k: v.detach() for k, v in params.items()}


def radon(imten: torch.Tensor, angles_deg: torch.Tensor) -> torch.Tensor:
    """lin_inverse.radon (modules/lin_inverse.py:19-40) for one image: imten [H, W] -> sinogram [nangles, W].

    The reference rotates with ``kornia.geometry.rotate`` (pinned kornia==0.6.5, requirements.txt:6; kornia is
    not installed here).  Its published algorithm, restated with the ATen ops it dispatches:
    ``get_rotation_matrix2d`` about ((W-1)/2, (H-1)/2), positive angle counter-clockwise; ``warp_affine`` =
    ``normalize_homography`` with (W-1, H-1), inverse, ``F.affine_grid`` + ``F.grid_sample`` (bilinear, zeros,
    align_corners=True); then ``.sum`` over the rows.  Pinned by the gt -> sinogram pair the reference stores in
    multiscale_results/ct/Original/WIRE_s12_o8_LR5e3_E2000_1/info.mat (tests/golden/ct_pair.npz): max |diff|
    1e-4 on values up to 47 (the stored run was fp32 on the authors' GPU)."""
    H, W = imten.shape
    dt = imten.dtype
    a = torch.deg2rad(angles_deg.to(dt))
    c, s = torch.cos(a), torch.sin(a)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    M = torch.zeros(len(a), 3, 3, dtype=dt)
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = c, s, (1 - c) * cx - s * cy
    M[:, 1, 0], M[:, 1, 1], M[:, 1, 2] = -s, c, s * cx + (1 - c) * cy
    M[:, 2, 2] = 1
    N = torch.tensor([[2.0 / (W - 1), 0, -1], [0, 2.0 / (H - 1), -1], [0, 0, 1]], dtype=dt)
    Mn = N @ M @ torch.linalg.inv(N)
    theta = torch.linalg.inv(Mn)[:, :2, :]
    grid = F.affine_grid(theta, [len(a), 1, H, W], align_corners=True)
    rot = F.grid_sample(imten[None, None].expand(len(a), 1, H, W), grid, mode="bilinear", padding_mode="zeros",
                        align_corners=True)
    return rot.sum(2)[:, 0, :]
