"""CPU oracle for the WIRE INR hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a numpy restatement of the reference's algorithm for the path
SURVEY.md section 8(a) lists.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it; nothing under
``wire_amd/`` does (the product path raises when the HIP library is missing).

Parity status: the reference has no tests of its own.  The
oracle is pinned by golden vectors generated in the build container by
importing the reference modules themselves (``tests/golden/make_golden.py``,
torch 2.10.0 CPU) -- see ``tests/test_oracle_golden.py``.  The third-party
arithmetic under the reference (ATen GEMM / exp) is therefore pinned only by
those vectors, as SURVEY.md section 8(c) records.

Every function is written closed-form (forward AND hand-derived backward) and
is dtype-generic: run it in float32/complex64 to mirror the reference's CPU
path, or in float64/complex128 as the "truth" the fp32 paths are compared to.

Parameter containers are plain dicts keyed by the reference's ``state_dict``
names (``net.{i}.linear.weight`` ...), so fixtures load straight in.

Reference citations are into /root/reference (not present on the GPU box).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np

Params = Dict[str, np.ndarray]


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _real_dtype(double: bool):
    return np.float64 if double else np.float32


def _cplx_dtype(double: bool):
    return np.complex128 if double else np.complex64


def cast_params(params: Params, double: bool) -> Params:
    """Cast a state-dict-style parameter dict to fp32/c64 or fp64/c128."""
    out = {}
    for k, v in params.items():
        v = np.asarray(v)
        if np.iscomplexobj(v):
            out[k] = v.astype(_cplx_dtype(double))
        else:
            out[k] = v.astype(_real_dtype(double))
    return out


def complex_width(hidden_features: int) -> int:
    """modules/wire.py:119 -- hidden_features = int(hidden_features / sqrt(2))."""
    return int(hidden_features / np.sqrt(2))


def wire2d_width(hidden_features: int) -> int:
    """modules/wire2d.py:92 -- hidden_features = int(hidden_features / 2)."""
    return int(hidden_features / 2)


# --------------------------------------------------------------------------
# WIRE (complex Gabor) -- modules/wire.py:88-93, :161-167
# --------------------------------------------------------------------------
def gabor_act(lin: np.ndarray, omega0, scale0) -> np.ndarray:
    """ComplexGaborLayer.forward's nonlinearity (modules/wire.py:90-93):
    omega = w0*lin; scale = s0*lin; exp(1j*omega - |scale|^2).
    Works for real ``lin`` (first layer) and complex ``lin`` (hidden)."""
    omega = omega0 * lin
    scale = scale0 * lin
    return np.exp(1j * omega - np.square(np.abs(scale)))


def gabor_act_grad(g_out: np.ndarray, lin: np.ndarray, out: np.ndarray,
                   omega0, scale0) -> np.ndarray:
    """Backward of gabor_act in PyTorch's complex-grad convention
    (grad = dL/dRe + j dL/dIm).  With c = conj(out)*g, P = Re c:
        complex lin:  g_lin = -2 s0^2 P lin - j w0 c
        real lin:     g_u   = -2 s0^2 u  P  + w0 Im c
    (checked against autograd in
    tests/test_oracle_golden.py)."""
    c = np.conj(out) * g_out
    P = c.real
    if np.iscomplexobj(lin):
        return (-2.0 * scale0 * scale0) * P * lin - 1j * omega0 * c
    return (-2.0 * scale0 * scale0) * lin * P + omega0 * c.imag


def wire_forward(params: Params, coords: np.ndarray, hidden_layers: int,
                 first_omega0, hidden_omega0, scale0,
                 keep: bool = False):
    """INR.forward of modules/wire.py:161-167 over coords [..., D].

    net[0]   real Linear D->K + Gabor           (modules/wire.py:127-134)
    net[1:L+1] complex Linear K->K + Gabor      (:136-141)
    net[L+1] complex Linear K->O, output .real  (:156-157, :164-165)
    """
    x = coords
    cache = {"x": x, "lin": [], "out": []}
    W = params["net.0.linear.weight"]
    b = params["net.0.linear.bias"]
    lin = x @ W.T + b
    out = gabor_act(lin, first_omega0, scale0)
    cache["lin"].append(lin)
    cache["out"].append(out)
    for l in range(1, hidden_layers + 1):
        W = params[f"net.{l}.linear.weight"]
        b = params[f"net.{l}.linear.bias"]
        lin = out @ W.T + b          # no conjugation (F.linear)
        out = gabor_act(lin, hidden_omega0, scale0)
        cache["lin"].append(lin)
        cache["out"].append(out)
    Wf = params[f"net.{hidden_layers + 1}.weight"]
    bf = params[f"net.{hidden_layers + 1}.bias"]
    y = (out @ Wf.T + bf).real
    if keep:
        return y, cache
    return y


def wire_backward(params: Params, cache, g_y: np.ndarray, hidden_layers: int,
                  first_omega0, hidden_omega0, scale0) -> Params:
    """Closed-form backward of wire_forward (what autograd derives from
    modules/wire.py:89-93).  ``g_y`` is dL/dy, real [..., O].  Returns grads
    keyed like the params.  Leading dims are flattened into N."""
    L = hidden_layers
    K = params["net.0.linear.weight"].shape[0]
    grads: Params = {}
    gy = g_y.reshape(-1, g_y.shape[-1])
    z = cache["out"][L].reshape(-1, K)
    Wf = params[f"net.{L + 1}.weight"]
    # final complex Linear + .real  
    grads[f"net.{L + 1}.weight"] = gy.T.astype(Wf.dtype) @ np.conj(z)
    grads[f"net.{L + 1}.bias"] = gy.sum(0).astype(Wf.dtype)  # imag part = 0
    g_out = gy.astype(Wf.dtype) @ np.conj(Wf)
    for l in range(L, 0, -1):
        lin = cache["lin"][l].reshape(-1, K)
        out = cache["out"][l].reshape(-1, K)
        zin = cache["out"][l - 1].reshape(-1, K)
        W = params[f"net.{l}.linear.weight"]
        g_lin = gabor_act_grad(g_out, lin, out, hidden_omega0, scale0)
        grads[f"net.{l}.linear.weight"] = g_lin.T @ np.conj(zin)
        grads[f"net.{l}.linear.bias"] = g_lin.sum(0)
        g_out = g_lin @ np.conj(W)
    u = cache["lin"][0].reshape(-1, K)
    out0 = cache["out"][0].reshape(-1, K)
    x = cache["x"].reshape(-1, cache["x"].shape[-1])
    g_u = gabor_act_grad(g_out, u, out0, first_omega0, scale0)
    grads["net.0.linear.weight"] = g_u.T @ x
    grads["net.0.linear.bias"] = g_u.sum(0)
    return grads


# --------------------------------------------------------------------------
# WIRE-2D  -- modules/wire2d.py:56-67
# --------------------------------------------------------------------------
def gabor2d_act(lin, sy, omega0, scale0):
    """freq_term * gauss_term of modules/wire2d.py:62-67."""
    freq = np.exp(1j * omega0 * lin)
    arg = np.square(np.abs(lin)) + np.square(np.abs(sy))
    return freq * np.exp(-scale0 * scale0 * arg)


def wire2d_forward(params: Params, coords, hidden_layers, first_omega0,
                   hidden_omega0, scale0, keep=False):
    x = coords
    cache = {"x": x, "lin": [], "sy": [], "out": []}
    out = x
    for l in range(0, hidden_layers + 1):
        W = params[f"net.{l}.linear.weight"]
        b = params[f"net.{l}.linear.bias"]
        V = params[f"net.{l}.scale_orth.weight"]
        c = params[f"net.{l}.scale_orth.bias"]
        lin = out @ W.T + b
        sy = out @ V.T + c
        out = gabor2d_act(lin, sy, first_omega0 if l == 0 else hidden_omega0,
                          scale0)
        cache["lin"].append(lin)
        cache["sy"].append(sy)
        cache["out"].append(out)
    Wf = params[f"net.{hidden_layers + 1}.weight"]
    bf = params[f"net.{hidden_layers + 1}.bias"]
    y = (out @ Wf.T + bf).real
    return (y, cache) if keep else y


def wire2d_backward(params: Params, cache, g_y, hidden_layers, first_omega0,
                    hidden_omega0, scale0) -> Params:
    """g_lin as in 1-D Gabor (with the 2-D out),
    g_sy = -2 s0^2 P sy, g_z = g_lin conj(W) + g_sy conj(V)."""
    L = hidden_layers
    K = params["net.0.linear.weight"].shape[0]
    grads: Params = {}
    gy = g_y.reshape(-1, g_y.shape[-1])
    z = cache["out"][L].reshape(-1, K)
    Wf = params[f"net.{L + 1}.weight"]
    grads[f"net.{L + 1}.weight"] = gy.T.astype(Wf.dtype) @ np.conj(z)
    grads[f"net.{L + 1}.bias"] = gy.sum(0).astype(Wf.dtype)
    g_out = gy.astype(Wf.dtype) @ np.conj(Wf)
    s2 = scale0 * scale0
    for l in range(L, -1, -1):
        lin = cache["lin"][l].reshape(-1, K)
        sy = cache["sy"][l].reshape(-1, K)
        out = cache["out"][l].reshape(-1, K)
        om = first_omega0 if l == 0 else hidden_omega0
        c = np.conj(out) * g_out
        P = c.real
        if l == 0:
            zin = cache["x"].reshape(-1, cache["x"].shape[-1])
            g_lin = -2.0 * s2 * lin * P + om * c.imag
            g_sy = -2.0 * s2 * sy * P
            grads["net.0.linear.weight"] = g_lin.T @ zin
            grads["net.0.linear.bias"] = g_lin.sum(0)
            grads["net.0.scale_orth.weight"] = g_sy.T @ zin
            grads["net.0.scale_orth.bias"] = g_sy.sum(0)
        else:
            zin = cache["out"][l - 1].reshape(-1, K)
            W = params[f"net.{l}.linear.weight"]
            V = params[f"net.{l}.scale_orth.weight"]
            g_lin = -2.0 * s2 * P * lin - 1j * om * c
            g_sy = -2.0 * s2 * P * sy
            grads[f"net.{l}.linear.weight"] = g_lin.T @ np.conj(zin)
            grads[f"net.{l}.linear.bias"] = g_lin.sum(0)
            grads[f"net.{l}.scale_orth.weight"] = g_sy.T @ np.conj(zin)
            grads[f"net.{l}.scale_orth.bias"] = g_sy.sum(0)
            g_out = g_lin @ np.conj(W) + g_sy @ np.conj(V)
    return grads


# --------------------------------------------------------------------------
# real-valued sweep nets (config 5): siren / gauss / relu(+posenc)
# --------------------------------------------------------------------------
def posenc(coords: np.ndarray, num_frequencies: int) -> np.ndarray:
    """PosEncoding.forward, modules/relu.py:62-75: raw coords, then for each
    frequency i, for each dim j: sin(2^i pi c_j), cos(2^i pi c_j)."""
    cols = [coords]
    D = coords.shape[-1]
    for i in range(num_frequencies):
        for j in range(D):
            c = coords[..., j]
            arg = (2 ** i) * np.pi * c
            # the reference multiplies a python float by an fp32 tensor, so
            # the product is rounded to the tensor dtype before sin/cos
            arg = arg.astype(coords.dtype)
            cols.append(np.sin(arg)[..., None])
            cols.append(np.cos(arg)[..., None])
    return np.concatenate(cols, axis=-1)


def posenc_num_frequencies(in_features: int, sidelength, use_nyquist=True) -> int:
    """modules/relu.py:38-60."""
    if in_features == 3:
        return 10
    if in_features == 2:
        if isinstance(sidelength, int):
            sidelength = (sidelength, sidelength)
        nf = 4
        if use_nyquist:
            samples = min(sidelength[0], sidelength[1])
            nyq = 1 / (2 * (2 * 1 / samples))
            nf = int(math.floor(math.log(nyq, 2)))
        return nf
    if in_features == 1:
        nf = 4
        if use_nyquist:
            nyq = 1 / (2 * (2 * 1 / sidelength))
            nf = int(math.floor(math.log(nyq, 2)))
        return nf
    return 4


def real_act(kind: str, lin, omega0, scale0):
    if kind == "siren":      # modules/siren.py:48-49
        return np.sin(omega0 * lin)
    if kind == "gauss":      # modules/gauss.py:27-28
        return np.exp(-np.square(scale0 * lin))
    if kind == "relu":       # modules/relu.py:28-29
        return np.maximum(lin, 0)
    raise ValueError(kind)


def real_act_grad(kind: str, g_out, lin, out, omega0, scale0):
    if kind == "siren":
        return g_out * omega0 * np.cos(omega0 * lin)
    if kind == "gauss":
        return g_out * out * (-2.0 * scale0 * scale0) * lin
    if kind == "relu":
        return g_out * (lin > 0)
    raise ValueError(kind)


def realnet_forward(kind: str, params: Params, coords, hidden_layers,
                    first_omega0, hidden_omega0, scale0,
                    num_frequencies: Optional[int] = None, keep=False):
    """siren/gauss/relu INR.forward (modules/siren.py:90-96,
    modules/gauss.py:71-74, modules/relu.py:124-130), outermost_linear=True."""
    x = coords if num_frequencies is None else posenc(coords, num_frequencies)
    cache = {"x": x, "lin": [], "out": []}
    out = x
    for l in range(hidden_layers + 1):
        W = params[f"net.{l}.linear.weight"]
        b = params[f"net.{l}.linear.bias"]
        lin = out @ W.T + b
        out = real_act(kind, lin, first_omega0 if l == 0 else hidden_omega0,
                       scale0)
        cache["lin"].append(lin)
        cache["out"].append(out)
    Wf = params[f"net.{hidden_layers + 1}.weight"]
    bf = params[f"net.{hidden_layers + 1}.bias"]
    y = out @ Wf.T + bf
    return (y, cache) if keep else y


def realnet_backward(kind: str, params: Params, cache, g_y, hidden_layers,
                     first_omega0, hidden_omega0, scale0, relu_masks=None) -> Params:
    """``relu_masks`` (relu only; one boolean array per layer 0..L, or None): use these ``lin > 0`` decisions instead of
    the cache's own.  The gradient of relu is discontinuous at lin = 0, so two correct implementations disagree on
    elements whose lin is round-off; a comparison of their gradients is meaningful only on identical decisions."""
    L = hidden_layers
    grads: Params = {}
    gy = g_y.reshape(-1, g_y.shape[-1])
    Kf = cache["out"][L].shape[-1]
    z = cache["out"][L].reshape(-1, Kf)
    Wf = params[f"net.{L + 1}.weight"]
    grads[f"net.{L + 1}.weight"] = gy.T @ z
    grads[f"net.{L + 1}.bias"] = gy.sum(0)
    g_out = gy @ Wf
    for l in range(L, -1, -1):
        lin = cache["lin"][l].reshape(-1, cache["lin"][l].shape[-1])
        out = cache["out"][l].reshape(lin.shape)
        zin = cache["x"] if l == 0 else cache["out"][l - 1]
        zin = zin.reshape(-1, zin.shape[-1])
        W = params[f"net.{l}.linear.weight"]
        if relu_masks is not None and kind == "relu":
            g_lin = g_out * relu_masks[l].reshape(lin.shape)
        else:
            g_lin = real_act_grad(kind, g_out, lin, out,
                                  first_omega0 if l == 0 else hidden_omega0, scale0)
        grads[f"net.{l}.linear.weight"] = g_lin.T @ zin
        grads[f"net.{l}.linear.bias"] = g_lin.sum(0)
        if l > 0:
            g_out = g_lin @ W
    return grads


# --------------------------------------------------------------------------
# training-step glue
# --------------------------------------------------------------------------
def mse_loss_and_grad(y: np.ndarray, target: np.ndarray):
    """loss = ((y - t)**2).mean() (wire_image_denoise.py:153;
    torch.nn.MSELoss at wire_occupancy.py:123,150) and dL/dy."""
    diff = y - target
    loss = np.mean(np.square(diff))
    return loss, (2.0 / diff.size) * diff


def avgpool_mse_loss_and_grad(y: np.ndarray, H: int, W: int, scale: int, gt_lr: np.ndarray):
    """Super-resolution loss of wire_SISR.py:151-161: ``rec = torch.nn.AvgPool2d(scale)(rec_hr)`` with
    rec_hr = y reshaped [H, W, O] (row n = i*W + j), ``loss = ((gt_lr - rec)**2).mean()`` over
    [H//scale, W//scale, O] (ceil_mode=False: ragged borders are dropped), and dL/dy [H*W, O].
    Returns (loss, g_y, rec) with rec [H2*W2, O]."""
    O = y.shape[-1]
    H2, W2 = H // scale, W // scale
    img = y.reshape(H, W, O)[:H2 * scale, :W2 * scale]
    rec = img.reshape(H2, scale, W2, scale, O).mean(axis=(1, 3))
    diff = rec - gt_lr.reshape(H2, W2, O)
    loss = np.mean(np.square(diff))
    g_rec = (2.0 / diff.size) * diff
    g = np.zeros((H, W, O), dtype=y.dtype)
    g[:H2 * scale, :W2 * scale] = np.repeat(np.repeat(g_rec, scale, axis=0), scale, axis=1) / (scale * scale)
    return loss, g.reshape(H * W, O), rec.reshape(H2 * W2, O)


def as_real_pairs(a: np.ndarray) -> np.ndarray:
    """torch.view_as_real equivalent (copy)."""
    if np.iscomplexobj(a):
        return np.stack([a.real, a.imag], axis=-1)
    return a


def adam_step(param: np.ndarray, grad: np.ndarray, m: np.ndarray, v: np.ndarray,
              step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.Adam update (defaults: wire_image_denoise.py:123-125).
    Operates on real arrays; complex parameters are passed as real pairs
    (torch handles them through view_as_real).  ``step`` is 1-based.
    Returns (param, m, v)."""
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    param = param - (lr / bc1) * (m / denom)
    return param, m, v


def lambda_lr(base_lr: float, epoch: int, niters: int, gamma: float = 0.1):
    """LambdaLR(optim, lambda x: gamma**min(x/niters, 1)):
    wire_image_denoise.py:128 (gamma 0.1), wire_occupancy.py:122 (0.2)."""
    return base_lr * gamma ** min(epoch / niters, 1)


# --------------------------------------------------------------------------
# coordinate grids and metrics
# --------------------------------------------------------------------------
def linspace_f32(n: int) -> np.ndarray:
    """torch.linspace(-1, 1, n) in fp32 (wire_image_denoise.py:63-64).  ATen's
    kernel (start + i*step below the midpoint, end - (n-1-i)*step above, fused
    multiply-add) is not reproduced op-for-op in numpy: the oracle calls torch
    itself, which is the library the reference calls."""
    import torch
    return torch.linspace(-1, 1, n).numpy().copy()


def image_coords(H: int, W: int) -> np.ndarray:
    """coords of wire_image_denoise.py:63-66: meshgrid(x, y, indexing='xy'),
    row n = i*W + j -> (x_j, y_i).  Shape [H*W, 2] fp32."""
    x = linspace_f32(W)
    y = linspace_f32(H)
    X, Y = np.meshgrid(x, y, indexing="xy")
    return np.hstack((X.reshape(-1, 1), Y.reshape(-1, 1)))


def volume_coords(H: int, W: int, T: int) -> np.ndarray:
    """utils.get_coords(H, W, T) (modules/utils.py:163-176): np.linspace in
    fp64, meshgrid 'xy', row n = (i*W + j)*T + k -> (x_j, y_i, z_k), cast fp32."""
    X, Y, Z = np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H),
                          np.linspace(-1, 1, T))
    c = np.hstack((X.reshape(-1, 1), Y.reshape(-1, 1), Z.reshape(-1, 1)))
    return c.astype(np.float32)


def psnr(x: np.ndarray, xhat: np.ndarray) -> float:
    """modules/utils.py:67-82 -- note max(x), not max(x)**2."""
    err = x - xhat
    denom = np.mean(np.power(err, 2))
    return float(10 * np.log10(np.max(x) / denom))


def iou(preds: np.ndarray, gt: np.ndarray, thres: Optional[float] = None) -> float:
    """volutils.get_IoU (modules/volutils.py:74-91).  The reference binarises
    ``preds`` in place; this restatement works on a copy."""
    p = np.array(preds, copy=True)
    if thres is not None:
        lo = p < thres
        p[lo] = 0.0
        p[~lo] = 1.0
    inter = np.logical_and(p, gt).sum()
    union = np.logical_or(p, gt).sum()
    return float(inter) / float(union)


# --------------------------------------------------------------------------
# parameter construction mirroring the reference's init RNG distribution
# (NOT its RNG stream: bit-identical init comes from the fixtures / torch).
# --------------------------------------------------------------------------
def wire_param_shapes(D: int, K: int, L: int, O: int) -> List[Tuple[str, Tuple[int, ...], bool]]:
    """(name, shape, is_complex) in state_dict order, trainable tensors only."""
    out = [("net.0.linear.weight", (K, D), False), ("net.0.linear.bias", (K,), False)]
    for l in range(1, L + 1):
        out.append((f"net.{l}.linear.weight", (K, K), True))
        out.append((f"net.{l}.linear.bias", (K,), True))
    out.append((f"net.{L + 1}.weight", (O, K), True))
    out.append((f"net.{L + 1}.bias", (O,), True))
    return out


def random_wire_params(rng: np.random.Generator, D: int, K: int, L: int, O: int) -> Params:
    """nn.Linear default init: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias,
    re and im independently for complex dtype."""
    p: Params = {}
    for name, shape, cplx in wire_param_shapes(D, K, L, O):
        fan_in = D if name.startswith("net.0.") else K
        bound = 1.0 / math.sqrt(fan_in)
        if cplx:
            a = rng.uniform(-bound, bound, shape) + 1j * rng.uniform(-bound, bound, shape)
            p[name] = a.astype(np.complex64)
        else:
            p[name] = rng.uniform(-bound, bound, shape).astype(np.float32)
    return p


def wire_flops_per_sample(K: int, L: int, D: int, O: int, backward: bool = True) -> int:
    """fwd+bwd F = 24 K^2 L + 4 D K + 12 K O; fwd 8 K^2 L + 2DK + 4KO."""
    if backward:
        return 24 * K * K * L + 4 * D * K + 12 * K * O
    return 8 * K * K * L + 2 * D * K + 4 * K * O


# --------------------------------------------------------------------------
# position-keyed shuffle (wire_amd: wire_perm_indices; replaces torch.randperm of
# wire_image_denoise.py:142 / wire_occupancy.py:137 in the sharded loops)
# --------------------------------------------------------------------------
def _splitmix64(z: int) -> int:
    M = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def hash_perm(n_total: int, seed: int, first: int = 0, count: Optional[int] = None) -> np.ndarray:
    """pi_seed(first .. first+count) of the keyed bijection of [0, n_total) that libwire_hip's
    perm_indices_kernel evaluates (wire_point.hip): four rounds of x = (x*M_r + K_r) mod 2^b, x ^= x >> s on
    the smallest power-of-two domain, cycle-walking back into [0, n_total).  Test infrastructure."""
    count = n_total - first if count is None else count
    b = 0
    while b < 63 and (1 << b) < n_total:
        b += 1
    b = max(b, 1)
    sh = max(b // 2, 1)
    mask = np.uint64((1 << b) - 1)
    m = [np.uint64(_splitmix64((seed * 8 + r) & ((1 << 64) - 1)) | 1) for r in range(4)]
    k = [np.uint64(_splitmix64((seed * 8 + 4 + r) & ((1 << 64) - 1))) for r in range(4)]
    x = np.arange(first, first + count, dtype=np.uint64)
    todo = np.ones(count, dtype=bool)
    with np.errstate(over="ignore"):
        while todo.any():
            v = x[todo]
            for r in range(4):
                v = (v * m[r] + k[r]) & mask
                v ^= v >> np.uint64(sh)
            x[todo] = v
            todo[todo] = v >= np.uint64(n_total)
    return x.astype(np.int64)
