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


# ---------------------------------------------------------------------------
# parity bound of SURVEY.md section 7 (protocol step ii) and a log of what was measured
# ---------------------------------------------------------------------------
RATIO_LOG = []


def within_ref(err_build, err_ref, label, factor=2.0, floor=1e-6):
    """err_build <= factor * err_ref + floor, where err_ref is the reference arithmetic's own fp32-vs-fp64
    error on the same inputs (``err_build <= 2 err_ref + 1e-6``).  Every comparison is
    logged; conftest.py writes the log to gpurun_out/parity_ratios.txt at the end of a GPU session."""
    RATIO_LOG.append((label, float(err_build), float(err_ref)))
    assert err_build <= factor * err_ref + floor, \
        f"{label}: err_build {err_build:.3e} > {factor} x err_ref {err_ref:.3e} + {floor:g}"


def final_bias_within_ref(g, g64, err_y_ref, ymax, O, label, factor=2.0, floor=1e-6, resid_max=0.0):
    """The gradient of the final bias is the MEAN of dL/dy: g_bf[o] = (2 / (n O)) sum_n (y - t)[n, o]
    (modules/wire.py:156-157 + the MSE of wire_image_denoise.py:153).  It is O <= 3 numbers -- for O = 1 a single
    one -- that can cancel to nearly zero, so max|error| / max|value| over the tensor itself is one noisy draw,
    not a statistic.  The protocol's bound on the forward output propagates exactly instead:
        |dg_bf[o]| <= (2 / (n O)) sum_n |dy[n, o]| <= (2 / O) max|dy| <= (2 / O) (2 err_ref_y + 1e-6) max|y64|.
    The fp32 sum of the residuals y - t itself rounds relative to THEIR size, so the scale of the bound is
    max(max|y64|, max|y64 - t|) when the caller passes the residuals' maximum (a net whose outputs are still
    small next to the target: relu at init)."""
    g, g64 = np.asarray(g, np.float64).ravel(), np.asarray(g64, np.float64).ravel()
    err = np.abs(g - g64).max() / ((2.0 / O) * max(float(ymax), float(resid_max)))
    RATIO_LOG.append((label + " [forward-propagated bound]", float(err), float(err_y_ref)))
    assert err <= factor * err_y_ref + floor, \
        f"{label}: |dg| / ((2/O) max|y|) = {err:.3e} > {factor} x err_ref_y {err_y_ref:.3e} + {floor:g}"


def family_ctx(fam):
    """Context manager selecting a GEMM family of libwire_hip -- 'x2': 2 x fp16 split on the f16 MFMA (the default at
    >= 4096 rows and the one bench.py times), 'x3': 3 x bf16 split on the bf16 MFMA (round 2's default; what smaller
    batches run), '3m' / '4m': fp32 MFMA -- and restoring the default afterwards."""
    import contextlib
    from wire_amd import _lib

    @contextlib.contextmanager
    def ctx():
        L = _lib.lib()
        sb, c3, f16 = {"x2": (1, 1, 1), "x3": (1, 1, 0), "3m": (0, 1, 1), "4m": (0, 0, 1)}[fam]
        _lib.check(L.wire_tune_set(b"split_bf16", sb))
        _lib.check(L.wire_tune_set(b"complex_3m", c3))
        _lib.check(L.wire_tune_set(b"split_f16", f16))
        try:
            yield L
        finally:
            _lib.check(L.wire_tune_set(b"split_bf16", 1))
            _lib.check(L.wire_tune_set(b"complex_3m", 1))
            _lib.check(L.wire_tune_set(b"split_f16", 1))
    return ctx()


def _host_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def oracle_grads_chunked(kind, P, coords, target, L, om1, om, sc, double, nf=None, chunk=16384, relu_masks=None):
    """Loss and every parameter gradient of the MSE over ALL rows (mean over n x O elements,
    wire_image_denoise.py:153), evaluated by the numpy oracle in row chunks so that BASELINE.json's full batches
    (262 144 x 256 complex128 activations = 0.5 GB per layer; 1 048 576 rows for wire2d) stay small; the chunks run on
    a thread pool (numpy releases the GIL; one BLAS thread per worker) and are summed in chunk order, in the chunk's
    own precision.  kind: 'wire' (modules/wire.py:161-167), 'wire2d' (modules/wire2d.py:56-67,124-130), 'siren' /
    'gauss' / 'relu' (modules/siren.py:90-96, gauss.py:71-74, relu.py:124-130; ``nf`` = number of positional-encoding
    frequencies, relu.py:62-75, or None).  ``relu_masks``: per-layer boolean [n, K] arrays forcing the relu decisions
    (oracle.realnet_backward); the returned dict then also holds "flips" = the number of forced decisions that differ from
    the oracle's own and "flip_lin_max" = the largest |lin| among them.  Returns (y, loss, grads)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import wire_oracle as wo
    rdt = np.float64 if double else np.float32
    p = wo.cast_params(P, double)
    n, O = target.shape[0], target.shape[1]

    def one(s):
        c = coords[s:s + chunk].astype(rdt)
        t = target[s:s + chunk].astype(rdt)
        a = (rdt(om1), rdt(om), rdt(sc))
        if kind == "wire":
            y, cache = wo.wire_forward(p, c, L, *a, keep=True)
        elif kind == "wire2d":
            y, cache = wo.wire2d_forward(p, c, L, *a, keep=True)
        else:
            y, cache = wo.realnet_forward(kind, p, c, L, *a, nf, keep=True)
        diff = y - t
        sq = float(np.square(diff.astype(np.float64)).sum())
        gy = (rdt(2.0) / rdt(n * O)) * diff
        if kind == "wire":
            g = wo.wire_backward(p, cache, gy, L, *a)
        elif kind == "wire2d":
            g = wo.wire2d_backward(p, cache, gy, L, *a)
        else:
            mk = None if relu_masks is None else [m[s:s + chunk] for m in relu_masks]
            g = wo.realnet_backward(kind, p, cache, gy, L, *a, relu_masks=mk)
            if mk is not None:
                fl, fm = 0, 0.0
                for l, m in enumerate(mk):
                    d = m != (cache["lin"][l] > 0)
                    fl += int(d.sum())
                    if d.any():
                        fm = max(fm, float(np.abs(cache["lin"][l][d]).max()))
                g = dict(g)
                g["__flips"] = np.array([fl, fm], np.float64)
        return y, sq, g

    starts = list(range(0, n, chunk))
    workers = min(_host_threads(), len(starts))
    if workers > 1:
        try:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=1)
        except Exception:
            limit = None
        try:
            with ThreadPoolExecutor(max_workers=workers) as ex:
                parts = list(ex.map(one, starts))
        finally:
            if limit is not None:
                limit.restore_original_limits()
    else:
        parts = [one(s) for s in starts]
    grads, sq, flips, flipmax = None, 0.0, 0, 0.0
    for _, q, g in parts:
        sq += q
        if "__flips" in g:
            g = dict(g)
            f = g.pop("__flips")
            flips += int(f[0]); flipmax = max(flipmax, float(f[1]))
        grads = g if grads is None else {k: grads[k] + g[k] for k in g}
    if relu_masks is not None:
        grads["flips"], grads["flip_lin_max"] = flips, flipmax
    return np.concatenate([y for y, _, _ in parts], 0), sq / (n * O), grads


def wire_oracle_grads_chunked(P, coords, target, L, om1, om, sc, double, chunk=16384):
    """oracle_grads_chunked for the 1-D Gabor net (modules/wire.py:161-167)."""
    return oracle_grads_chunked("wire", P, coords, target, L, om1, om, sc, double, None, chunk)
