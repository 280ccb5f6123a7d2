"""The whole-net forward as ONE kernel (wire_amd/csrc/wire_fused.hip): forward-only calls -- ``with torch.no_grad():
model(coords)``, ``FusedTrainer.render`` (modules/volutils.py:124-133, wire_multi_sr.py:215-217) -- of the 256-feature
real nets and of `wire` at padded widths 192 / 256 / 384 keep the activations in the wave's registers from the coordinates
to the output.  Replaces ``self.net(coords)`` of modules/siren.py:90-96, gauss.py:71-74, relu.py:124-130,
wire.py:161-165.

Checked here: against the layer-by-layer kernels on the same weights (knob "fused_fwd" = 0; same split arithmetic, another
summation order inside an MFMA and another activation scale: agreement to fp32 round-off) and against the numpy fp64
oracle under the protocol's ``err_build <= 2 err_ref + 1e-6``, at row counts with a ragged last 128-row tile, for every
shape that has a kernel and for nets with 1, 2, 3 and 4 hidden layers (the accumulator sets alternate between layers).
"""
import numpy as np
import pytest
import torch

from _util import params_np, relmax, within_ref
from oracle import wire_oracle as wo

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = {
    # name: (get_INR kwargs, hidden_layers)
    "siren_4x256": (dict(nonlin="siren", hidden_features=256, first_omega_0=30.0, hidden_omega_0=30.0), 4),
    "gauss_4x256": (dict(nonlin="gauss", hidden_features=256, scale=10.0), 4),
    "relu_4x256": (dict(nonlin="relu", hidden_features=256), 4),
    "siren_3x256": (dict(nonlin="siren", hidden_features=256, first_omega_0=30.0, hidden_omega_0=30.0), 3),
    "gauss_1x256": (dict(nonlin="gauss", hidden_features=256, scale=10.0), 1),
    "relu_2x256": (dict(nonlin="relu", hidden_features=256), 2),
    "wire_k181_4x": (dict(nonlin="wire", hidden_features=256, first_omega_0=20.0, hidden_omega_0=20.0, scale=30.0), 4),
    "wire_k181_classdef": (dict(nonlin="wire", hidden_features=256, first_omega_0=30.0, hidden_omega_0=30.0, scale=10.0), 4),
    "wire_k128_3x": (dict(nonlin="wire", hidden_features=182, first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0), 3),
    "wire_k90_2x_cfg1": (dict(nonlin="wire", hidden_features=128, first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0), 2),
    # positional encoding (modules/relu.py:62-75, sidelength 512 -> 7 frequencies, 30 -> 64 padded features): layer 0 is a GEMM
    # layer of the kernel, its operand evaluated in the lanes
    "relu_posenc_4x256": (dict(nonlin="relu", hidden_features=256, pos_encode=True, sidelength=512), 4),
    "relu_posenc_1x256": (dict(nonlin="relu", hidden_features=256, pos_encode=True, sidelength=512), 1),
}


def _oracle(kind, P, coords, L, om1, om, sc, double, nf=None):
    p = wo.cast_params(P, double)
    rdt = np.float64 if double else np.float32
    c = coords.astype(rdt)
    if kind == "wire":
        return wo.wire_forward(p, c, L, rdt(om1), rdt(om), rdt(sc))
    return wo.realnet_forward(kind, p, c, L, rdt(om1), rdt(om), rdt(sc), nf)


@pytest.mark.parametrize("case", list(CASES))
def test_fused_forward_vs_layerwise_and_fp64_oracle(case):
    from wire_amd import _lib
    from wire_amd.modules import models
    kw, Ln = CASES[case]
    kw = dict(kw)
    kind = kw["nonlin"]
    torch.manual_seed(2)
    model = models.get_INR(in_features=2, out_features=3, hidden_layers=Ln, **kw).to(DEV)
    n = 128 * 70 + 37                                    # >= 4096 rows; ragged last workgroup (37 rows: 2 waves + 5 rows)
    coords_np = wo.image_coords(512, 512)[::29][:n]
    assert coords_np.shape[0] == n
    coords = torch.tensor(coords_np, device=DEV)
    L = _lib.lib()
    assert L.wire_tune_get(b"fused_fwd") == 1
    with torch.no_grad():
        y_fused = model(coords[None])[0].cpu().numpy()
        _lib.check(L.wire_tune_set(b"fused_fwd", 0))
        try:
            y_layer = model(coords[None])[0].cpu().numpy()
        finally:
            _lib.check(L.wire_tune_set(b"fused_fwd", 1))
    assert not np.array_equal(y_fused, y_layer), "the knob did not switch kernels"
    P = params_np(model)
    om1, om, sc = kw.get("first_omega_0", 30.0), kw.get("hidden_omega_0", 30.0), kw.get("scale", 10.0)
    nf = wo.posenc_num_frequencies(2, kw["sidelength"]) if kw.get("pos_encode") else None
    y64 = _oracle(kind, P, coords_np, Ln, om1, om, sc, True, nf)
    y32 = _oracle(kind, P, coords_np, Ln, om1, om, sc, False, nf)
    err_ref = relmax(y32, y64)
    e_f, e_l = relmax(y_fused, y64), relmax(y_layer, y64)
    print(f"fused_fwd[{case}]: fused {e_f:.2e}  layer-by-layer {e_l:.2e}  numpy fp32 {err_ref:.2e}  "
          f"fused vs layer-by-layer {relmax(y_fused, y_layer):.2e}")
    within_ref(e_f, err_ref, f"fused_fwd[{case}] y")
    # the two GPU paths do the same arithmetic up to summation order: they agree as well as either agrees with fp64
    assert relmax(y_fused, y_layer) <= 2 * max(e_f, e_l) + 1e-6


def test_fused_render_equals_model_call_and_covers_every_row():
    """FusedTrainer.render (tiles of 2^20 rows through wire_mlp_fwd with save_for_bwd = 0) runs the fused kernel: same
    values as the module call, every row written (no stale tile tails), sigmoid variant included."""
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    torch.manual_seed(4)
    model = models.get_INR(nonlin="siren", in_features=2, out_features=3, hidden_features=256, hidden_layers=4,
                           first_omega_0=30.0, hidden_omega_0=30.0).to(DEV)
    H, W = 131, 97
    tr = FusedTrainer(model, (H, W), torch.zeros(H * W, 3), lr=0.0)
    out = tr.render(tile=5000)                           # 3 tiles, the last one ragged (and < 4096 rows: layer-by-layer)
    coords = torch.tensor(wo.image_coords(H, W), device=DEV)
    with torch.no_grad():
        ref = model(coords[None])[0]
    assert relmax(out.cpu().numpy(), ref.cpu().numpy()) <= 2e-6
    assert torch.equal(out[:5000], ref[:5000])           # a tile of the fused kernel == the same rows of one big launch


TRAIN_CASES = {
    "siren_4x256": (dict(nonlin="siren", hidden_features=256, first_omega_0=30.0, hidden_omega_0=30.0), 4),
    "gauss_3x256": (dict(nonlin="gauss", hidden_features=256, scale=10.0), 3),
    "relu_4x256": (dict(nonlin="relu", hidden_features=256), 4),
    "wire_k128_2x": (dict(nonlin="wire", hidden_features=182, first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0), 2),
    "siren_1x256": (dict(nonlin="siren", hidden_features=256, first_omega_0=30.0, hidden_omega_0=30.0), 1),
    "relu_posenc_3x256": (dict(nonlin="relu", hidden_features=256, pos_encode=True, sidelength=512), 3),
    # 3-D coordinates and one output (the occupancy drivers' shape, wire_occupancy.py:43-44): D = 3 in the chain's first-layer
    # sums, O = 1 in the final stage, two members in the weight-gradient batch
    # widths below 256 (padded to P = 256: the pad features' activations are act(0) -- 1 for the Gaussian -- behind zero weights)
    # and more layers than the bench nets
    "gauss_3x200": (dict(nonlin="gauss", hidden_features=200, scale=10.0), 3),
    "siren_6x250": (dict(nonlin="siren", hidden_features=250, first_omega_0=30.0, hidden_omega_0=30.0), 6),
    "relu_5x193_4out": (dict(nonlin="relu", hidden_features=193, out_features=4), 5),
    "gauss_2x256_3d": (dict(nonlin="gauss", hidden_features=256, scale=10.0, in_features=3, out_features=1), 2),
    "relu_2x256_3d": (dict(nonlin="relu", hidden_features=256, in_features=3, out_features=1), 2),
}


FINAL_CASES = ["siren_4x256", "gauss_3x256", "relu_4x256", "siren_1x256", "relu_posenc_3x256", "gauss_2x256_3d", "gauss_3x200",
               "relu_5x193_4out"]


@pytest.mark.parametrize("case,final", [(c, 0) for c in TRAIN_CASES] + [(c, 1) for c in FINAL_CASES])
def test_fused_training_forward_vs_layerwise_and_fp64_oracle(case, final):
    """The TRAINING forward as one kernel (knob "fused_train"): FusedTrainer.step stores lin_l / out_l from inside
    fused_fwd_kernel -- lin in the reference's units, out_0 fp32 + its maximum, the inner out_l as unscaled fp16 pairs (relu:
    fp32 + maxima) -- and the unchanged final stage, data-gradient and weight-gradient kernels read them.  Loss, output and
    every gradient against the layer-by-layer forward ("fused_train" = 0) and the fp64 oracle; ragged row count (the
    kernel stores whole 128-row tiles: the padding rows of the act buffer take the overhang).

    final = 1: knob "fused_final" -- the real nets' final linear layer, MSE terms, dL/dy, g_lin_L and the final layer's
    gradient sums formed in the tail of the same kernel (fx_tail_loss; not the default: measured no faster)."""
    from _util import oracle_grads_chunked
    from wire_amd import _lib
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    kw, Ln = TRAIN_CASES[case]
    kw = dict(kw)
    kind = kw["nonlin"]
    L = _lib.lib()
    assert L.wire_tune_get(b"fused_train") == 1
    was_final = L.wire_tune_get(b"fused_final")
    _lib.check(L.wire_tune_set(b"fused_final", final))
    Dn, On = kw.pop("in_features", 2), kw.pop("out_features", 3)
    grid = (131, 97) if Dn == 2 else (23, 29, 19)        # 12 707 rows: 99 workgroups and 35 rows; 12 673 = 99 and 1 row
    N = int(np.prod(grid))
    g = torch.Generator().manual_seed(9)
    target = torch.rand(N, On, generator=g)
    perm = torch.randperm(N, generator=g)
    res = {}
    try:
        for knob in (1, 0):
            _lib.check(L.wire_tune_set(b"fused_train", knob))
            torch.manual_seed(6)
            model = models.get_INR(in_features=Dn, out_features=On, hidden_layers=Ln, **kw).to(DEV)
            tr = FusedTrainer(model, grid, target, lr=0.0, keep_rec=True, coords_style="numpy" if Dn == 3 else "torch")
            loss = tr.step(perm.to(DEV))
            torch.cuda.synchronize()
            res[knob] = (float(loss.item()), tr.flat_grad.cpu().numpy().copy(), tr.rec.cpu().numpy()[perm.numpy()].copy())
            offsets = list(tr.offsets)
    finally:
        _lib.check(L.wire_tune_set(b"fused_train", 1))
        _lib.check(L.wire_tune_set(b"fused_final", was_final))
    assert not np.array_equal(res[0][1], res[1][1]), "the knob did not switch kernels"
    P = params_np(model)
    coords = (wo.image_coords(*grid) if Dn == 2 else wo.volume_coords(*grid))[perm.numpy()]
    tgt = target.numpy()[perm.numpy()]
    om1, om, sc = kw.get("first_omega_0", 30.0), kw.get("hidden_omega_0", 30.0), kw.get("scale", 10.0)
    nf = wo.posenc_num_frequencies(2, kw["sidelength"]) if kw.get("pos_encode") else None
    y64, l64, g64 = oracle_grads_chunked(kind, P, coords, tgt, Ln, om1, om, sc, True, nf)
    y32, l32, g32 = oracle_grads_chunked(kind, P, coords, tgt, Ln, om1, om, sc, False, nf)
    err_y = relmax(y32, y64)
    within_ref(relmax(res[1][2], y64), err_y, f"fused_train[{case}] y")
    assert abs(res[1][0] - l64) <= (2 * abs(l32 - l64) / l64 + 1e-5) * l64
    names = [k for k in model.state_dict().keys() if "omega_0" not in k and "scale_0" not in k]
    for name, off in zip(names, offsets):
        ref = wo.as_real_pairs(g64[name]).astype(np.float64).ravel()
        ref32 = wo.as_real_pairs(g32[name]).astype(np.float64).ravel()
        mine, lay = res[1][1][off:off + ref.size], res[0][1][off:off + ref.size]
        if name == f"net.{Ln + 1}.bias":
            # the final bias is the mean of dL/dy, a few nearly cancelling numbers: against the layer-by-layer path only
            assert relmax(mine, lay) <= 2e-5, f"{case} {name}: {relmax(mine, lay):.3e}"
        elif kind == "relu":
            # relu: a gradient sum moves by whole terms where a pre-activation is round-off (the kink decisions of two correct
            # fp32 implementations differ: test_gpu_timed_kernels.py forces them at bench size).  Here the yardstick is the
            # worse of the two OTHER fp32 implementations at hand, numpy and the layer-by-layer kernels
            within_ref(relmax(mine, ref), max(relmax(ref32, ref), relmax(lay, ref)), f"fused_train[{case}] grad {name}")
        else:
            within_ref(relmax(mine, ref), relmax(ref32, ref), f"fused_train[{case}] grad {name}")


@pytest.mark.parametrize("nonlin", ["siren", "relu"])
def test_wgrad_batch_matches_per_layer_launches(nonlin):
    """Behind the data-gradient chain the weight gradients of layers 1 .. L run as ONE launch of gemmx2_tn16_kernel (knob
    "wgrad_batch", default 1: blockIdx.y = layer, operands a fixed step apart, each member accumulating L times the rows
    into a quarter of the slabs).  Same products, another split of the row sum: every gradient agrees with the per-layer
    launches to fp32 round-off of a 65 536-row sum, and the first layer's and the final layer's -- not touched -- bit for bit.
    Autograd of the hidden Linear layers, modules/siren.py:48-49, relu.py:28-29."""
    from wire_amd import _lib
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    L = _lib.lib()
    assert L.wire_tune_get(b"wgrad_batch") == 1
    grid = (256, 256)
    N = grid[0] * grid[1]
    g = torch.Generator().manual_seed(3)
    target = torch.rand(N, 3, generator=g)
    perm = torch.randperm(N, generator=g)
    res = {}
    try:
        for knob in (1, 0):
            _lib.check(L.wire_tune_set(b"wgrad_batch", knob))
            torch.manual_seed(8)
            kw = dict(first_omega_0=30.0, hidden_omega_0=30.0) if nonlin == "siren" else {}
            model = models.get_INR(nonlin=nonlin, in_features=2, out_features=3, hidden_features=256, hidden_layers=4, **kw).to(DEV)
            tr = FusedTrainer(model, grid, target, lr=0.0)
            tr.step(perm.to(DEV))
            torch.cuda.synchronize()
            res[knob] = tr.flat_grad.cpu().numpy().copy()
            offsets, names = list(tr.offsets), [k for k in model.state_dict().keys() if "omega_0" not in k and "scale_0" not in k]
    finally:
        _lib.check(L.wire_tune_set(b"wgrad_batch", 1))
    sizes = np.diff(offsets + [res[1].size])
    switched = False
    for name, off, sz in zip(names, offsets, sizes):
        a, b = res[1][off:off + sz], res[0][off:off + sz]
        layer = int(name.split(".")[1])
        if 1 <= layer <= 4:                              # (layer 1 is a member too: its operand r_0 / relu's out_0 has the others' form)
            switched = switched or not np.array_equal(a, b)
            assert relmax(a, b) <= 1e-5, f"{name}: {relmax(a, b):.3e}"
        else:
            assert np.array_equal(a, b), f"{name} moved although its kernels did not change"
    assert switched, "the knob did not switch kernels"
