"""CPU-only checks of the boundary: the C-ABI library loads and exports what
include/wire_hip.h declares, the size queries are consistent, get_INR keeps the
reference's call styles, and the product path refuses to run without a GPU /
without the HIP library (no silent fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from _util import ROOT


def test_header_and_library_agree():
    from wire_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "wire_hip.h")).read()
    declared = set(re.findall(r"\b(wire_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.wire_abi_version() == 1
    assert lib.wire_blocked_width(181) == 384 and lib.wire_blocked_width(256) == 512


def test_size_queries_match_reference_shapes():
    from wire_amd import _lib
    from wire_amd.modules import models, utils
    lib = _lib.lib()
    torch.manual_seed(0)
    for nonlin, kw in [("wire", {}), ("wire2d", {}), ("siren", {}), ("gauss", {}), ("relu", {}),
                       ("relu", {"pos_encode": True})]:
        m = models.get_INR(nonlin=nonlin, in_features=2, out_features=3, hidden_features=64,
                           hidden_layers=2, **kw)
        d = m.net_desc()
        tensors = m.param_tensors()
        assert lib.wire_num_param_tensors(C.byref(d)) == len(tensors)
        total = 0
        for i, t in enumerate(tensors):
            fl = lib.wire_param_tensor_floats(C.byref(d), i)
            assert fl == t.numel() * (2 if t.is_complex() else 1), (nonlin, i)
            total += t.numel()
        assert total == utils.count_parameters(m)
        assert lib.wire_packed_floats(C.byref(d)) > 0
        a1 = lib.wire_act_bytes(C.byref(d), 1000, 1)
        a0 = lib.wire_act_bytes(C.byref(d), 1000, 0)
        assert a1 > a0 > 0
        assert lib.wire_bwd_scratch_bytes(C.byref(d), 1000) > 0
    bad = _lib.make_desc("wire", 7, 64, 2, 3, 30, 30, 10)
    assert lib.wire_packed_floats(C.byref(bad)) < 0
    assert b"in_features" in lib.wire_last_error()


def test_get_inr_call_styles():
    from wire_amd.modules import models, utils
    # wire_image_denoise.py:106-118 style (keywords, no scaled_hidden_features)
    m = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=300,
                       hidden_layers=2, first_omega_0=7.0, hidden_omega_0=7.0, scale=8.0,
                       scale_tensor=[1.0], pos_encode=False, sidelength=678)
    assert utils.count_parameters(m) == 91587          # Agg_results.md:3
    assert m.complex and m.wavelet == "gabor" and m.pos_encode is False
    assert m.net[0].linear.weight.shape == (212, 2)
    # bspline_image_denoise.py:95-108 style (scaled_hidden_features given)
    m2 = models.get_INR("wire", 2, 300, 0, 2, 3, scale_tensor=[0.0])
    assert utils.count_parameters(m2) == 91587
    # wire_occupancy.py:107-116 style
    m3 = models.get_INR(nonlin="relu", in_features=3, out_features=1, hidden_features=64,
                        hidden_layers=2, pos_encode=True, sidelength=128)
    assert m3.positional_encoding.out_dim == 63 and m3.net[0].linear.in_features == 63
    m4 = models.get_INR(nonlin="wire2d", in_features=2, out_features=3, hidden_features=256, hidden_layers=4)
    assert utils.count_parameters(m4) == 133251
    with pytest.raises(NotImplementedError):
        models.get_INR(nonlin="mfn", in_features=2, out_features=3, hidden_features=64, hidden_layers=2)
    with pytest.raises(TypeError):
        models.get_INR("wire", 2, 64)
    # standalone layer constructor, modules/wire.py:59-66
    from wire_amd.modules.wire import ComplexGaborLayer
    lay = ComplexGaborLayer(5, 7)
    assert lay.linear.weight.dtype == torch.complex64 and abs(lay.scale_0.item() - 40.0) < 1e-6
    assert not lay.omega_0.requires_grad


def test_no_cpu_fallback_and_no_oracle_import():
    from wire_amd import _lib
    from wire_amd.modules import models
    m = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=64, hidden_layers=1)
    with pytest.raises(_lib.WireHipError):
        m(torch.zeros(1, 8, 2))
    # a missing library must raise, not fall back
    saved, cached = _lib.LIB_PATH, _lib._lib
    try:
        _lib.LIB_PATH, _lib._lib = "/nonexistent/libwire_hip.so", None
        with pytest.raises(_lib.WireHipError):
            _lib.lib()
    finally:
        _lib.LIB_PATH, _lib._lib = saved, cached
    # the product package never imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "wire_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dirpath, f)


def test_shard_bounds_cover_batch():
    from wire_amd.parallel import shard_bounds, shard_weight
    for B in (0, 1, 7, 200000, 262144, 134217729):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard_bounds(B, world, r)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == B
            if B:
                assert abs(sum(shard_weight(B, world, r) for r in range(world)) - 1.0) < 1e-12


def test_utils_match_reference_known_answers():
    from _util import load_golden
    from wire_amd.modules import utils
    misc = load_golden("misc")
    assert abs(utils.psnr(misc["psnr_x"], misc["psnr_xhat"]) - float(misc["psnr_val"])) < 1e-4
    np.testing.assert_array_equal(utils.get_coords(6, 5, 4).numpy(), misc["coords3d_6_5_4"])
    np.testing.assert_array_equal(utils.get_coords(7, 9).numpy(), misc["coords2d_7_9"])
    tx, ty, _ = utils.axis_tables(5, 7, style="torch")
    np.testing.assert_array_equal(tx.numpy(), misc["linspace_7"])
    np.testing.assert_array_equal(ty.numpy(), misc["linspace_5"])
    # (PosEncoding.forward runs on the device -- wire_posenc_fwd; its known answer is checked in tests/test_gpu_parity.py)
    from wire_amd.modules.relu import PosEncoding
    from wire_amd._lib import WireHipError
    import pytest
    pe = PosEncoding(2, sidelength=512)
    assert pe.num_frequencies == 7 and pe.out_dim == 30
    with pytest.raises(WireHipError):
        pe(torch.tensor(misc["posenc2_in"]))


def test_fused_trainer_rejects_layerwise_and_trainable_nets():
    """FusedTrainer runs the descriptor's net (linear output, fixed omega_0 / scale_0).  A net that
    HipINR.forward runs layer by layer -- outermost_linear=False (modules/siren.py:81-84, gauss.py:63-66,
    relu.py:116-119) or trainable=True layers (modules/wire.py:80-81) -- is a different function there and must be
    refused, not silently trained as something else."""
    import pytest
    import torch
    from wire_amd.modules import models
    from wire_amd.modules.wire import ComplexGaborLayer
    from wire_amd.trainer import FusedTrainer
    tgt = torch.zeros(16 * 16, 3)
    m = models.get_INR(nonlin="siren", in_features=2, out_features=3, hidden_features=32, hidden_layers=1,
                       outermost_linear=False)
    with pytest.raises(NotImplementedError, match="outermost_linear"):
        FusedTrainer(m, (16, 16), tgt)
    m = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=32, hidden_layers=1)
    m.net[1] = ComplexGaborLayer(m.net[1].linear.in_features, m.net[1].linear.out_features, omega0=5.0, sigma0=5.0,
                                 trainable=True)
    with pytest.raises(NotImplementedError, match="trainable"):
        FusedTrainer(m, (16, 16), tgt)
