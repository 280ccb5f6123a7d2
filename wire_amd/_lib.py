"""ctypes binding of libwire_hip.so (include/wire_hip.h).

The product path has no CPU or PyTorch-op fallback: if the HIP library is not
built, ``lib()`` raises.  Build it with ``python -c "import __graft_entry__ as
g; g.build()"`` or ``make -C wire_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch must load ITS HIP runtime (torch/lib/libamdhip64.so) before this library
# pulls one in by SONAME: two runtimes in one process see no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwire_hip.so")

KIND = {"wire": 0, "wire2d": 1, "siren": 2, "gauss": 3, "relu": 4}
ABI_VERSION = 1

# every symbol include/wire_hip.h declares (tests check the .so exports them)
SYMBOLS = [
    "wire_abi_version", "wire_last_error", "wire_num_param_tensors",
    "wire_param_tensor_floats", "wire_packed_floats", "wire_act_bytes",
    "wire_bwd_scratch_bytes", "wire_pack_params", "wire_mlp_fwd", "wire_mlp_bwd",
    "wire_layer_ws_bytes", "wire_gabor_fwd", "wire_gabor_bwd", "wire_final_fwd",
    "wire_final_bwd", "wire_coords_from_index", "wire_mse_grad",
    "wire_adam_step_flat", "wire_blocked_width", "wire_c64_to_blocked",
    "wire_blocked_to_c64", "wire_prof_enable", "wire_prof_read", "wire_tune_set", "wire_tune_get", "wire_avgpool_mse_grad", "wire_layer2d_ws_bytes", "wire_gabor2d_fwd", "wire_gabor2d_bwd", "wire_eval_metric", "wire_real_layer_fwd", "wire_real_layer_bwd", "wire_train_fwd_bwd", "wire_perm_indices", "wire_gabor_hparam_grad", "wire_track_best", "wire_sigmoid_inplace", "wire_radon_fwd", "wire_radon_bwd", "wire_gabor2d_hparam_grad", "wire_posenc_fwd", "wire_act_out_offset", "wire_train_fwd_bwd_hooked",
]


class NetDesc(C.Structure):
    """struct wire_net_desc"""
    _fields_ = [("kind", C.c_int32), ("in_features", C.c_int32), ("width", C.c_int32),
                ("hidden_layers", C.c_int32), ("out_features", C.c_int32),
                ("posenc_freqs", C.c_int32), ("first_omega0", C.c_float),
                ("hidden_omega0", C.c_float), ("scale0", C.c_float)]


# wire_grad_ready_fn (include/wire_hip.h): void (*)(void* user, int first_tensor, int n_tensors)
GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)


class WireHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def _declare(l: C.CDLL) -> None:
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    dp = C.POINTER(NetDesc)
    l.wire_abi_version.restype = i32
    l.wire_last_error.restype = C.c_char_p
    l.wire_num_param_tensors.argtypes = [dp]
    l.wire_param_tensor_floats.argtypes = [dp, i32]
    l.wire_param_tensor_floats.restype = i64
    l.wire_packed_floats.argtypes = [dp]
    l.wire_packed_floats.restype = i64
    l.wire_act_bytes.argtypes = [dp, i64, i32]
    l.wire_act_bytes.restype = i64
    l.wire_bwd_scratch_bytes.argtypes = [dp, i64]
    l.wire_bwd_scratch_bytes.restype = i64
    l.wire_pack_params.argtypes = [vp, dp, C.POINTER(vp), vp]
    l.wire_mlp_fwd.argtypes = [vp, dp, vp, vp, i64, vp, vp, i64, i32]
    l.wire_mlp_bwd.argtypes = [vp, dp, vp, vp, i64, vp, vp, i64, vp, i64, C.POINTER(vp)]
    l.wire_layer_ws_bytes.argtypes = [i64, i32, i32]
    l.wire_layer_ws_bytes.restype = i64
    l.wire_gabor_fwd.argtypes = [vp, vp, vp, vp, f32, f32, i64, i32, i32, i32, vp, vp, vp, i64]
    l.wire_gabor_bwd.argtypes = [vp, vp, vp, vp, vp, f32, f32, i64, i32, i32, i32, vp, vp, vp, vp, i64]
    l.wire_gabor_hparam_grad.argtypes = [vp, vp, vp, vp, vp, f32, f32, i64, i32, i32, i32, vp, vp, i64]
    l.wire_track_best.argtypes = [vp, vp, vp, i32, vp, vp, i64, vp]
    l.wire_sigmoid_inplace.argtypes = [vp, vp, i64]
    l.wire_posenc_fwd.argtypes = [vp, vp, i64, i32, i32, vp]
    l.wire_act_out_offset.argtypes = [dp, i64, i32]
    l.wire_act_out_offset.restype = i64
    l.wire_radon_fwd.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    l.wire_radon_bwd.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    l.wire_final_fwd.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp, i64]
    l.wire_final_bwd.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, i64]
    l.wire_coords_from_index.argtypes = [vp, vp, i64, i64, vp, i32, vp, i32, vp, i32, vp]
    l.wire_perm_indices.argtypes = [vp, C.c_uint64, i64, i64, i64, vp]
    l.wire_mse_grad.argtypes = [vp, vp, vp, vp, i64, i64, i32, f32, vp, vp, vp, vp]
    l.wire_adam_step_flat.argtypes = [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, i64]
    l.wire_blocked_width.argtypes = [i32]
    l.wire_c64_to_blocked.argtypes = [vp, vp, i64, i32, vp]
    l.wire_blocked_to_c64.argtypes = [vp, vp, i64, i32, vp]
    l.wire_tune_set.argtypes = [C.c_char_p, i32]
    l.wire_tune_get.argtypes = [C.c_char_p]
    l.wire_layer2d_ws_bytes.argtypes = [i64, i32, i32]
    l.wire_layer2d_ws_bytes.restype = i64
    l.wire_gabor2d_fwd.argtypes = [vp, vp, vp, vp, vp, vp, f32, f32, i64, i32, i32, i32, vp, vp, i64]
    l.wire_gabor2d_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, f32, f32, i64, i32, i32, i32, vp, vp, vp, vp, vp,
                                   vp, i64]
    l.wire_gabor2d_hparam_grad.argtypes = [vp, vp, vp, vp, vp, vp, vp, f32, f32, i64, i32, i32, i32, vp, vp, i64]
    l.wire_avgpool_mse_grad.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    l.wire_train_fwd_bwd.argtypes = [vp, dp, vp, vp, i64, vp, vp, i64, f32, vp, vp, vp, vp, vp, vp, i64, vp, i64,
                                     C.POINTER(vp)]
    l.wire_train_fwd_bwd_hooked.argtypes = [vp, dp, vp, vp, i64, vp, vp, i64, f32, vp, vp, vp, vp, vp, vp, i64, vp, i64,
                                            C.POINTER(vp), GRAD_READY_FN, vp]
    l.wire_real_layer_fwd.argtypes = [vp, i32, vp, vp, vp, f32, f32, i64, i32, i32, vp, vp, i64]
    l.wire_real_layer_bwd.argtypes = [vp, i32, vp, vp, vp, vp, f32, f32, i64, i32, i32, vp, vp, vp, vp, i64]
    l.wire_eval_metric.argtypes = [vp, i32, vp, vp, i64, f32, vp, vp]
    l.wire_prof_enable.argtypes = [i32]
    l.wire_prof_read.argtypes = [C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]


def lib() -> C.CDLL:
    """Load libwire_hip.so once; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WireHipError(
                f"{LIB_PATH} is not built: run `make -C wire_amd/csrc` (needs hipcc, "
                "--offload-arch=gfx950).  wire_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        _declare(l)
        v = l.wire_abi_version()
        if v != ABI_VERSION:
            raise WireHipError(f"libwire_hip.so ABI {v} != expected {ABI_VERSION}")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = lib().wire_last_error().decode(errors="replace")
        raise WireHipError(f"{what or 'libwire_hip'} failed ({rc}): {msg}")
    return rc


def make_desc(kind: str, in_features: int, width: int, hidden_layers: int, out_features: int,
              first_omega0: float, hidden_omega0: float, scale0: float,
              posenc_freqs: int = 0) -> NetDesc:
    return NetDesc(KIND[kind], int(in_features), int(width), int(hidden_layers),
                   int(out_features), int(posenc_freqs), float(first_omega0),
                   float(hidden_omega0), float(scale0))


def ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
