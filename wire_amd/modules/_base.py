"""Shared machinery of the wire_amd model modules.

The reference builds every INR the same way (e.g. modules/wire.py:127-159):
``net = Sequential(first layer, L hidden layers, final nn.Linear)``.  Here one
base class owns that structure for all nonlinearities and routes
``forward(coords)`` to the fused HIP path (``wire_mlp_fwd`` / ``wire_mlp_bwd``);
the per-file subclasses only pin the reference's constructor signatures,
attribute names and init order (same ``torch.manual_seed`` -> same
``state_dict``, bit for bit).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from .. import _lib, functional as Fh


class ActivationLayer(nn.Module):
    """Linear + nonlinearity with the attribute surface the reference's layer
    classes expose (``linear``, ``omega_0``, ``is_first``, ``in_features``)."""

    kind = "wire"

    def _build_linear(self, in_features: int, out_features: int, bias: bool, complex_dtype: bool):
        dtype = torch.cfloat if complex_dtype else torch.float
        return nn.Linear(in_features, out_features, bias=bias, dtype=dtype)

    def _bias_or_zeros(self, lin: nn.Linear) -> torch.Tensor:
        if lin.bias is not None:
            return lin.bias
        return torch.zeros(lin.out_features, dtype=lin.weight.dtype, device=lin.weight.device)


class FinalLinear(nn.Linear):
    """The outermost nn.Linear (modules/wire.py:156-157).  Inside ``INR.forward``
    it is executed by the fused HIP path; it is kept as an ``nn.Linear`` subclass
    so ``state_dict`` keys (``net.{L+1}.weight/bias``), dtype and default init
    match the reference."""


class HipINR(nn.Module):
    """Base of every ``INR`` class in this package."""

    kind = "wire"

    def _finish(self, layers: List[nn.Module], in_features: int, width: int, hidden_layers: int,
                out_features: int, first_omega_0: float, hidden_omega_0: float, scale: float,
                posenc_freqs: int = 0, outermost_linear: bool = True) -> None:
        self.net = nn.Sequential(*layers)
        self._arch = dict(in_features=int(in_features), width=int(width),
                          hidden_layers=int(hidden_layers), out_features=int(out_features),
                          first_omega0=float(first_omega_0), hidden_omega0=float(hidden_omega_0),
                          scale0=float(scale), posenc_freqs=int(posenc_freqs))
        # outermost_linear=False (modules/siren.py:81-84, gauss.py:63-66, relu.py:116-119): the last module is
        # one more activation layer (K -> O).  The fused whole-net path ends in a linear layer, so such a net
        # runs layer by layer through the per-layer HIP entry points (same kernels, one autograd node each).
        self._layerwise = not outermost_linear
        if any(isinstance(getattr(m, "omega_0", None), nn.Parameter) for m in layers):
            # omega_0 / scale_0 live in the state_dict (modules/wire.py:80-81, wire2d.py:44-45): a checkpoint
            # or a manual edit must reach the fused path's descriptor
            self.register_load_state_dict_post_hook(lambda m, _keys: m.refresh_hparams())

    def refresh_hparams(self) -> None:
        """Re-read omega_0 / scale_0 from the layers' Parameters into the fused path's descriptor (after
        ``load_state_dict`` -- done automatically -- or a manual edit such as ``net[0].omega_0.fill_()``).
        One host sync; not on the hot path.  The fused kernels take ONE first-layer omega, one hidden omega and
        one scale: per-layer values that differ raise."""
        acts = [m for m in self.net if isinstance(getattr(m, "omega_0", None), nn.Parameter)]
        if not acts:
            return
        for m in acts:
            m.refresh_hparams()
        self._arch["first_omega0"] = acts[0]._w
        self._arch["scale0"] = acts[0]._s
        if len(acts) > 1:
            self._arch["hidden_omega0"] = acts[1]._w
            if len({m._s for m in acts}) > 1 or len({m._w for m in acts[1:]}) > 1:
                raise NotImplementedError("per-layer omega_0 / scale_0 values differ; the fused path supports one "
                                          "first-layer omega, one hidden omega and one scale")

    # -- descriptor ---------------------------------------------------------
    def _layer_hparams(self):
        """omega / scale of the fused path's descriptor: the constructor's values, replaced by the layers'
        ``omega_0`` / ``scale_0`` Parameters whenever ``refresh_hparams`` runs (load_state_dict does)."""
        return self._arch["first_omega0"], self._arch["hidden_omega0"], self._arch["scale0"]

    def net_desc(self) -> _lib.NetDesc:
        w1, w, s = self._layer_hparams()
        a = self._arch
        return _lib.make_desc(self.kind, a["in_features"], a["width"], a["hidden_layers"],
                              a["out_features"], w1, w, s, a["posenc_freqs"])

    def param_tensors(self) -> List[torch.Tensor]:
        """Trainable tensors in state_dict order (the ABI's params[] order)."""
        out: List[torch.Tensor] = []
        for m in self.net:
            if isinstance(m, FinalLinear):
                out += [m.weight, m.bias]
            else:
                out += m.abi_tensors()
        return out

    def forward(self, coords: torch.Tensor) -> torch.Tensor:
        if getattr(self, "pos_encode", False) and self._layerwise:
            coords = self.positional_encoding(coords)
        if self._layerwise or any(getattr(m, "trainable", False) for m in self.net):
            out = coords
            for m in self.net:
                if isinstance(m, FinalLinear):      # Re(z W_f^T + b_f) (modules/wire.py:156-157,164-165)
                    if not m.weight.is_complex():
                        raise NotImplementedError("layer-by-layer execution ends in an activation layer or in the "
                                                  "complex final linear of wire / wire2d")
                    out = Fh.final_linear_real(out, m.weight, m.bias)
                else:
                    out = m(out)
            return out
        return Fh.inr_forward(coords, self.net_desc(), self.param_tensors())


def _scalar_param(value: float, trainable: bool) -> nn.Parameter:
    return nn.Parameter(float(value) * torch.ones(1), requires_grad=bool(trainable))


def _param_value(p) -> float:
    if isinstance(p, torch.Tensor):
        return float(p.detach().reshape(-1)[0].item())
    return float(p)
