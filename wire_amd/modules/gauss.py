"""Gaussian-activation INR -- drop-in for the reference's modules/gauss.py.

  GaussLayer(in_features, out_features, bias, is_first, omega_0, scale)  modules/gauss.py:15-28
      exp(-(scale * linear(x))**2)
  INR(in_features, hidden_features, hidden_layers, out_features, ...)    :31-74
"""
from __future__ import annotations

import torch
from torch import nn

from ._base import ActivationLayer, FinalLinear, HipINR


class GaussLayer(ActivationLayer):
    kind = "gauss"

    def __init__(self, in_features, out_features, bias=True, is_first=False, omega_0=30,
                 scale=10.0):
        super().__init__()
        self.in_features = in_features
        self.omega_0 = omega_0
        self.scale = scale
        self.is_first = is_first
        self.linear = self._build_linear(in_features, out_features, bias, complex_dtype=False)

    def abi_tensors(self):
        return [self.linear.weight, self._bias_or_zeros(self.linear)]

    def forward(self, input):
        from .. import functional as Fh
        return Fh.real_layer(self.kind, input, self.linear.weight, self._bias_or_zeros(self.linear),
                             0.0, float(self.scale))


class INR(HipINR):
    kind = "gauss"

    def __init__(self, in_features, hidden_features, hidden_layers, out_features,
                 outermost_linear=True, first_omega_0=30, hidden_omega_0=30., scale=10.0,
                 pos_encode=False, sidelength=512, fn_samples=None, use_nyquist=True):
        super().__init__()
        self.pos_encode = pos_encode
        self.complex = False
        self.nonlin = GaussLayer
        layers = [GaussLayer(in_features, hidden_features, is_first=True, omega_0=first_omega_0,
                             scale=scale)]
        layers += [GaussLayer(hidden_features, hidden_features, is_first=False,
                              omega_0=hidden_omega_0, scale=scale) for _ in range(hidden_layers)]
        if outermost_linear:
            layers.append(FinalLinear(hidden_features, out_features, dtype=torch.float))
        else:                                   # modules/gauss.py:63-66
            layers.append(GaussLayer(hidden_features, out_features, is_first=False, omega_0=hidden_omega_0,
                                     scale=scale))
        self._finish(layers, in_features, hidden_features, hidden_layers, out_features,
                     first_omega_0, hidden_omega_0, scale, outermost_linear=outermost_linear)
