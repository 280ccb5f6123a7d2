"""SIREN on the shared fp32-MFMA skeleton -- drop-in for the reference's modules/siren.py.

  SineLayer(in_features, out_features, bias, is_first, omega_0, scale,
            init_weights)                          modules/siren.py:26-49
      sin(omega_0 * linear(x)); SIREN init         :39-46
  INR(in_features, hidden_features, hidden_layers, out_features, ...)  :51-96
      final nn.Linear init U(+-sqrt(6/h)/omega)                        :78-80
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ._base import ActivationLayer, FinalLinear, HipINR


class SineLayer(ActivationLayer):
    kind = "siren"

    def __init__(self, in_features, out_features, bias=True, is_first=False, omega_0=30,
                 scale=10.0, init_weights=True):
        super().__init__()
        self.omega_0 = omega_0
        self.is_first = is_first
        self.in_features = in_features
        self.linear = self._build_linear(in_features, out_features, bias, complex_dtype=False)
        if init_weights:
            self.init_weights()

    def init_weights(self):
        bound = (1 / self.in_features) if self.is_first else \
            (np.sqrt(6 / self.in_features) / self.omega_0)
        with torch.no_grad():
            self.linear.weight.uniform_(-bound, bound)

    def abi_tensors(self):
        return [self.linear.weight, self._bias_or_zeros(self.linear)]

    def forward(self, input):
        from .. import functional as Fh
        return Fh.real_layer(self.kind, input, self.linear.weight, self._bias_or_zeros(self.linear),
                             float(self.omega_0), 0.0)


class INR(HipINR):
    kind = "siren"

    def __init__(self, in_features, hidden_features, hidden_layers, out_features,
                 outermost_linear=True, first_omega_0=30, hidden_omega_0=30., scale=10.0,
                 pos_encode=False, sidelength=512, fn_samples=None, use_nyquist=True):
        super().__init__()
        if pos_encode:
            raise NotImplementedError("the reference's siren.INR has no positional_encoding "
                                      "attribute either (modules/siren.py:91-92 would raise)")
        self.pos_encode = pos_encode
        self.nonlin = SineLayer
        layers = [SineLayer(in_features, hidden_features, is_first=True, omega_0=first_omega_0,
                            scale=scale)]
        layers += [SineLayer(hidden_features, hidden_features, is_first=False,
                             omega_0=hidden_omega_0, scale=scale) for _ in range(hidden_layers)]
        if outermost_linear:
            final = FinalLinear(hidden_features, out_features, dtype=torch.float)
            with torch.no_grad():
                const = np.sqrt(6 / hidden_features) / max(hidden_omega_0, 1e-12)
                final.weight.uniform_(-const, const)
            layers.append(final)
        else:                                   # modules/siren.py:81-84
            layers.append(SineLayer(hidden_features, out_features, is_first=False, omega_0=hidden_omega_0,
                                    scale=scale))
        self._finish(layers, in_features, hidden_features, hidden_layers, out_features,
                     first_omega_0, hidden_omega_0, scale, outermost_linear=outermost_linear)
