"""WIRE with the 2-D Gabor activation -- drop-in for the reference's modules/wire2d.py.

  ComplexGaborLayer2D(in_features, out_features, bias, is_first, omega0, sigma0,
                      trainable)                  modules/wire2d.py:21-54
      out = exp(j w0 lin) * exp(-s0^2 (|lin|^2 + |scale_orth(x)|^2))   :56-67
  INR(in_features, hidden_features, hidden_layers, out_features, ...)  :70-127
      hidden width = int(hidden_features / 2)                          :92
Forward/backward of ``INR.forward`` run in libwire_hip.so: both Linears of a
layer are one fp32-MFMA GEMM with a 128-column wave tile, so lin and scale_orth
land in the same lane for the fused epilogue.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as Fh
from ._base import ActivationLayer, FinalLinear, HipINR, _param_value, _scalar_param


class ComplexGaborLayer2D(ActivationLayer):
    kind = "wire2d"

    def __init__(self, in_features, out_features, bias=True, is_first=False,
                 omega0=10.0, sigma0=10.0, trainable=False):
        super().__init__()
        self.trainable = bool(trainable)
        self.is_first = is_first
        self.in_features = in_features
        self.omega_0 = _scalar_param(omega0, trainable)
        self.scale_0 = _scalar_param(sigma0, trainable)
        self.linear = self._build_linear(in_features, out_features, bias, complex_dtype=not is_first)
        # second Gaussian window (modules/wire2d.py:50-54)
        self.scale_orth = self._build_linear(in_features, out_features, bias, complex_dtype=not is_first)
        self._w = float(omega0)
        self._s = float(sigma0)

    def refresh_hparams(self):
        self._w = _param_value(self.omega_0)
        self._s = _param_value(self.scale_0)

    def abi_tensors(self):
        return [self.linear.weight, self._bias_or_zeros(self.linear),
                self.scale_orth.weight, self._bias_or_zeros(self.scale_orth)]

    def forward(self, input):
        if self.trainable:      # omega_0 / scale_0 receive gradients (modules/wire2d.py:42-43 with trainable=True)
            return Fh.gabor2d_layer_trainable(input, self.linear.weight, self._bias_or_zeros(self.linear),
                                              self.scale_orth.weight, self._bias_or_zeros(self.scale_orth),
                                              self.omega_0, self.scale_0, self.is_first)
        return Fh.gabor2d_layer(input, self.linear.weight, self._bias_or_zeros(self.linear),
                                self.scale_orth.weight, self._bias_or_zeros(self.scale_orth),
                                self._w, self._s, self.is_first)


class INR(HipINR):
    kind = "wire2d"

    def __init__(self, in_features, hidden_features, hidden_layers, out_features,
                 outermost_linear=True, first_omega_0=10, hidden_omega_0=10., scale=10.0,
                 pos_encode=False, sidelength=512, fn_samples=None, use_nyquist=True):
        super().__init__()
        self.nonlin = ComplexGaborLayer2D
        width = int(hidden_features / 2)      # modules/wire2d.py:92
        self.complex = True
        self.wavelet = 'gabor'
        self.pos_encode = False
        layers = [ComplexGaborLayer2D(in_features, width, omega0=first_omega_0, sigma0=scale,
                                      is_first=True, trainable=False)]
        layers += [ComplexGaborLayer2D(width, width, omega0=hidden_omega_0, sigma0=scale)
                   for _ in range(hidden_layers)]
        layers.append(FinalLinear(width, out_features, dtype=torch.cfloat))
        self._finish(layers, in_features, width, hidden_layers, out_features,
                     first_omega_0, hidden_omega_0, scale)
