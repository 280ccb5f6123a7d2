"""WIRE (complex Gabor) INR on MI355X -- drop-in for the reference's modules/wire.py.

API parity (reference file:line):
  ComplexGaborLayer(in_features, out_features, bias, is_first, omega0, sigma0,
                    trainable)                    modules/wire.py:59-86
      .forward(x) -> complex64 activations        modules/wire.py:88-93
  INR(in_features, hidden_features, scaled_hidden_features, hidden_layers,
      out_features, ...)                          modules/wire.py:96-159
      .forward(coords) -> real [..., out]         modules/wire.py:161-167
state_dict keys, dtypes (complex64 parameters) and default-init RNG order are
the reference's, so checkpoints move both ways.  The arithmetic is not PyTorch:
forward and backward run in libwire_hip.so -- by default the hidden-layer GEMMs as
a 2 x fp16 split on the f16 matrix cores with fp32 accumulation and fused Gabor
epilogues (wire_gemmx2h.hip; forward-only calls of the narrower nets as ONE kernel,
wire_fused.hip), selectable: 3 x bf16 split, exact-fp32 MFMA; there
is no CPU path.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .. import functional as Fh
from ._base import ActivationLayer, FinalLinear, HipINR, _param_value, _scalar_param


class ComplexGaborLayer(ActivationLayer):
    """exp(j*omega0*lin - |sigma0*lin|^2) after a (complex) Linear."""

    kind = "wire"

    def __init__(self, in_features, out_features, bias=True, is_first=False,
                 omega0=10.0, sigma0=40.0, trainable=False):
        super().__init__()
        self.trainable = bool(trainable)
        self.is_first = is_first
        self.in_features = in_features
        # same creation order as the reference so the RNG stream lines up
        self.omega_0 = _scalar_param(omega0, trainable)
        self.scale_0 = _scalar_param(sigma0, trainable)
        self.linear = self._build_linear(in_features, out_features, bias, complex_dtype=not is_first)
        self._w = float(omega0)
        self._s = float(sigma0)

    def refresh_hparams(self):
        self._w = _param_value(self.omega_0)
        self._s = _param_value(self.scale_0)

    def abi_tensors(self):
        return [self.linear.weight, self._bias_or_zeros(self.linear)]

    def forward(self, input):
        if self.trainable:
            # omega_0 / scale_0 receive gradients (modules/wire.py:80-81 with trainable=True): their current
            # values are read per call (one host sync) and two column sums are added to the backward
            return Fh.gabor_layer_trainable(input, self.linear.weight, self._bias_or_zeros(self.linear),
                                            self.omega_0, self.scale_0, self.is_first)
        return Fh.gabor_layer(input, self.linear.weight, self._bias_or_zeros(self.linear),
                              self._w, self._s, self.is_first)


class INR(HipINR):
    kind = "wire"

    def __init__(self, in_features, hidden_features, scaled_hidden_features, hidden_layers,
                 out_features, outermost_linear=True, first_omega_0=30, hidden_omega_0=30.,
                 scale=10.0, scale_tensor=[], pos_encode=False, multi_scale=False,
                 sidelength=512, fn_samples=None, use_nyquist=True):
        super().__init__()
        self.nonlin = ComplexGaborLayer
        # complex features carry two reals each: the reference narrows the net
        # by sqrt(2) (modules/wire.py:119)
        width = int(hidden_features / np.sqrt(2))
        self.complex = True
        self.wavelet = 'gabor'
        self.pos_encode = False      # legacy flag, always False (modules/wire.py:125)

        layers = [ComplexGaborLayer(in_features, width, omega0=first_omega_0, sigma0=scale,
                                    is_first=True, trainable=False)]
        layers += [ComplexGaborLayer(width, width, omega0=hidden_omega_0, sigma0=scale)
                   for _ in range(hidden_layers)]
        layers.append(FinalLinear(width, out_features, dtype=torch.cfloat))
        self._finish(layers, in_features, width, hidden_layers, out_features,
                     first_omega_0, hidden_omega_0, scale)
