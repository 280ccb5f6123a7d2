"""ReLU (+ NeRF positional encoding) INR -- drop-in for the reference's modules/relu.py.

  ReLULayer(in_features, out_features, bias, is_first, omega_0, scale)   modules/relu.py:17-29
  PosEncoding(in_features, sidelength, fn_samples, use_nyquist)          :31-75
      out_dim = D + 2*D*F; per frequency i, per dim j: sin(2^i pi c_j), cos(2^i pi c_j)
  INR(in_features, hidden_features, hidden_layers, out_features, ...)    :77-130
Inside ``INR.forward`` the encoding is a HIP prologue kernel feeding the first
layer's GEMM; ``PosEncoding.forward`` as a stand-alone module is index/trig
plumbing on the caller's device.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from ._base import ActivationLayer, FinalLinear, HipINR


class ReLULayer(ActivationLayer):
    kind = "relu"

    def __init__(self, in_features, out_features, bias=True, is_first=False, omega_0=30,
                 scale=10.0):
        super().__init__()
        self.in_features = in_features
        self.omega_0 = omega_0
        self.is_first = is_first
        self.linear = self._build_linear(in_features, out_features, bias, complex_dtype=False)

    def abi_tensors(self):
        return [self.linear.weight, self._bias_or_zeros(self.linear)]

    def forward(self, input):
        from .. import functional as Fh
        return Fh.real_layer(self.kind, input, self.linear.weight, self._bias_or_zeros(self.linear),
                             0.0, 0.0)


class PosEncoding(nn.Module):
    """Frequency count rules of modules/relu.py:38-60."""

    def __init__(self, in_features, sidelength=None, fn_samples=None, use_nyquist=True):
        super().__init__()
        self.in_features = in_features
        nf = 4
        if in_features == 3:
            nf = 10
        elif in_features == 2:
            assert sidelength is not None
            if isinstance(sidelength, int):
                sidelength = (sidelength, sidelength)
            if use_nyquist:
                nf = self.get_num_frequencies_nyquist(min(sidelength[0], sidelength[1]))
        elif in_features == 1:
            fn_samples = sidelength
            if use_nyquist:
                nf = self.get_num_frequencies_nyquist(fn_samples)
        self.num_frequencies = nf
        self.out_dim = in_features + 2 * in_features * nf

    def get_num_frequencies_nyquist(self, samples):
        nyquist_rate = 1 / (2 * (2 * 1 / samples))
        return int(math.floor(math.log(nyquist_rate, 2)))

    def forward(self, coords):
        """modules/relu.py:62-75 on the device: ``wire_posenc_fwd`` (the kernel the fused path runs as its first-layer
        prologue).  No gradient flows to the coordinates (no caller in the reference asks for one)."""
        import ctypes  # noqa: F401
        from .. import _lib
        if not coords.is_cuda:
            raise _lib.WireHipError(f"PosEncoding input is on {coords.device}; wire_amd runs on an MI355X only")
        x = coords.detach().to(torch.float32).reshape(coords.shape[0], -1, self.in_features).contiguous()
        B, n = x.shape[0], x.shape[1]
        out = torch.empty(B, n, self.out_dim, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().wire_posenc_fwd(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), B * n,
                                              self.in_features, self.num_frequencies, out.data_ptr()), "wire_posenc_fwd")
        return out


class INR(HipINR):
    kind = "relu"

    def __init__(self, in_features, hidden_features, hidden_layers, out_features,
                 outermost_linear=True, first_omega_0=30, hidden_omega_0=30., scale=10.0,
                 pos_encode=False, sidelength=512, fn_samples=None, use_nyquist=True):
        super().__init__()
        self.pos_encode = pos_encode
        self.complex = False
        self.nonlin = ReLULayer
        first_in = in_features
        freqs = 0
        if pos_encode:
            self.positional_encoding = PosEncoding(in_features=in_features, sidelength=sidelength,
                                                   fn_samples=fn_samples, use_nyquist=use_nyquist)
            first_in = self.positional_encoding.out_dim
            freqs = self.positional_encoding.num_frequencies
        layers = [ReLULayer(first_in, hidden_features, is_first=True, omega_0=first_omega_0,
                            scale=scale)]
        layers += [ReLULayer(hidden_features, hidden_features, is_first=False,
                             omega_0=hidden_omega_0, scale=scale) for _ in range(hidden_layers)]
        if outermost_linear:
            layers.append(FinalLinear(hidden_features, out_features, dtype=torch.float))
        else:                                   # modules/relu.py:116-119
            layers.append(ReLULayer(hidden_features, out_features, is_first=False, omega_0=hidden_omega_0,
                                    scale=scale))
        self._finish(layers, in_features, hidden_features, hidden_layers, out_features,
                     first_omega_0, hidden_omega_0, scale, posenc_freqs=freqs, outermost_linear=outermost_linear)
