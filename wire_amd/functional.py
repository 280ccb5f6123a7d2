"""torch.autograd.Functions over the libwire_hip C ABI.

PyTorch here is plumbing only: it owns device buffers (so the caching allocator
and stream semantics apply), records the autograd edge, and hands raw pointers
plus the current HIP stream to the library.  All arithmetic of the MLP stack --
forward and backward -- happens in the HIP kernels.

Backward runs on autograd's own thread; nothing here depends on thread-local
state except the library's error string.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _lib


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.WireHipError(
            f"{what} is on {t.device}; wire_amd runs on an MI355X only (no CPU fallback). "
            "Move the model and its inputs to 'cuda'.")


def _native(t: torch.Tensor) -> torch.Tensor:
    """Contiguous tensor whose storage is the native layout the ABI expects
    (complex64 = interleaved floats)."""
    return t.detach().contiguous()


class _INRFunction(torch.autograd.Function):
    """Whole-network forward/backward: wire_mlp_fwd / wire_mlp_bwd."""

    @staticmethod
    def forward(ctx, coords: torch.Tensor, desc: _lib.NetDesc, grad_mode: bool, *params: torch.Tensor):
        L = _lib.lib()
        _require_cuda(coords, "coords")
        dev = coords.device
        D, O = desc.in_features, desc.out_features
        if coords.shape[-1] != D:
            raise ValueError(f"coords last dim {coords.shape[-1]} != in_features {D}")
        x = coords.detach().to(torch.float32).contiguous()
        n = x.numel() // D
        nat = [_native(p) for p in params]
        for p in nat:
            _require_cuda(p, "a parameter")
        stream = _stream_ptr(dev)
        packed = torch.empty(L.wire_packed_floats(C.byref(desc)), dtype=torch.float32, device=dev)
        _lib.check(L.wire_pack_params(stream, C.byref(desc), _lib.ptr_array([p.data_ptr() for p in nat]),
                                      packed.data_ptr()), "wire_pack_params")
        # (inside torch.no_grad() needs_input_grad still reports the parameters' requires_grad; no graph is recorded
        #  there, so nothing is saved and the forward-only kernels run.  grad_mode = torch.is_grad_enabled() at the call:
        #  inside forward() it is always off)
        need_bwd = grad_mode and any(ctx.needs_input_grad[3:])
        act_bytes = _lib.check(L.wire_act_bytes(C.byref(desc), n, int(need_bwd)), "wire_act_bytes")
        act = torch.empty(act_bytes, dtype=torch.uint8, device=dev)
        y = torch.empty(tuple(coords.shape[:-1]) + (O,), dtype=torch.float32, device=dev)
        _lib.check(L.wire_mlp_fwd(stream, C.byref(desc), packed.data_ptr(), x.data_ptr(), n,
                                  y.data_ptr(), act.data_ptr(), act_bytes, int(need_bwd)),
                   "wire_mlp_fwd")
        if need_bwd:
            ctx.desc, ctx.n = desc, n
            ctx.packed, ctx.act, ctx.x = packed, act, x
            ctx.meta = [(p.shape, p.dtype) for p in params]
        return y

    @staticmethod
    def backward(ctx, g_y: torch.Tensor):
        L = _lib.lib()
        desc, n = ctx.desc, ctx.n
        dev = g_y.device
        gy = g_y.detach().to(torch.float32).contiguous()
        grads = [torch.empty(shape, dtype=dtype, device=dev) for shape, dtype in ctx.meta]
        stream = _stream_ptr(dev)
        sbytes = _lib.check(L.wire_bwd_scratch_bytes(C.byref(desc), n), "wire_bwd_scratch_bytes")
        scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.wire_mlp_bwd(stream, C.byref(desc), ctx.packed.data_ptr(), ctx.x.data_ptr(), n,
                                  gy.data_ptr(), ctx.act.data_ptr(), ctx.act.numel(),
                                  scratch.data_ptr(), sbytes,
                                  _lib.ptr_array([g.data_ptr() for g in grads])), "wire_mlp_bwd")
        # the saved buffers stay with the graph node (freed with it), so backward(retain_graph=True) can run again.
        # No gradient is returned for `coords` (the reference's autograd would produce one; no caller in the
        # reference asks for it -- INTEGRATION.md)
        return (None, None, None, *grads)


def inr_forward(coords: torch.Tensor, desc: _lib.NetDesc, params: Sequence[torch.Tensor]) -> torch.Tensor:
    return _INRFunction.apply(coords, desc, torch.is_grad_enabled(), *params)


class _GaborLayerFunction(torch.autograd.Function):
    """One ComplexGaborLayer on native tensors: wire_gabor_fwd / wire_gabor_bwd."""

    @staticmethod
    def forward(ctx, x, W, b, omega0: float, scale0: float, is_first: bool):
        L = _lib.lib()
        _require_cuda(x, "layer input")
        _require_cuda(W, "layer weight")
        dev = x.device
        out_f, in_f = W.shape
        xin = x.detach().to(torch.float32 if is_first else torch.complex64).contiguous()
        n = xin.numel() // in_f
        Wn, bn = _native(W), _native(b)
        ws_bytes = _lib.check(L.wire_layer_ws_bytes(n, in_f, out_f), "wire_layer_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        act = torch.empty(tuple(x.shape[:-1]) + (out_f,), dtype=torch.complex64, device=dev)
        _lib.check(L.wire_gabor_fwd(_stream_ptr(dev), xin.data_ptr(), Wn.data_ptr(), bn.data_ptr(),
                                    omega0, scale0, n, in_f, out_f, int(is_first), None,
                                    act.data_ptr(), ws.data_ptr(), ws_bytes), "wire_gabor_fwd")
        ctx.save_for_backward(xin, Wn, bn)
        ctx.cfg = (omega0, scale0, is_first, n, in_f, out_f, tuple(x.shape))
        return act

    @staticmethod
    def backward(ctx, g_act):
        L = _lib.lib()
        xin, Wn, bn = ctx.saved_tensors
        omega0, scale0, is_first, n, in_f, out_f, xshape = ctx.cfg
        dev = g_act.device
        g = g_act.detach().to(torch.complex64).contiguous()
        ws_bytes = _lib.check(L.wire_layer_ws_bytes(n, in_f, out_f), "wire_layer_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        gW = torch.empty_like(Wn)
        gb = torch.empty_like(bn)
        gx = None if is_first else torch.empty(xshape, dtype=torch.complex64, device=dev)
        _lib.check(L.wire_gabor_bwd(_stream_ptr(dev), g.data_ptr(), xin.data_ptr(), Wn.data_ptr(),
                                    bn.data_ptr(), omega0, scale0, n, in_f, out_f, int(is_first),
                                    None if gx is None else gx.data_ptr(), gW.data_ptr(),
                                    gb.data_ptr(), ws.data_ptr(), ws_bytes), "wire_gabor_bwd")
        return gx, gW, gb, None, None, None


def gabor_layer(x, W, b, omega0: float, scale0: float, is_first: bool):
    return _GaborLayerFunction.apply(x, W, b, float(omega0), float(scale0), bool(is_first))


class _GaborLayerTrainableFunction(torch.autograd.Function):
    """ComplexGaborLayer with trainable omega_0 / scale_0 (modules/wire.py:80-81, trainable=True):
    wire_gabor_fwd / wire_gabor_bwd + wire_gabor_hparam_grad."""

    @staticmethod
    def forward(ctx, x, W, b, omega, scale, is_first: bool):
        omega0, scale0 = float(omega.detach().reshape(-1)[0]), float(scale.detach().reshape(-1)[0])
        with torch.no_grad():
            act = _GaborLayerFunction.apply(x.detach(), W.detach(), b.detach(), omega0, scale0, is_first)
        in_f = W.shape[1]
        xin = x.detach().to(torch.float32 if is_first else torch.complex64).contiguous()
        ctx.save_for_backward(xin, _native(W), _native(b))
        ctx.cfg = (omega0, scale0, is_first, xin.numel() // in_f, in_f, W.shape[0], tuple(x.shape), omega.shape,
                   scale.shape)
        return act

    @staticmethod
    def backward(ctx, g_act):
        L = _lib.lib()
        xin, Wn, bn = ctx.saved_tensors
        omega0, scale0, is_first, n, in_f, out_f, xshape, oshape, sshape = ctx.cfg
        dev = g_act.device
        g = g_act.detach().to(torch.complex64).contiguous()
        ws_bytes = _lib.check(L.wire_layer_ws_bytes(n, in_f, out_f), "wire_layer_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        gW, gb = torch.empty_like(Wn), torch.empty_like(bn)
        gx = None if is_first else torch.empty(xshape, dtype=torch.complex64, device=dev)
        stream = _stream_ptr(dev)
        _lib.check(L.wire_gabor_bwd(stream, g.data_ptr(), xin.data_ptr(), Wn.data_ptr(), bn.data_ptr(), omega0,
                                    scale0, n, in_f, out_f, int(is_first), None if gx is None else gx.data_ptr(),
                                    gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws_bytes), "wire_gabor_bwd")
        hp = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.check(L.wire_gabor_hparam_grad(stream, g.data_ptr(), xin.data_ptr(), Wn.data_ptr(), bn.data_ptr(),
                                            omega0, scale0, n, in_f, out_f, int(is_first), hp.data_ptr(),
                                            ws.data_ptr(), ws_bytes), "wire_gabor_hparam_grad")
        return gx, gW, gb, hp[0].reshape(oshape), hp[1].reshape(sshape), None


def gabor_layer_trainable(x, W, b, omega: torch.Tensor, scale: torch.Tensor, is_first: bool):
    return _GaborLayerTrainableFunction.apply(x, W, b, omega, scale, bool(is_first))


class _Gabor2DLayerFunction(torch.autograd.Function):
    """ComplexGaborLayer2D.forward (modules/wire2d.py:56-67): wire_gabor2d_fwd / wire_gabor2d_bwd."""

    @staticmethod
    def forward(ctx, x, W, b, V, c, omega0: float, scale0: float, is_first: bool):
        L = _lib.lib()
        _require_cuda(x, "layer input")
        _require_cuda(W, "layer weight")
        dev = x.device
        out_f, in_f = W.shape
        xin = x.detach().to(torch.float32 if is_first else torch.complex64).contiguous()
        n = xin.numel() // in_f
        Wn, bn, Vn, cn = _native(W), _native(b), _native(V), _native(c)
        ws_bytes = _lib.check(L.wire_layer2d_ws_bytes(n, in_f, out_f), "wire_layer2d_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        act = torch.empty(tuple(x.shape[:-1]) + (out_f,), dtype=torch.complex64, device=dev)
        _lib.check(L.wire_gabor2d_fwd(_stream_ptr(dev), xin.data_ptr(), Wn.data_ptr(), bn.data_ptr(),
                                      Vn.data_ptr(), cn.data_ptr(), omega0, scale0, n, in_f, out_f,
                                      int(is_first), act.data_ptr(), ws.data_ptr(), ws_bytes), "wire_gabor2d_fwd")
        ctx.save_for_backward(xin, Wn, bn, Vn, cn)
        ctx.cfg = (omega0, scale0, is_first, n, in_f, out_f, tuple(x.shape))
        return act

    @staticmethod
    def backward(ctx, g_act):
        L = _lib.lib()
        xin, Wn, bn, Vn, cn = ctx.saved_tensors
        omega0, scale0, is_first, n, in_f, out_f, xshape = ctx.cfg
        dev = g_act.device
        g = g_act.detach().to(torch.complex64).contiguous()
        ws_bytes = _lib.check(L.wire_layer2d_ws_bytes(n, in_f, out_f), "wire_layer2d_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        gW, gb, gV, gc = (torch.empty_like(t) for t in (Wn, bn, Vn, cn))
        gx = None if is_first else torch.empty(xshape, dtype=torch.complex64, device=dev)
        _lib.check(L.wire_gabor2d_bwd(_stream_ptr(dev), g.data_ptr(), xin.data_ptr(), Wn.data_ptr(),
                                      bn.data_ptr(), Vn.data_ptr(), cn.data_ptr(), omega0, scale0, n, in_f,
                                      out_f, int(is_first), None if gx is None else gx.data_ptr(),
                                      gW.data_ptr(), gb.data_ptr(), gV.data_ptr(), gc.data_ptr(),
                                      ws.data_ptr(), ws_bytes), "wire_gabor2d_bwd")
        return gx, gW, gb, gV, gc, None, None, None


def gabor2d_layer(x, W, b, V, c, omega0: float, scale0: float, is_first: bool):
    return _Gabor2DLayerFunction.apply(x, W, b, V, c, float(omega0), float(scale0), bool(is_first))


class _Gabor2DLayerTrainableFunction(torch.autograd.Function):
    """ComplexGaborLayer2D with trainable omega_0 / scale_0 (modules/wire2d.py:42-43, trainable=True):
    wire_gabor2d_fwd / wire_gabor2d_bwd + wire_gabor2d_hparam_grad."""

    @staticmethod
    def forward(ctx, x, W, b, V, c, omega, scale, is_first: bool):
        omega0, scale0 = float(omega.detach().reshape(-1)[0]), float(scale.detach().reshape(-1)[0])
        with torch.no_grad():
            act = _Gabor2DLayerFunction.apply(x.detach(), W.detach(), b.detach(), V.detach(), c.detach(), omega0,
                                              scale0, is_first)
        in_f = W.shape[1]
        xin = x.detach().to(torch.float32 if is_first else torch.complex64).contiguous()
        ctx.save_for_backward(xin, _native(W), _native(b), _native(V), _native(c))
        ctx.cfg = (omega0, scale0, is_first, xin.numel() // in_f, in_f, W.shape[0], tuple(x.shape), omega.shape,
                   scale.shape)
        return act

    @staticmethod
    def backward(ctx, g_act):
        L = _lib.lib()
        xin, Wn, bn, Vn, cn = ctx.saved_tensors
        omega0, scale0, is_first, n, in_f, out_f, xshape, oshape, sshape = ctx.cfg
        dev = g_act.device
        g = g_act.detach().to(torch.complex64).contiguous()
        ws_bytes = _lib.check(L.wire_layer2d_ws_bytes(n, in_f, out_f), "wire_layer2d_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        gW, gb, gV, gc = (torch.empty_like(t) for t in (Wn, bn, Vn, cn))
        gx = None if is_first else torch.empty(xshape, dtype=torch.complex64, device=dev)
        stream = _stream_ptr(dev)
        _lib.check(L.wire_gabor2d_bwd(stream, g.data_ptr(), xin.data_ptr(), Wn.data_ptr(), bn.data_ptr(), Vn.data_ptr(),
                                      cn.data_ptr(), omega0, scale0, n, in_f, out_f, int(is_first),
                                      None if gx is None else gx.data_ptr(), gW.data_ptr(), gb.data_ptr(),
                                      gV.data_ptr(), gc.data_ptr(), ws.data_ptr(), ws_bytes), "wire_gabor2d_bwd")
        hp = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.check(L.wire_gabor2d_hparam_grad(stream, g.data_ptr(), xin.data_ptr(), Wn.data_ptr(), bn.data_ptr(),
                                              Vn.data_ptr(), cn.data_ptr(), omega0, scale0, n, in_f, out_f,
                                              int(is_first), hp.data_ptr(), ws.data_ptr(), ws_bytes),
                   "wire_gabor2d_hparam_grad")
        return gx, gW, gb, gV, gc, hp[0].reshape(oshape), hp[1].reshape(sshape), None


def gabor2d_layer_trainable(x, W, b, V, c, omega: torch.Tensor, scale: torch.Tensor, is_first: bool):
    return _Gabor2DLayerTrainableFunction.apply(x, W, b, V, c, omega, scale, bool(is_first))


class _FinalLinearFunction(torch.autograd.Function):
    """Re(z W_f^T + b_f): wire_final_fwd / wire_final_bwd."""

    @staticmethod
    def forward(ctx, z, Wf, bf):
        L = _lib.lib()
        _require_cuda(z, "final-layer input")
        dev = z.device
        out_f, in_f = Wf.shape
        zin = z.detach().to(torch.complex64).contiguous()
        n = zin.numel() // in_f
        Wn, bn = _native(Wf), _native(bf)
        ws_bytes = _lib.check(L.wire_layer_ws_bytes(n, in_f, out_f), "wire_layer_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        y = torch.empty(tuple(z.shape[:-1]) + (out_f,), dtype=torch.float32, device=dev)
        _lib.check(L.wire_final_fwd(_stream_ptr(dev), zin.data_ptr(), Wn.data_ptr(), bn.data_ptr(), n,
                                    in_f, out_f, y.data_ptr(), ws.data_ptr(), ws_bytes),
                   "wire_final_fwd")
        ctx.save_for_backward(zin, Wn)
        ctx.cfg = (n, in_f, out_f, tuple(z.shape), bn.shape)
        return y

    @staticmethod
    def backward(ctx, g_y):
        L = _lib.lib()
        zin, Wn = ctx.saved_tensors
        n, in_f, out_f, zshape, bshape = ctx.cfg
        dev = g_y.device
        gy = g_y.detach().to(torch.float32).contiguous()
        ws_bytes = _lib.check(L.wire_layer_ws_bytes(n, in_f, out_f), "wire_layer_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        gz = torch.empty(zshape, dtype=torch.complex64, device=dev)
        gW = torch.empty_like(Wn)
        gb = torch.empty(bshape, dtype=torch.complex64, device=dev)
        _lib.check(L.wire_final_bwd(_stream_ptr(dev), gy.data_ptr(), zin.data_ptr(), Wn.data_ptr(), n,
                                    in_f, out_f, gz.data_ptr(), gW.data_ptr(), gb.data_ptr(),
                                    ws.data_ptr(), ws_bytes), "wire_final_bwd")
        return gz, gW, gb


def final_linear_real(z, Wf, bf):
    return _FinalLinearFunction.apply(z, Wf, bf)


class _RealLayerFunction(torch.autograd.Function):
    """SineLayer / GaussLayer / ReLULayer on native tensors: wire_real_layer_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, W, b, kind: str, omega0: float, scale0: float):
        L = _lib.lib()
        _require_cuda(x, "layer input")
        _require_cuda(W, "layer weight")
        dev = x.device
        out_f, in_f = W.shape
        xin = x.detach().to(torch.float32).contiguous()
        n = xin.numel() // in_f
        Wn, bn = _native(W), _native(b)
        ws_bytes = _lib.check(L.wire_layer_ws_bytes(n, in_f, out_f), "wire_layer_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        act = torch.empty(tuple(x.shape[:-1]) + (out_f,), dtype=torch.float32, device=dev)
        _lib.check(L.wire_real_layer_fwd(_stream_ptr(dev), _lib.KIND[kind], xin.data_ptr(), Wn.data_ptr(),
                                         bn.data_ptr(), omega0, scale0, n, in_f, out_f, act.data_ptr(),
                                         ws.data_ptr(), ws_bytes), "wire_real_layer_fwd")
        ctx.save_for_backward(xin, Wn, bn)
        ctx.cfg = (kind, omega0, scale0, n, in_f, out_f, tuple(x.shape), x.requires_grad)
        return act

    @staticmethod
    def backward(ctx, g_act):
        L = _lib.lib()
        xin, Wn, bn = ctx.saved_tensors
        kind, omega0, scale0, n, in_f, out_f, xshape, need_gx = ctx.cfg
        dev = g_act.device
        g = g_act.detach().to(torch.float32).contiguous()
        ws_bytes = _lib.check(L.wire_layer_ws_bytes(n, in_f, out_f), "wire_layer_ws_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        gW = torch.empty_like(Wn)
        gb = torch.empty_like(bn)
        gx = torch.empty(xshape, dtype=torch.float32, device=dev) if need_gx else None
        _lib.check(L.wire_real_layer_bwd(_stream_ptr(dev), _lib.KIND[kind], g.data_ptr(), xin.data_ptr(),
                                         Wn.data_ptr(), bn.data_ptr(), omega0, scale0, n, in_f, out_f,
                                         None if gx is None else gx.data_ptr(), gW.data_ptr(), gb.data_ptr(),
                                         ws.data_ptr(), ws_bytes), "wire_real_layer_bwd")
        return gx, gW, gb, None, None, None


def real_layer(kind: str, x, W, b, omega0: float, scale0: float):
    return _RealLayerFunction.apply(x, W, b, kind, float(omega0), float(scale0))
