"""Linear inverse-problem operators on the hot path's contract (reference modules/lin_inverse.py).

Only the CT forward operator the WIRE driver uses (wire_ct.py:128-133): ``radon``.  The reference rotates the
image once per angle with ``kornia.geometry.rotate`` and sums over the rows; here the rotate-and-sum and its
adjoint are one HIP kernel each (wire_radon_fwd / wire_radon_bwd), wrapped in an autograd.Function so that
``loss.backward()`` reaches the model.  Video compressive-sensing masks and the other helpers of that file are out
of scope.
"""
from __future__ import annotations

import torch

from .. import _lib


class _RadonFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, angles):
        L = _lib.lib()
        if not img.is_cuda:
            raise _lib.WireHipError("radon: tensors must be on the MI355X ('cuda'); wire_amd has no CPU path")
        H, W = img.shape[-2], img.shape[-1]
        x = img.detach().to(torch.float32).contiguous().reshape(-1, H, W)
        ang = angles.detach().to(img.device, torch.float32).contiguous()
        A = ang.numel()
        out = torch.empty(x.shape[0], A, W, dtype=torch.float32, device=img.device)
        stream = torch.cuda.current_stream(img.device).cuda_stream
        for i in range(x.shape[0]):
            _lib.check(L.wire_radon_fwd(stream, x[i].data_ptr(), ang.data_ptr(), H, W, A, out[i].data_ptr()),
                       "wire_radon_fwd")
        ctx.save_for_backward(ang)
        ctx.shape = (tuple(img.shape), H, W, A)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        (ang,) = ctx.saved_tensors
        shape, H, W, A = ctx.shape
        gg = g.detach().to(torch.float32).contiguous().reshape(-1, A, W)
        gi = torch.empty(gg.shape[0], H, W, dtype=torch.float32, device=g.device)
        stream = torch.cuda.current_stream(g.device).cuda_stream
        for i in range(gg.shape[0]):
            _lib.check(L.wire_radon_bwd(stream, gg[i].data_ptr(), ang.data_ptr(), H, W, A, gi[i].data_ptr()),
                       "wire_radon_bwd")
        return gi.reshape(shape), None


def radon(imten, angles, is_3d=False):
    """Forward Radon transform (modules/lin_inverse.py:19-40).

    imten: (1, nimg, H, W) image tensor; angles: (nangles) degrees, same device.
    Returns the sinogram: (nangles, W) for one image (the reference's ``.squeeze()``), (nimg, nangles, W) with
    ``is_3d=True``."""
    if imten.dim() != 4 or imten.shape[0] != 1:
        raise ValueError("radon expects imten of shape (1, nimg, H, W)")
    sino = _RadonFunction.apply(imten[0], angles)          # (nimg, nangles, W)
    if is_3d:
        return sino
    return sino.permute(1, 0, 2).squeeze()
