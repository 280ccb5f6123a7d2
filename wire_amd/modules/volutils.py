"""Volume helpers on the hot path's contract (reference modules/volutils.py).

Only what the occupancy training / evaluation loop touches (wire_occupancy.py:160-172): the IoU metric.  Mesh
extraction (marching cubes, open3d), noise models and SSIM helpers are out of scope.
The counts are reduced on the device (wire_eval_metric mode 1): no host pass over the 134 M voxels of a 512^3
volume.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


def get_I_and_U(preds, gt, thres=None):
    """Intersection and union counts (modules/volutils.py:78-91).  As in the reference, ``preds`` is binarised
    IN PLACE when ``thres`` is given -- the caller's tensor afterwards holds 0 / 1 (wire_occupancy.py keeps that
    binarised volume as ``best_img``).  CUDA float32 tensors take the device reduction; numpy arrays the
    reference's own numpy expression."""
    if isinstance(preds, np.ndarray):
        if thres is not None:
            preds[preds < thres] = 0.0
            preds[preds >= thres] = 1.0
        return np.logical_and(preds, gt).sum(), np.logical_or(preds, gt).sum()
    if not preds.is_cuda:
        raise _lib.WireHipError("get_I_and_U: tensors must be on the MI355X ('cuda'); wire_amd has no CPU path")
    L = _lib.lib()
    p = preds.detach()
    g = gt.detach().to(p.device, torch.float32).contiguous()
    if thres is not None:
        p.masked_fill_(p < thres, 0.0)
        p.masked_fill_(p >= thres, 1.0)
    # without a threshold the reference's logical_and / logical_or test for non-zero (either sign)
    flat = p.to(torch.float32).contiguous() if thres is not None else (p != 0).to(torch.float32).contiguous()
    # exact integer counts, as the reference's logical_and(...).sum(): the device reduction accumulates in fp32, which
    # holds every integer up to 2^24, so the volume goes in slabs of at most 2^24 voxels (a 512^3 volume has 2^27) whose
    # counts are converted to int64 before they are added up -- all on the device, no host sync
    out = torch.empty(2, dtype=torch.float32, device=p.device)
    partial = torch.empty(4096, dtype=torch.float32, device=p.device)
    flat, g = flat.reshape(-1), g.reshape(-1)
    inter = torch.zeros((), dtype=torch.int64, device=p.device)
    union = torch.zeros((), dtype=torch.int64, device=p.device)
    stream = torch.cuda.current_stream(p.device).cuda_stream
    for b in range(0, flat.numel(), 1 << 24):
        n = min(1 << 24, flat.numel() - b)
        # the volume now holds 0 / 1: counting "pred >= 0.5" counts its ones
        _lib.check(L.wire_eval_metric(stream, 1, flat.data_ptr() + 4 * b, g.data_ptr() + 4 * b, n, 0.5, out.data_ptr(),
                                      partial.data_ptr()), "wire_eval_metric")
        cnt = out.to(torch.int64)
        inter = inter + cnt[0]
        union = union + cnt[1]
    return inter, union


def get_IoU(preds, gt, thres=None):
    """modules/volutils.py:74-76."""
    intersection, union = get_I_and_U(preds, gt, thres)
    return intersection / union


def get_IoU_batch(preds, gt, thres=None, maxpoints=pow(2, 24)):
    """modules/volutils.py:54-71: IoU accumulated over slabs of ``maxpoints`` voxels."""
    preds, gt = preds.flatten(), gt.flatten()
    inter, union = [], []
    for b in range(0, preds.numel(), maxpoints):
        i, u = get_I_and_U(preds[b:b + maxpoints], gt[b:b + maxpoints], thres)
        inter.append(i)
        union.append(u)
    return sum(inter) / sum(union)


def query_occupancy(coords, cube_res, model, batchsize, occupancy=None):
    """The dense query of ``export_mesh`` (modules/volutils.py:113-133): ``model`` is evaluated on ``coords``
    ((cube_res^3, 3), host or device) in batches of ``batchsize``, ``torch.sigmoid`` of the output fills a
    (cube_res, cube_res, cube_res) float32 numpy cube -- what the reference then hands to ``mcubes.marching_cubes`` /
    open3d (out of scope here).  The network forward is the fused HIP path; the sigmoid is wire_sigmoid_inplace."""
    n = cube_res ** 3
    if occupancy is None:
        occupancy = np.zeros((n, 1), dtype=np.float32)
    else:
        occupancy[...] = 0
        occupancy = occupancy.reshape(-1, 1)
    L = _lib.lib()
    with torch.no_grad():
        for b in range(0, n, batchsize):
            b2 = min(b + batchsize, n)
            sub = coords[b:b2, :].cuda()
            y = model(sub).contiguous()
            _lib.check(L.wire_sigmoid_inplace(torch.cuda.current_stream(y.device).cuda_stream, y.data_ptr(), y.numel()),
                       "wire_sigmoid_inplace")
            occupancy[b:b2, :] = y.cpu().numpy()
    return occupancy.reshape(cube_res, cube_res, cube_res)
