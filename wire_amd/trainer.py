"""Sync-free fused training step for the image-fit / occupancy loops.

Restates the inner loop of the reference's drivers
(wire_image_denoise.py:142-157, wire_occupancy.py:137-158):

    b_coords = coords[b_indices]; pix = model(b_coords); rec[b_indices] = pix
    loss = ((pix - gt[b_indices])**2).mean(); zero_grad; backward; Adam.step

as a fixed sequence of libwire_hip launches on one HIP stream with no host
synchronisation: coordinates are generated on the device from the flat indices
(per-axis linspace tables, bit-identical to the reference's grids), the MSE
gradient, the whole backward and a flat Adam update (complex parameters as real
pairs, exactly torch.optim.Adam's arithmetic) are HIP kernels.  The host gather
+ H2D copy and the per-step ``loss.item()`` of the reference are gone.

Multi-GPU: one process per GPU; the global batch is sharded contiguously
(parallel.shard_bounds), each shard's gradient is pre-scaled by n_g/B and the flat
gradient buffer is all-reduced (SUM) over RCCL: in one piece on the compute stream
after the backward (default), layer by layer on a side stream as soon as the backward
has produced each slice (WIRE_DP_OVERLAP=layer, wire_train_fwd_bwd_hooked), or -- with
``micro_shards > 1`` -- micro-shard i's under the forward and backward of micro-shard i+1.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib
from .modules._base import HipINR
from .modules.utils import axis_tables
from .parallel import FlatGradAllReducer, replicas_identical, shard_bounds


class FusedTrainer:
    def __init__(self, model: HipINR, grid: Sequence[int], target: torch.Tensor, lr: float = 5e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, gamma: float = 0.1, niters: int = 2000,
                 coords_style: str = "torch", keep_rec: bool = False, micro_shards: int = 1,
                 group: Optional[dist.ProcessGroup] = None):
        self.L = _lib.lib()
        # the fused step runs net = first layer, L hidden layers, final LINEAR with omega / scale taken from the
        # descriptor: a net that HipINR.forward runs layer by layer is a different function here
        if getattr(model, "_layerwise", False):
            raise NotImplementedError("FusedTrainer needs outermost_linear=True: this net ends in an activation layer "
                                      "(modules/siren.py:81-84, gauss.py:63-66, relu.py:116-119) and runs layer by layer; "
                                      "train it through model(coords) + torch.optim")
        if any(getattr(m, "trainable", False) for m in model.net):
            raise NotImplementedError("FusedTrainer keeps omega_0 / scale_0 fixed (they are not in its flat parameter "
                                      "buffer); a net with trainable=True layers (modules/wire.py:80-81) trains through "
                                      "model(coords) + torch.optim")
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise _lib.WireHipError("FusedTrainer needs the model on an MI355X ('cuda')")
        self.model = model
        self.dev = p0.device
        self.desc = model.net_desc()
        self.grid = tuple(int(g) for g in grid)
        if len(self.grid) not in (2, 3) or len(self.grid) != self.desc.in_features:
            raise ValueError("grid must be (H, W) for 2 inputs or (H, W, T) for 3")
        H, W = self.grid[0], self.grid[1]
        T = self.grid[2] if len(self.grid) == 3 else None
        tx, ty, tz = axis_tables(H, W, T, style=coords_style)
        self.tx, self.ty = tx.to(self.dev), ty.to(self.dev)
        self.tz = tz.to(self.dev) if tz is not None else None
        self.npoints = H * W * (T or 1)
        self.O = self.desc.out_features
        self.target = target.to(self.dev, torch.float32).reshape(self.npoints, self.O).contiguous()
        self.rec = torch.zeros_like(self.target) if keep_rec else None

        # ---- flat parameter / gradient / Adam-state buffers; parameters become views
        tensors = model.param_tensors()
        self.sizes = [int(_lib.check(self.L.wire_param_tensor_floats(C.byref(self.desc), i)))
                      for i in range(len(tensors))]
        # every tensor starts on a 16-byte boundary (complex views need an even
        # float offset); the pad floats stay zero in params, grads and Adam state
        padded = [(sz + 3) // 4 * 4 for sz in self.sizes]
        self.count = sum(padded)
        self.flat = torch.zeros(self.count, dtype=torch.float32, device=self.dev)
        off = 0
        self.offsets: List[int] = []
        with torch.no_grad():
            for t, sz in zip(tensors, self.sizes):
                src = torch.view_as_real(t.detach()).reshape(-1) if t.is_complex() else t.detach().reshape(-1)
                if src.numel() != sz:
                    raise _lib.WireHipError("parameter size does not match the ABI's layout")
                view = self.flat[off:off + sz]
                view.copy_(src)
                t.data = torch.view_as_complex(view.view(*t.shape, 2)) if t.is_complex() \
                    else view.view(t.shape)
                self.offsets.append(off)
                off += (sz + 3) // 4 * 4
        self.micro = max(1, int(micro_shards))
        # +1: the loss rides at the end of each gradient buffer through the all-reduce
        self.gbuf = [torch.zeros(self.count + 1, dtype=torch.float32, device=self.dev)
                     for _ in range(self.micro)]
        self.exp_avg = torch.zeros(self.count, dtype=torch.float32, device=self.dev)
        self.exp_avg_sq = torch.zeros(self.count, dtype=torch.float32, device=self.dev)
        fp = self.flat.data_ptr()
        self.param_ptrs = _lib.ptr_array([fp + 4 * o for o in self.offsets])
        self.grad_ptrs = [_lib.ptr_array([g.data_ptr() + 4 * o for o in self.offsets]) for g in self.gbuf]
        self.packed = torch.empty(self.L.wire_packed_floats(C.byref(self.desc)), dtype=torch.float32,
                                  device=self.dev)
        self.partial = torch.empty(4096, dtype=torch.float32, device=self.dev)

        self.base_lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.gamma, self.niters = float(gamma), int(niters)
        self.epoch = 0
        self.t = 0
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # one micro-shard: nothing to overlap the collective with, so it is issued from the compute stream
        # (two cross-stream hand-offs fewer per step); several: a side stream overlaps it with the next shard
        side = self.micro > 1 and os.environ.get("WIRE_DP_SIDE_STREAM", "1") != "0" or \
            os.environ.get("WIRE_DP_SIDE_STREAM", "") == "1"
        # WIRE_DP_OVERLAP=layer (one micro-shard): PER-LAYER overlap -- wire_train_fwd_bwd_hooked announces each layer's
        # gradient as soon as its last kernel is enqueued, and that slice of the flat buffer is all-reduced on a side stream
        # while the compute stream runs the backward of the layers below it; only the first layer's (smallest, last) slice is
        # exposed.  Default "none": one all-reduce of the whole 2.1 MB buffer after the backward -- measured on the one-rank
        # communicator (tools/dp_overhead.sh) the six cross-stream hand-offs of "layer" cost 0.07 - 0.09 ms per step, as
        # much as the whole latency-bound all-reduce they would hide.
        self.overlap = self.micro == 1 and os.environ.get("WIRE_DP_OVERLAP", "none") == "layer"
        self.reducers = [FlatGradAllReducer(g, group, use_side_stream=side or self.overlap) for g in self.gbuf]
        # (host-staged exchange -- the gloo rehearsal with device gradients -- would block inside the announcement callback,
        #  which must not wait for the device (include/wire_hip.h): one reduction after the backward instead)
        self.overlap = self.overlap and self.reducers[0].active and not self.reducers[0].stage_host
        self._cb_err: Optional[BaseException] = None
        # the C callback only exists with the per-layer overlap, and holds the trainer weakly: no reference cycle, so
        # `del trainer` releases the act / scratch / gradient buffers at once
        self._ready_cb = None
        if self.overlap:
            import weakref
            ref = weakref.ref(self)

            def _cb(user, first, n):
                tr = ref()
                if tr is not None:
                    tr._on_grad_ready(user, first, n)
            self._ready_cb = _lib.GRAD_READY_FN(_cb)
        # WIRE_DP_CHECK=k: every k optimizer steps the replicas compare a checksum of their parameters (one host sync);
        # a divergence raises instead of training on silently
        self.check_every = max(0, int(os.environ.get("WIRE_DP_CHECK", "0")))
        self._cap = 0
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self._hidx: Optional[torch.Tensor] = None
        if self.world > 1:
            # replicas start from rank 0's parameters whatever seed each rank was built under; they stay
            # identical because every rank applies the same Adam update to the same reduced gradient
            self._broadcast_from_rank0(self.flat)

    def _broadcast_from_rank0(self, t: torch.Tensor) -> None:
        if dist.get_backend(self.group) == "gloo" and t.is_cuda:     # rehearsal backend: stage through the host
            host = t.detach().cpu()
            dist.broadcast(host, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)
            t.copy_(host)
        else:
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)

    def _ready_order(self):
        """(first_tensor, n_tensors) in the order wire_train_fwd_bwd_hooked announces them (include/wire_hip.h)."""
        nt = len(self.offsets)
        hidden = int(self.desc.hidden_layers)
        per = (nt - 2) // (hidden + 1)
        yield nt - 2, 2
        for l in range(hidden, 0, -1):
            yield per * l, per
        yield 0, per

    def _on_grad_ready(self, user, first: int, n: int) -> None:
        """wire_grad_ready_fn: parameter tensors [first, first + n) have their final gradient enqueued on the compute
        stream -> start the all-reduce of their slice of the flat buffer (the loss rides behind the last tensor)."""
        try:
            lo = self.offsets[first]
            hi = self.offsets[first + n] if first + n < len(self.offsets) else self.count + 1
            self.reducers[0].launch_range(lo, hi)
        except BaseException as e:                       # noqa: BLE001 -- must not propagate through the C frame
            self._cb_err = e

    # ------------------------------------------------------------------ buffers
    def _reserve(self, n: int) -> None:
        if n <= self._cap:
            return
        d = C.byref(self.desc)
        self.act_bytes = _lib.check(self.L.wire_act_bytes(d, n, 1))
        self.scr_bytes = _lib.check(self.L.wire_bwd_scratch_bytes(d, n))
        self.act = torch.empty(self.act_bytes, dtype=torch.uint8, device=self.dev)
        self.scratch = torch.empty(self.scr_bytes, dtype=torch.uint8, device=self.dev)
        self.coords = torch.empty(n * self.desc.in_features, dtype=torch.float32, device=self.dev)
        self.y = torch.empty(n * self.O, dtype=torch.float32, device=self.dev)
        self.gy = torch.empty(n * self.O, dtype=torch.float32, device=self.dev)
        self._cap = n

    # ------------------------------------------------------------------ permutations
    def permutation(self, n: Optional[int] = None) -> torch.Tensor:
        """torch.randperm(n) for the next epoch (wire_image_denoise.py:142), generated on a side
        stream while the current step computes: returns the permutation prepared by the previous
        call (made visible to the compute stream) and starts the next one.  Same generator, same
        sequence as calling torch.randperm(n, device=...) in a loop."""
        n = self.npoints if n is None else int(n)
        cur = torch.cuda.current_stream(self.dev)
        if getattr(self, "_perm_stream", None) is None:
            self._perm_stream = torch.cuda.Stream(device=self.dev)
            self._perm_next = None
        if self._perm_next is None or self._perm_next.numel() != n:
            with torch.cuda.stream(self._perm_stream):
                self._perm_next = torch.randperm(n, device=self.dev)
        cur.wait_stream(self._perm_stream)
        out = self._perm_next
        out.record_stream(cur)
        with torch.cuda.stream(self._perm_stream):
            self._perm_next = torch.randperm(n, device=self.dev)
        return out

    # ------------------------------------------------------------------ schedule
    def current_lr(self) -> float:
        """LambdaLR(lambda x: gamma**min(x/niters, 1)) (wire_image_denoise.py:128)."""
        return self.base_lr * self.gamma ** min(self.epoch / self.niters, 1)

    def scheduler_step(self) -> None:
        self.epoch += 1

    # ------------------------------------------------------------------ one step
    def step(self, indices: Optional[torch.Tensor] = None, first: int = 0,
             count: Optional[int] = None) -> torch.Tensor:
        """One optimizer step on a global batch: rows ``indices`` (contiguous int64 device
        tensor of flat grid indices, every rank passes the same one) or the range
        [first, first+count).  Returns the (device-resident, all-reduced) batch loss; no host sync."""
        if indices is not None:
            if indices.dtype != torch.int64 or not indices.is_cuda or indices.dim() != 1:
                raise ValueError("indices must be a 1-D CUDA int64 tensor")
            if not indices.is_contiguous():
                raise ValueError("indices must be contiguous (a strided view such as perm[::2] would be read "
                                 "as its underlying storage); call .contiguous()")
            B = indices.numel()
            if os.environ.get("WIRE_CHECK_INDICES", "0") == "1" and B > 0:
                # opt-in (one host sync): the kernels trust the indices, an out-of-range one reads / writes
                # outside the target and rec buffers
                lo_i, hi_i = int(indices.min()), int(indices.max())
                if lo_i < 0 or hi_i >= self.npoints:
                    raise ValueError(f"indices span [{lo_i}, {hi_i}] outside the grid of {self.npoints} points")
        else:
            B = int(count if count is not None else self.npoints - first)
            if first < 0 or first + B > self.npoints:
                raise ValueError(f"rows [{first}, {first + B}) outside the grid of {self.npoints} points")
        lo, hi = shard_bounds(B, self.world, self.rank)
        idx_ptr = (indices.data_ptr() + 8 * lo) if indices is not None else None
        return self._step_local(idx_ptr, first + lo, hi - lo, B)

    def step_hashed(self, seed: int, first: int = 0, count: Optional[int] = None) -> torch.Tensor:
        """One optimizer step on positions [first, first+count) of the epoch's shuffle ``pi_seed`` -- the
        reference's ``indices = torch.randperm(H*W); b_indices = indices[b_idx:b_idx+maxpoints]``
        (wire_image_denoise.py:142-146, wire_occupancy.py:137-142) with the permutation evaluated per position on
        the device (wire_perm_indices): this rank generates only ITS shard of the batch, so the cost of the
        shuffle does not grow with the world size or with the grid (512^3: no 1 GB index vector).  ``seed`` is
        the epoch counter; every rank passes the same (seed, first, count).  No host sync."""
        B = int(count if count is not None else self.npoints - first)
        if first < 0 or first + B > self.npoints:
            raise ValueError(f"positions [{first}, {first + B}) outside the epoch of {self.npoints} points")
        lo, hi = shard_bounds(B, self.world, self.rank)
        nloc = hi - lo
        if self._hidx is None or self._hidx.numel() < nloc:
            self._hidx = torch.empty(max(nloc, 1), dtype=torch.int64, device=self.dev)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(self.L.wire_perm_indices(stream, int(seed) & 0xFFFFFFFFFFFFFFFF, self.npoints, first + lo, nloc,
                                            self._hidx.data_ptr()), "perm_indices")
        return self._step_local(self._hidx.data_ptr(), 0, nloc, B)

    def _step_local(self, idx_ptr: Optional[int], first: int, nloc: int, B: int) -> torch.Tensor:
        """This rank's shard of a global batch of B rows: ``nloc`` rows whose flat grid indices are the int64
        device array at ``idx_ptr`` (or the range starting at ``first`` when None)."""
        L, d = self.L, C.byref(self.desc)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(L.wire_pack_params(stream, d, self.param_ptrs, self.packed.data_ptr()), "pack")
        per = (nloc + self.micro - 1) // self.micro
        self._reserve(max(per, 1))
        tz_ptr = self.tz.data_ptr() if self.tz is not None else None
        Tn = self.grid[2] if self.tz is not None else 1
        for m in range(self.micro):
            mlo = m * per
            n = max(0, min(nloc, mlo + per) - mlo)
            g = self.gbuf[m]
            if n == 0:
                g.zero_()
                if self.overlap:                         # the same slices in the same order as the ranks that have rows
                    for t0, nt in self._ready_order():
                        self._on_grad_ready(None, t0, nt)
                    if self._cb_err is not None:
                        raise self._cb_err
                    continue
            else:
                ip = (idx_ptr + 8 * mlo) if idx_ptr is not None else None
                _lib.check(L.wire_coords_from_index(stream, ip, first + mlo, n, self.tx.data_ptr(),
                                                    self.grid[1], self.ty.data_ptr(), self.grid[0],
                                                    tz_ptr, Tn, self.coords.data_ptr()), "coords")
                args = (stream, d, self.packed.data_ptr(), self.coords.data_ptr(), n, self.target.data_ptr(),
                        ip, first + mlo, n / float(B), self.y.data_ptr(), self.gy.data_ptr(),
                        g.data_ptr() + 4 * self.count,
                        self.rec.data_ptr() if self.rec is not None else None, self.partial.data_ptr(),
                        self.act.data_ptr(), self.act_bytes, self.scratch.data_ptr(), self.scr_bytes,
                        self.grad_ptrs[m])
                if self.overlap:
                    self._cb_err = None
                    _lib.check(L.wire_train_fwd_bwd_hooked(*args, self._ready_cb, None), "train_fwd_bwd")
                    if self._cb_err is not None:
                        raise self._cb_err
                    continue                             # every slice is on its way: wait() below joins the streams
                _lib.check(L.wire_train_fwd_bwd(*args), "train_fwd_bwd")
            self.reducers[m].launch()
        for r in self.reducers:
            r.wait()
        gsum = self.gbuf[0]
        for m in range(1, self.micro):
            gsum.add_(self.gbuf[m])
        self.t += 1
        _lib.check(L.wire_adam_step_flat(stream, self.flat.data_ptr(), gsum.data_ptr(),
                                         self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.count,
                                         self.current_lr(), self.betas[0], self.betas[1], self.eps,
                                         self.t), "adam")
        self.loss = gsum[self.count:self.count + 1].clone()   # the buffer is reused next step
        if self.check_every and self.world > 1 and self.t % self.check_every == 0:
            if not replicas_identical(self.flat, self.group):
                raise RuntimeError(f"data-parallel replicas diverged at optimizer step {self.t} (WIRE_DP_CHECK)")
        return self.loss

    # ------------------------------------------------------------------ super-resolution step
    def step_downsampled(self, gt_lr: torch.Tensor, scale: int,
                         rec_lr: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimizer step of the super-resolution loop (wire_SISR.py:151-178): the model is
        evaluated on the WHOLE high-resolution grid in raster order, ``torch.nn.AvgPool2d(scale)``
        brings the reconstruction to the resolution of ``gt_lr`` ([H//scale * W//scale, O], device
        float32), loss = mean squared error there; backward through the pooling and the network,
        Adam.  The high-resolution reconstruction of this step stays in ``self.y`` ([H*W, O]).
        2-D grids, single process (a pooled window must not straddle two shards)."""
        if self.tz is not None or len(self.grid) != 2:
            raise ValueError("step_downsampled needs a 2-D grid")
        if self.world != 1:
            raise NotImplementedError("step_downsampled is single-process")
        H, W = int(self.grid[0]), int(self.grid[1])
        scale = int(scale)
        if scale < 1 or scale > min(H, W):
            raise ValueError(f"scale {scale} outside 1..min(H, W)")
        H2, W2 = H // scale, W // scale
        gt = gt_lr.detach()
        if not gt.is_cuda or gt.dtype != torch.float32 or gt.numel() != H2 * W2 * self.O:
            raise ValueError(f"gt_lr must be a CUDA float32 tensor of {H2 * W2} x {self.O} elements")
        gt = gt.contiguous()
        if rec_lr is not None and (not rec_lr.is_cuda or rec_lr.dtype != torch.float32
                                   or rec_lr.numel() != gt.numel() or not rec_lr.is_contiguous()):
            raise ValueError("rec_lr must be a contiguous CUDA float32 tensor shaped like gt_lr")
        L, d = self.L, C.byref(self.desc)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        n = H * W
        self._reserve(n)
        g = self.gbuf[0]
        _lib.check(L.wire_pack_params(stream, d, self.param_ptrs, self.packed.data_ptr()), "pack")
        _lib.check(L.wire_coords_from_index(stream, None, 0, n, self.tx.data_ptr(), W, self.ty.data_ptr(), H,
                                            None, 1, self.coords.data_ptr()), "coords")
        _lib.check(L.wire_mlp_fwd(stream, d, self.packed.data_ptr(), self.coords.data_ptr(), n,
                                  self.y.data_ptr(), self.act.data_ptr(), self.act_bytes, 1), "fwd")
        _lib.check(L.wire_avgpool_mse_grad(stream, self.y.data_ptr(), H, W, self.O, scale, gt.data_ptr(),
                                           self.gy.data_ptr(), rec_lr.data_ptr() if rec_lr is not None else None,
                                           g.data_ptr() + 4 * self.count, self.partial.data_ptr()), "avgpool_mse_grad")
        _lib.check(L.wire_mlp_bwd(stream, d, self.packed.data_ptr(), self.coords.data_ptr(), n,
                                  self.gy.data_ptr(), self.act.data_ptr(), self.act_bytes,
                                  self.scratch.data_ptr(), self.scr_bytes, self.grad_ptrs[0]), "bwd")
        self.t += 1
        _lib.check(L.wire_adam_step_flat(stream, self.flat.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                         self.exp_avg_sq.data_ptr(), self.count, self.current_lr(),
                                         self.betas[0], self.betas[1], self.eps, self.t), "adam")
        self.loss = g[self.count:self.count + 1].clone()
        return self.loss

    # ------------------------------------------------------------------ CT step
    def step_radon(self, sinogram: torch.Tensor, thetas: torch.Tensor) -> torch.Tensor:
        """One optimizer step of the CT loop (wire_ct.py:128-139): the model is evaluated on the WHOLE grid in raster
        order (``img_estim = model(coords).reshape(-1, H, W)``), ``lin_inverse.radon`` turns it into a sinogram
        ([nangles, W]), loss = mean squared error against ``sinogram``; backward through the adjoint Radon kernel
        and the network, Adam.  The image estimate of this step stays in ``self.y`` ([H*W, 1]).  2-D grids, one output
        feature, single process."""
        if self.tz is not None or len(self.grid) != 2 or self.O != 1:
            raise ValueError("step_radon needs a 2-D grid and out_features == 1")
        if self.world != 1:
            raise NotImplementedError("step_radon is single-process")
        H, W = int(self.grid[0]), int(self.grid[1])
        th = thetas.detach().to(self.dev, torch.float32).contiguous()
        A = th.numel()
        tgt = sinogram.detach()
        if not tgt.is_cuda or tgt.dtype != torch.float32 or tgt.numel() != A * W:
            raise ValueError(f"sinogram must be a CUDA float32 tensor of {A} x {W} elements")
        tgt = tgt.contiguous()
        L, d = self.L, C.byref(self.desc)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        n = H * W
        self._reserve(n)
        if getattr(self, "_sino", None) is None or self._sino.numel() != A * W:
            self._sino = torch.empty(A * W, dtype=torch.float32, device=self.dev)
            self._gsino = torch.empty(A * W, dtype=torch.float32, device=self.dev)
        g = self.gbuf[0]
        _lib.check(L.wire_pack_params(stream, d, self.param_ptrs, self.packed.data_ptr()), "pack")
        _lib.check(L.wire_coords_from_index(stream, None, 0, n, self.tx.data_ptr(), W, self.ty.data_ptr(), H,
                                            None, 1, self.coords.data_ptr()), "coords")
        _lib.check(L.wire_mlp_fwd(stream, d, self.packed.data_ptr(), self.coords.data_ptr(), n,
                                  self.y.data_ptr(), self.act.data_ptr(), self.act_bytes, 1), "fwd")
        _lib.check(L.wire_radon_fwd(stream, self.y.data_ptr(), th.data_ptr(), H, W, A, self._sino.data_ptr()), "radon")
        _lib.check(L.wire_mse_grad(stream, self._sino.data_ptr(), tgt.data_ptr(), None, 0, A * W, 1, 1.0,
                                   self._gsino.data_ptr(), g.data_ptr() + 4 * self.count, None,
                                   self.partial.data_ptr()), "mse_grad")
        _lib.check(L.wire_radon_bwd(stream, self._gsino.data_ptr(), th.data_ptr(), H, W, A, self.gy.data_ptr()),
                   "radon_bwd")
        _lib.check(L.wire_mlp_bwd(stream, d, self.packed.data_ptr(), self.coords.data_ptr(), n,
                                  self.gy.data_ptr(), self.act.data_ptr(), self.act_bytes,
                                  self.scratch.data_ptr(), self.scr_bytes, self.grad_ptrs[0]), "bwd")
        self.t += 1
        _lib.check(L.wire_adam_step_flat(stream, self.flat.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                         self.exp_avg_sq.data_ptr(), self.count, self.current_lr(),
                                         self.betas[0], self.betas[1], self.eps, self.t), "adam")
        self.loss = g[self.count:self.count + 1].clone()
        return self.loss

    @property
    def flat_grad(self) -> torch.Tensor:
        return self.gbuf[0][:self.count]

    # ------------------------------------------------------------------ metrics
    def psnr(self, rec: torch.Tensor, gt: Optional[torch.Tensor] = None) -> torch.Tensor:
        """utils.psnr(gt, rec) = 10 log10(max(gt) / mse) (modules/utils.py:67-82) computed on the
        device; returns a 0-dim device tensor (no host sync, no image copy)."""
        out = self._metric(0, rec, self.target if gt is None else gt, 0.0)
        return 10.0 * torch.log10(out[1] / (out[0] / rec.numel()))

    def iou(self, pred: torch.Tensor, gt: Optional[torch.Tensor] = None, thres: float = 0.5) -> torch.Tensor:
        """volutils.get_IoU(pred, gt, thres) (modules/volutils.py:74-91) on the device; ``pred`` is
        left untouched (the reference binarises it in place)."""
        out = self._metric(1, pred, self.target if gt is None else gt, thres)
        return out[0] / out[1]

    def _metric(self, mode: int, rec: torch.Tensor, gt: torch.Tensor, thres: float) -> torch.Tensor:
        rec = rec.detach().to(torch.float32).contiguous()
        gt = gt.detach().to(self.dev, torch.float32).contiguous()
        if rec.numel() != gt.numel():
            raise ValueError("metric operands differ in size")
        out = torch.empty(2, dtype=torch.float32, device=self.dev)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(self.L.wire_eval_metric(stream, mode, rec.data_ptr(), gt.data_ptr(), rec.numel(), thres,
                                           out.data_ptr(), self.partial.data_ptr()), "metric")
        return out

    # ------------------------------------------------------------------ best-so-far tracking
    def update_best(self, metric: torch.Tensor, image: torch.Tensor, force: bool = False) -> None:
        """Device-side restatement of the drivers' best-result bookkeeping -- ``if (mse_array[epoch] < best_mse) or
        (epoch == 0): best_mse = mse_array[epoch]; best_img = imrec`` (wire_image_denoise.py:176-178; pass
        ``force=(epoch == 0)``) and ``if lossval < best_mse: ...; best_img = copy.deepcopy(im_estim)``
        (wire_occupancy.py:170-172): ``metric`` is a 1-element device tensor (an MSE, ``tr.loss`` ...), ``image`` the
        reconstruction (``tr.rec``, a render).  No ``.item()``, no per-epoch copy of the image to the host: a
        compare-and-copy kernel keeps ``self.best_metric`` / ``self.best_img`` current."""
        img = image.detach()
        if not img.is_cuda or img.dtype != torch.float32 or not img.is_contiguous():
            raise ValueError("image must be a contiguous CUDA float32 tensor")
        m = metric.detach().reshape(-1)
        if not m.is_cuda or m.dtype != torch.float32 or m.numel() != 1:
            raise ValueError("metric must be a 1-element CUDA float32 tensor")
        if getattr(self, "best_img", None) is None or self.best_img.numel() != img.numel():
            self.best_img = torch.zeros_like(img)
            self.best_metric = torch.full((1,), float("inf"), dtype=torch.float32, device=self.dev)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(self.L.wire_track_best(stream, m.data_ptr(), self.best_metric.data_ptr(), int(bool(force)),
                                          img.data_ptr(), self.best_img.data_ptr(), img.numel(), None), "track_best")

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def render(self, first: int = 0, count: Optional[int] = None, tile: int = 1 << 20,
               sigmoid: bool = False) -> torch.Tensor:
        """Forward-only dense query of grid rows [first, first+count) in tiles
        (no saved activations): the reference's full-image / volume evaluation.  ``sigmoid=True`` applies
        ``torch.sigmoid`` to the result, the occupancy cube ``export_mesh`` hands to marching cubes
        (modules/volutils.py:124-133; the meshing itself is out of scope)."""
        L, d = self.L, C.byref(self.desc)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        count = int(count if count is not None else self.npoints - first)
        out = torch.empty(count, self.O, dtype=torch.float32, device=self.dev)
        _lib.check(L.wire_pack_params(stream, d, self.param_ptrs, self.packed.data_ptr()), "pack")
        tile = min(tile, max(count, 1))
        ab = _lib.check(L.wire_act_bytes(d, tile, 0))
        act = torch.empty(ab, dtype=torch.uint8, device=self.dev)
        coords = torch.empty(tile * self.desc.in_features, dtype=torch.float32, device=self.dev)
        tz_ptr = self.tz.data_ptr() if self.tz is not None else None
        Tn = self.grid[2] if self.tz is not None else 1
        for s in range(0, count, tile):
            n = min(tile, count - s)
            _lib.check(L.wire_coords_from_index(stream, None, first + s, n, self.tx.data_ptr(), self.grid[1],
                                                self.ty.data_ptr(), self.grid[0], tz_ptr, Tn,
                                                coords.data_ptr()), "coords")
            _lib.check(L.wire_mlp_fwd(stream, d, self.packed.data_ptr(), coords.data_ptr(), n,
                                      out.data_ptr() + 4 * s * self.O, act.data_ptr(), ab, 0), "fwd")
        if sigmoid:
            _lib.check(L.wire_sigmoid_inplace(stream, out.data_ptr(), out.numel()), "sigmoid")
        return out
