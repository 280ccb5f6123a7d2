"""Coordinate-batch data parallelism: one process per GPU, RCCL over xGMI.

The reference is single-GPU.  The path shards
naturally: every coordinate is an independent sample and the parameters
(<= 2.1 MB) are replicated, so the only exchange per optimizer step is one
SUM all-reduce of the flat fp32 gradient buffer (complex gradients travel as
real pairs).  The loss of the reference is a mean over the batch
(wire_image_denoise.py:153, wire_occupancy.py:150), so shard g of a global batch
of B rows scales its local-mean gradient by n_g / B before the sum -- exact for
the ragged tail batch too (wire_occupancy.py:142).

This module is compute-agnostic (it never touches the HIP library), which is
what lets tests/test_parallel_gloo.py run it with world_size 2 on CPU.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal split of ``batch`` rows: rank r gets
    [lo, hi).  The first ``batch % world`` ranks get one extra row."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(int(batch), world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_weight(batch: int, world: int, rank: int) -> float:
    """n_g / B: factor that turns the shard's local-mean loss gradient into its
    share of the global-mean gradient."""
    lo, hi = shard_bounds(batch, world, rank)
    return (hi - lo) / float(batch) if batch > 0 else 0.0


class FlatGradAllReducer:
    """All-reduce(SUM) of a flat gradient buffer, optionally in buckets issued
    on a side stream so that the collective of bucket i overlaps whatever the
    compute stream does next (the next micro-shard's forward/backward).

    The buffer is tiny (1-2 MB), so the collective is latency-bound on xGMI;
    a single bucket is the default.  ``group=None`` with an uninitialised
    process group degrades to a no-op (world size 1).
    """

    def __init__(self, flat: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                 bucket_floats: Optional[int] = None, use_side_stream: bool = True):
        self.flat = flat
        self.group = group
        # WIRE_DP_FORCE=1 keeps the collective path live with a single rank (rehearsal of the RCCL
        # stream plumbing on a one-GPU box)
        import os
        force = os.environ.get("WIRE_DP_FORCE", "0") == "1"
        self.active = dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size(group) > 1 or force)
        n = flat.numel()
        if not bucket_floats or bucket_floats >= n:
            self.buckets: List[Tuple[int, int]] = [(0, n)]
        else:
            self.buckets = [(o, min(n, o + bucket_floats)) for o in range(0, n, bucket_floats)]
        self.stream = None
        # gloo cannot reduce device tensors: stage through pinned host memory (rehearsal only;
        # the production backend is RCCL, which reduces the device buffer in place)
        self.stage_host = bool(self.active and flat.is_cuda and dist.get_backend(group) == "gloo")
        if self.active and flat.is_cuda and use_side_stream and not self.stage_host:
            self.stream = torch.cuda.Stream(device=flat.device)
        self._pending: List = []

    def launch(self, tensor: Optional[torch.Tensor] = None) -> None:
        """Start the all-reduce of ``tensor`` (default: the flat buffer)."""
        if not self.active:
            return
        t = self.flat if tensor is None else tensor
        if self.stage_host:
            host = t.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(t.device))
            with torch.cuda.stream(self.stream):
                for lo, hi in self.buckets:
                    self._pending.append(dist.all_reduce(t[lo:hi], op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
        else:
            for lo, hi in self.buckets:
                self._pending.append(dist.all_reduce(t[lo:hi], op=dist.ReduceOp.SUM,
                                                     group=self.group, async_op=True))

    def wait(self) -> None:
        """Make the compute stream (or the host, on CPU) wait for the collective."""
        if not self.active:
            return
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)

    def __call__(self) -> None:
        self.launch()
        self.wait()
