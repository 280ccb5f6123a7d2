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

import ctypes as C
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class RcclDirect:
    """ncclAllReduce of librccl.so's C ABI issued ON THE CALLER'S HIP STREAM.

    torch.distributed's NCCL process group runs every collective on a stream of its own and brackets it with events
    against the current stream -- two cross-stream hand-offs per all-reduce, measured at ~3 % of a training step
    (tools/dp_overhead.sh).  The gradient all-reduce of a step has nothing to overlap with when it is the step's last
    work item before Adam, so here it is simply the next operation of the compute stream: rank 0 draws a
    ``ncclUniqueId``, torch.distributed (whatever its backend) only broadcasts those 128 bytes, and every rank joins
    a communicator of its own with ``ncclCommInitRank`` on its current device.  The library is the librccl.so that
    torch itself loaded (torch/lib), so both share one HIP runtime.  xGMI transport, ring / tree choice and the
    reduction order (identical on every rank -> replicas stay bit-identical) are RCCL's.
    """

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            l = C.CDLL(path)
            l.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
            l.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
            l.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            l.ncclCommDestroy.argtypes = [C.c_void_p]
            l.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            l.ncclCommCuDevice.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            l.ncclGetErrorString.restype = C.c_char_p
            l.ncclGetErrorString.argtypes = [C.c_int]
            cls._lib = l
        return cls._lib

    def __init__(self, device: torch.device, group: Optional[dist.ProcessGroup] = None):
        """Collective: every rank of ``group`` must call this.  The set-up is SYMMETRIC: every rank sends and
        receives the same messages whatever fails where --
          1. each rank loads librccl and checks its device locally; the outcomes are agreed on (MIN all-reduce);
          2. rank 0 draws the unique id and ALWAYS broadcasts status + 128 id bytes, also when drawing it failed;
          3. only if every rank is still fine do all of them enter ``ncclCommInitRank`` (itself a rendezvous), and its
             outcome is agreed on again;
        any failure raises the same RuntimeError on every rank, so the caller's fallback is taken by all of them."""
        self.comm = C.c_void_p()
        self.device = device
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        on_dev = dist.get_backend(group) == "nccl"
        src = dist.get_global_rank(group, 0) if group is not None else 0

        def agree(ok: bool, what: str) -> None:
            flag = torch.tensor([1.0 if ok else 0.0], device=device if on_dev else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if float(flag.item()) < 1.0:
                raise RuntimeError(f"direct RCCL communicator: {what} failed on at least one rank")

        err = None
        l = None
        try:
            l = self.lib()
            if device.type != "cuda":
                raise RuntimeError("not a GPU device")
        except Exception as e:                              # noqa: BLE001
            err = e
        agree(err is None, f"loading librccl ({err})")
        uid = _NcclUniqueId()
        msg = torch.zeros(129, dtype=torch.uint8)
        if self.rank == 0:
            rc = l.ncclGetUniqueId(C.byref(uid))
            if rc == 0:
                msg[0] = 1
                msg[1:] = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8)
        # status + 128 bytes through the existing process group (device tensor for nccl, host tensor for gloo)
        t = msg.to(device) if on_dev else msg
        dist.broadcast(t, src=src, group=group)
        t = t.cpu()
        if int(t[0]) != 1:
            raise RuntimeError("direct RCCL communicator: ncclGetUniqueId failed on rank 0")
        C.memmove(C.byref(uid), bytes(t[1:].numpy().tobytes()), 128)
        with torch.cuda.device(device):
            rc = l.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
        try:
            agree(rc == 0, "ncclCommInitRank")
        except RuntimeError:
            if rc == 0:
                self.close()
            raise

    def comm_count(self) -> int:
        """Number of ranks RCCL itself reports for this communicator (ncclCommCount)."""
        n = C.c_int(-1)
        self._check(self.lib().ncclCommCount(self.comm, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def comm_device(self) -> int:
        """HIP device RCCL itself reports for this communicator (ncclCommCuDevice)."""
        d = C.c_int(-1)
        self._check(self.lib().ncclCommCuDevice(self.comm, C.byref(d)), "ncclCommCuDevice")
        return int(d.value)

    @staticmethod
    def _check(rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what} failed: {RcclDirect.lib().ncclGetErrorString(rc).decode()} ({rc})")

    def all_reduce_sum_(self, t: torch.Tensor) -> None:
        """In-place SUM over the ranks of a contiguous float32 device tensor, on torch's current stream."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("RcclDirect reduces contiguous float32 device tensors")
        stream = torch.cuda.current_stream(t.device).cuda_stream
        self._check(self.lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), 7, 0, self.comm, stream),
                    "ncclAllReduce")      # ncclFloat32 = 7, ncclSum = 0

    def close(self) -> None:
        if getattr(self, "comm", None) is not None and self.comm.value:
            self.lib().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


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


def replicas_identical(flat: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> bool:
    """True when every rank holds the same bits in ``flat`` (float32): an order-independent 64-bit checksum of the bit
    patterns, compared through one MIN and one MAX all-reduce.  One host sync -- the opt-in replica check of
    FusedTrainer (WIRE_DP_CHECK=k: every k optimizer steps)."""
    cs = flat.detach().contiguous().view(torch.int32).to(torch.int64).sum().reshape(1)
    pair = torch.cat([cs, -cs])
    if pair.is_cuda and dist.get_backend(group) == "gloo":
        host = pair.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        pair = host
    else:
        dist.all_reduce(pair, op=dist.ReduceOp.MAX, group=group)
    return int(pair[0].item()) == -int(pair[1].item())      # max == min


class FlatGradAllReducer:
    """All-reduce(SUM) of a flat gradient buffer, optionally in buckets issued
    on a side stream so that the collective of bucket i overlaps whatever the
    compute stream does next (the next micro-shard's forward/backward).

    The buffer is tiny (1-2 MB), so the collective is latency-bound on xGMI;
    a single bucket is the default.  ``group=None`` with an uninitialised
    process group degrades to a no-op (world size 1).
    """

    # one direct communicator per (device, group): the reducers of several micro-shards / trainers share it
    _direct_comms = {}

    def __init__(self, flat: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                 bucket_floats: Optional[int] = None, use_side_stream: bool = True):
        self.flat = flat
        self.group = group
        self.direct: Optional[RcclDirect] = None
        # WIRE_DP_FORCE=1 keeps the collective path live with a single rank (rehearsal of the RCCL
        # stream plumbing on a one-GPU box)
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
        # RCCL backend: by default torch.distributed.all_reduce (the process group's own stream, two event hand-offs per
        # collective: ~3 % of a step on one rank, tools/dp_overhead.sh).  WIRE_DP_DIRECT=1 issues ncclAllReduce straight
        # from here on the compute stream (RcclDirect: 0 - 0.5 %) -- OPT-IN until a run with two or more ranks has been
        # recorded (no builder session has had a second GPU).  RcclDirect's set-up is symmetric: a failure
        # anywhere raises on every rank, and every rank falls back to the process group together.
        if (self.active and flat.is_cuda and not self.stage_host
                and dist.get_backend(group) == "nccl" and os.environ.get("WIRE_DP_DIRECT", "0") == "1"):
            key = (flat.device.index, id(group))
            comm = FlatGradAllReducer._direct_comms.get(key)
            if comm is None:
                try:
                    comm = RcclDirect(flat.device, group)   # raises on EVERY rank or on none
                except Exception as e:                      # noqa: BLE001 -- any failure means "use the process group"
                    import warnings
                    warnings.warn(f"direct RCCL communicator unavailable ({e}); using torch.distributed.all_reduce")
                    comm = None
                FlatGradAllReducer._direct_comms[key] = comm if comm is not None else False
            self.direct = comm if comm else None

    def launch(self, tensor: Optional[torch.Tensor] = None) -> None:
        """Start the all-reduce of ``tensor`` (default: the flat buffer)."""
        if not self.active:
            return
        t = self.flat if tensor is None else tensor
        if self.direct is not None:
            if self.stream is not None:
                self.stream.wait_stream(torch.cuda.current_stream(t.device))
                with torch.cuda.stream(self.stream):
                    for lo, hi in self.buckets:
                        self.direct.all_reduce_sum_(t[lo:hi])
            else:
                for lo, hi in self.buckets:
                    self.direct.all_reduce_sum_(t[lo:hi])
            return
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

    def launch_range(self, lo: int, hi: int) -> None:
        """Start the all-reduce of flat[lo:hi] behind everything the compute stream holds so far and return at once:
        the per-layer overlap of FusedTrainer (a layer's gradient slice is reduced on the side stream while the compute
        stream goes on with the backward of the layers below it).  ``wait()`` joins the streams again.  Every rank
        calls this with the same ranges in the same order."""
        if not self.active or hi <= lo:
            return
        t = self.flat[lo:hi]
        if self.stage_host:
            host = t.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
            return
        if self.stream is None or not t.is_cuda:
            if self.direct is not None:
                self.direct.all_reduce_sum_(t)
            else:
                self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        self.stream.wait_stream(torch.cuda.current_stream(t.device))
        with torch.cuda.stream(self.stream):
            if self.direct is not None:
                self.direct.all_reduce_sum_(t)
            else:
                self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

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
