"""world_size-2 (gloo, CPU) test of the data-parallel exchange: each rank
computes the gradient of ITS shard's loss (scaled by n_g/B) with the oracle,
FlatGradAllReducer sums the flat buffers, and the result must equal the
full-batch gradient -- including an odd batch (unequal shards) and bucketed
launches.  The compute here is the oracle; the exchange logic under test is the
product's (wire_amd/parallel.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import ROOT, load_golden, meta


def _flat(grads, order):
    from oracle import wire_oracle as wo
    return np.concatenate([wo.as_real_pairs(grads[k]).astype(np.float64).ravel() for k in order])


def _worker(rank, world, port, B, bucket, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import wire_oracle as wo
    from wire_amd.parallel import FlatGradAllReducer, shard_bounds, shard_weight
    rec = load_golden("small_wire_d2")
    m = meta(rec)
    P = wo.cast_params({k[2:]: v for k, v in rec.items() if k.startswith("p:") and "omega" not in k
                        and "scale_0" not in k}, True)
    order = list(P.keys())
    coords = rec["coords"][0, :B].astype(np.float64)
    target = rec["target"][0, :B].astype(np.float64)
    lo, hi = shard_bounds(B, world, rank)
    w = shard_weight(B, world, rank)
    y, cache = wo.wire_forward(P, coords[lo:hi], m["L"], m["om1"], m["om"], m["sc"], keep=True)
    loss, gy = wo.mse_loss_and_grad(y, target[lo:hi])
    g = wo.wire_backward(P, cache, gy * w, m["L"], m["om1"], m["om"], m["sc"])
    flat = torch.tensor(np.concatenate([_flat(g, order), [loss * w]]))
    red = FlatGradAllReducer(flat, bucket_floats=None if bucket == "ranges" else bucket)
    if bucket == "ranges":
        # the per-layer overlap of FusedTrainer: slices announced from the last tensor to the first (launch_range), one join
        assert red.active
        sizes = [wo.as_real_pairs(g[k]).size for k in order]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        nt = len(order)
        red.launch_range(int(offs[nt - 2]), flat.numel())            # final layer + the loss behind it
        for t in range(nt - 4, 0, -2):
            red.launch_range(int(offs[t]), int(offs[t + 2]))
        red.launch_range(0, int(offs[2]))
        red.wait()
    else:
        assert red.active and len(red.buckets) == (1 if not bucket else -(-flat.numel() // bucket))
        red()
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B,bucket", [(256, None), (257, 1000), (255, "ranges")])
def test_sharded_gradient_equals_full_batch(tmp_path, B, bucket):
    from oracle import wire_oracle as wo
    port = 29500 + (os.getpid() % 2000) + (0 if not bucket else (1 if bucket == 1000 else 2))
    mp.spawn(_worker, args=(2, port, B, bucket, str(tmp_path)), nprocs=2, join=True)
    red = np.load(tmp_path / "reduced.npy")
    rec = load_golden("small_wire_d2")
    m = meta(rec)
    P = wo.cast_params({k[2:]: v for k, v in rec.items() if k.startswith("p:") and "omega" not in k
                        and "scale_0" not in k}, True)
    y, cache = wo.wire_forward(P, rec["coords"][0, :B].astype(np.float64), m["L"], m["om1"], m["om"], m["sc"], keep=True)
    loss, gy = wo.mse_loss_and_grad(y, rec["target"][0, :B].astype(np.float64))
    g = wo.wire_backward(P, cache, gy, m["L"], m["om1"], m["om"], m["sc"])
    full = np.concatenate([_flat(g, list(P.keys())), [loss]])
    assert np.abs(red - full).max() <= 1e-12 * np.abs(full).max()


def _check_worker(rank, world, port, diverge):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wire_amd.parallel import replicas_identical
    flat = torch.linspace(-1, 1, 1001, dtype=torch.float32)
    if diverge and rank == 1:
        flat[517] = torch.nextafter(flat[517], torch.tensor(2.0))      # one bit in one parameter of one replica
    same = replicas_identical(flat)
    assert same == (not diverge), (rank, same)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("diverge", [False, True])
def test_replica_check_sees_a_single_bit(diverge):
    """WIRE_DP_CHECK (FusedTrainer): every rank learns whether all replicas hold the same parameter bits -- an
    order-independent checksum through one MAX all-reduce; a one-ulp difference in one element of one rank shows."""
    port = 31500 + (os.getpid() % 2000) + int(diverge)
    mp.spawn(_check_worker, args=(2, port, diverge), nprocs=2, join=True)
