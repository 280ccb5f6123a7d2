"""Data-parallel FusedTrainer on the GPU: two ranks (sharing the one visible card; gloo carries the
flat-gradient all-reduce because RCCL refuses two ranks on one device) must reproduce the
single-process training trajectory on the same global batches -- equal and ragged shards, micro-shards,
index minibatches.  On a multi-GPU node the identical code path runs with backend "nccl" (bench.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import ROOT

pytestmark = pytest.mark.gpu

H, W = 24, 21            # 504 points: odd shard sizes with 2 ranks x 2 micro-shards


def _make(dev, micro, seed=0):
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    torch.manual_seed(seed)
    model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=64,
                           hidden_layers=2, first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0).to(dev)
    g = torch.Generator().manual_seed(3)
    target = torch.rand(H * W, 3, generator=g)
    return model, FusedTrainer(model, (H, W), target, lr=5e-3, niters=100, micro_shards=micro)


HB, WB = 256, 256        # 65 536 points per step: the whole-net kernels incl. the batched weight-gradient launch (>= 65 536 rows)


def _make_big(dev, seed=0):
    """A 256-feature sine net on enough rows for the round-4 path: storing forward, data-gradient chain (its gradients become
    ready in the order L + 1, L .. 1, 0 of the announcements), all hidden weight gradients as one launch."""
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    torch.manual_seed(seed)
    model = models.get_INR(nonlin="siren", in_features=2, out_features=3, hidden_features=256, hidden_layers=3,
                           first_omega_0=30.0, hidden_omega_0=30.0).to(dev)
    g = torch.Generator().manual_seed(3)
    target = torch.rand(HB * WB, 3, generator=g)
    return model, FusedTrainer(model, (HB, WB), target, lr=1e-4, niters=100)


def _run_big(tr, dev):
    losses = [tr.step_hashed(e) for e in range(4)]
    torch.cuda.synchronize()
    return [float(l.item()) for l in losses]


def _run(tr, dev, hashed=False):
    g = torch.Generator().manual_seed(11)
    losses = []
    for it in range(4):
        perm = torch.randperm(H * W, generator=g)
        for b in range(0, H * W, 200):                       # 200 + 200 + 104 (ragged)
            if hashed:      # every rank evaluates only ITS slice of the epoch's position-keyed shuffle
                losses.append(tr.step_hashed(it, first=b, count=min(200, H * W - b)))
            else:
                losses.append(tr.step(perm[b:b + 200].to(dev)))
        tr.scheduler_step()
    torch.cuda.synchronize()
    return [float(l.item()) for l in losses]


def _worker(rank, world, port, micro, out_dir, hashed=False, overlap="layer"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WIRE_DP_OVERLAP"] = overlap
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # hashed mode: the ranks are BUILT from different seeds -- FusedTrainer broadcasts rank 0's parameters
    model, tr = _make(dev, micro, seed=100 + rank if hashed else 0)
    assert tr.world == world and tr.rank == rank
    # one micro-shard: every layer's gradient slice is reduced as soon as wire_train_fwd_bwd_hooked announces it
    # (two ranks on one card exchange through gloo, i.e. host-staged: there the per-layer overlap falls back to one
    #  reduction after the backward -- a blocking host exchange may not run inside the announcement callback, ADVICE r03;
    #  the overlap machinery itself runs on the RCCL communicator in test_direct_rccl_allreduce_... below)
    assert tr.overlap == (micro == 1 and overlap == "layer" and not tr.reducers[0].stage_host)
    losses = _run(tr, dev, hashed)
    flat = tr.flat.detach().cpu().numpy()
    np.save(os.path.join(out_dir, f"flat_{rank}.npy"), flat)
    np.save(os.path.join(out_dir, f"loss_{rank}.npy"), np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("micro,overlap", [(1, "layer"), (1, "none"), (2, "layer")])
def test_two_ranks_match_single_process(tmp_path, micro, overlap):
    port = 29700 + (os.getpid() % 1000) + micro + (5 if overlap == "none" else 0)
    mp.spawn(_worker, args=(2, port, micro, str(tmp_path), False, overlap), nprocs=2, join=True)
    dev = torch.device("cuda", 0)
    model, tr = _make(dev, 1)
    ref_losses = _run(tr, dev)
    ref_flat = tr.flat.detach().cpu().numpy()
    f0, f1 = np.load(tmp_path / "flat_0.npy"), np.load(tmp_path / "flat_1.npy")
    l0 = np.load(tmp_path / "loss_0.npy")
    np.testing.assert_array_equal(f0, f1)                    # replicas stay bit-identical
    np.testing.assert_allclose(l0, ref_losses, rtol=2e-4)    # same trajectory as one process
    # parameters: Adam normalises the step, so compare against the step size
    assert np.abs(f0 - ref_flat).max() < 0.05 * 5e-3


def test_two_ranks_hashed_shuffle_and_broadcast(tmp_path):
    """step_hashed across two ranks == one process on the same positions of the shuffle; replicas built from
    different seeds are made identical by the rank-0 broadcast in FusedTrainer.__init__."""
    port = 29900 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, 1, str(tmp_path), True), nprocs=2, join=True)
    dev = torch.device("cuda", 0)
    model, tr = _make(dev, 1, seed=100)
    ref_losses = _run(tr, dev, hashed=True)
    f0, f1 = np.load(tmp_path / "flat_0.npy"), np.load(tmp_path / "flat_1.npy")
    np.testing.assert_array_equal(f0, f1)
    np.testing.assert_allclose(np.load(tmp_path / "loss_0.npy"), ref_losses, rtol=2e-4)
    assert np.abs(f0 - tr.flat.detach().cpu().numpy()).max() < 0.05 * 5e-3


def _worker_rccl_direct(rank, world, port, out_dir, overlap="layer", big=False):
    """One rank, backend nccl (= RCCL), WIRE_DP_FORCE=1: the collective path stays live, FlatGradAllReducer opens its own
    communicator (parallel.RcclDirect: ncclGetUniqueId -> broadcast of the 128 bytes -> ncclCommInitRank) and issues
    ncclAllReduce on the compute stream."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WIRE_DP_FORCE"] = "1"
    os.environ["WIRE_DP_DIRECT"] = "1"                 # opt-in since round 4 (default: the process group's all_reduce)
    os.environ["WIRE_DP_CHECK"] = "2"                  # the replica check runs too (trivially true on one rank)
    os.environ["WIRE_DP_OVERLAP"] = overlap
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    model, tr = _make_big(dev) if big else _make(dev, 1)
    assert tr.reducers[0].active and tr.reducers[0].direct is not None, "direct RCCL communicator was not set up"
    # "layer": six ncclAllReduce calls per step on the side stream (final layer + loss, hidden layers, first layer), each
    # behind an event of the compute stream; "none": one call on the compute stream after the backward
    assert tr.overlap == (overlap == "layer") and (tr.reducers[0].stream is not None) == (overlap == "layer")
    losses = _run_big(tr, dev) if big else _run(tr, dev)
    np.save(os.path.join(out_dir, "flat_direct.npy"), tr.flat.detach().cpu().numpy())
    np.save(os.path.join(out_dir, "loss_direct.npy"), np.array(losses))
    # the reduced buffer of a 1-rank communicator is the buffer itself: check the call really ran in place
    t = torch.arange(1000, dtype=torch.float32, device=dev)
    tr.reducers[0].direct.all_reduce_sum_(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", ["layer", "none"])
def test_direct_rccl_allreduce_single_rank_matches_plain_run(tmp_path, overlap):
    """ncclAllReduce through librccl.so's C ABI on the compute stream (no ProcessGroup stream
    hand-offs).  RCCL refuses two ranks on one device, so a one-GPU box can only run the 1-rank communicator: the whole
    plumbing (unique id, communicator, stream, in-place reduce) with a sum over one rank = the plain trajectory, bit for
    bit.  wire_occupancy.py:137-158 is the loop being sharded."""
    port = 29500 + (os.getpid() % 400) + (17 if overlap == "layer" else 23)
    mp.spawn(_worker_rccl_direct, args=(1, port, str(tmp_path), overlap), nprocs=1, join=True)
    dev = torch.device("cuda", 0)
    model, tr = _make(dev, 1)
    losses = _run(tr, dev)
    assert np.array_equal(np.load(tmp_path / "loss_direct.npy"), np.array(losses))
    assert np.array_equal(np.load(tmp_path / "flat_direct.npy"), tr.flat.detach().cpu().numpy())


@pytest.mark.parametrize("overlap", ["layer", "none"])
def test_direct_rccl_with_whole_net_kernels_matches_plain_run(tmp_path, overlap):
    """The same on the round-4 path (wire_fused.hip): a 256-feature sine net at 65 536 rows per step runs the storing forward,
    the data-gradient chain and ONE weight-gradient launch for all hidden layers; with WIRE_DP_OVERLAP=layer every layer's
    slice is still announced (wire_grad_ready_fn) after ITS reduction and reduced on the side stream -- same trajectory as
    the plain run, bit for bit.  wire_occupancy.py:137-158 is the loop being sharded."""
    port = 29300 + (os.getpid() % 400) + (31 if overlap == "layer" else 37)
    mp.spawn(_worker_rccl_direct, args=(1, port, str(tmp_path), overlap, True), nprocs=1, join=True)
    dev = torch.device("cuda", 0)
    model, tr = _make_big(dev)
    losses = _run_big(tr, dev)
    assert np.array_equal(np.load(tmp_path / "loss_direct.npy"), np.array(losses))
    assert np.array_equal(np.load(tmp_path / "flat_direct.npy"), tr.flat.detach().cpu().numpy())
