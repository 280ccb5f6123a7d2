"""The position-keyed shuffle that replaces ``torch.randperm(H*W)`` (wire_image_denoise.py:142,
wire_occupancy.py:137) in the sharded training loops: oracle properties on CPU, device kernel == oracle on GPU."""
import numpy as np
import pytest
import torch

from _util import ROOT  # noqa: F401  (puts the repo root on sys.path)
from oracle import wire_oracle as wo
from wire_amd.parallel import shard_bounds


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 100, 4097, 262144])
def test_hash_perm_is_a_permutation(n):
    for seed in (0, 1, 12345678901234567):
        p = wo.hash_perm(n, seed)
        assert p.dtype == np.int64 and p.shape == (n,)
        assert np.array_equal(np.sort(p), np.arange(n))


def test_hash_perm_epochs_differ_and_look_shuffled():
    n = 65536
    a, b = wo.hash_perm(n, 3), wo.hash_perm(n, 4)
    assert (a != b).mean() > 0.99
    assert abs(np.corrcoef(np.arange(n), a)[0, 1]) < 0.02          # no trend with the position
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.02            # neighbours unrelated
    # every 4096-position minibatch covers the grid about uniformly (16 bins, expected 256 each)
    hist = np.stack([np.bincount(a[s:s + 4096] >> 12, minlength=16) for s in range(0, n, 4096)])
    assert hist.min() > 180 and hist.max() < 340


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_rank_slices_tile_the_global_batch(world):
    """Each rank evaluates only positions [first + lo, first + hi) of the epoch's shuffle (FusedTrainer.step_hashed);
    together the ranks train on exactly the reference's ``indices[b_idx : b_idx + maxpoints]``."""
    n, first, B = 100003, 4000, 20001
    full = wo.hash_perm(n, 9)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        parts.append(wo.hash_perm(n, 9, first + lo, hi - lo))
    assert np.array_equal(np.concatenate(parts), full[first:first + B])


@pytest.mark.gpu
@pytest.mark.parametrize("n,first,count", [(1, 0, 1), (7, 2, 5), (262144, 0, 262144), (262144 * 8, 262144 * 3, 262144),
                                            (512 ** 3, 512 ** 3 - 70000, 70000)])
def test_device_kernel_matches_oracle(n, first, count):
    from wire_amd import _lib
    L = _lib.lib()
    out = torch.empty(count, dtype=torch.int64, device="cuda")
    for seed in (0, 77):
        _lib.check(L.wire_perm_indices(torch.cuda.current_stream().cuda_stream, seed, n, first, count, out.data_ptr()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), wo.hash_perm(n, seed, first, count))
    assert L.wire_perm_indices(None, 0, n, first, count + (n - first - count) + 1, out.data_ptr()) < 0   # past the end


@pytest.mark.gpu
def test_step_hashed_equals_step_on_the_same_indices():
    """FusedTrainer.step_hashed(seed) == FusedTrainer.step(pi_seed as an index tensor): same loss, same gradient,
    bit for bit (one process)."""
    from wire_amd.modules import models
    from wire_amd.trainer import FusedTrainer
    H, W = 48, 40
    res = []
    for mode in ("hashed", "indices"):
        torch.manual_seed(0)
        model = models.get_INR(nonlin="wire", in_features=2, out_features=3, hidden_features=64, hidden_layers=2,
                               first_omega_0=7.0, hidden_omega_0=7.0, scale=6.0).to("cuda")
        g = torch.Generator().manual_seed(2)
        tr = FusedTrainer(model, (H, W), torch.rand(H * W, 3, generator=g), lr=1e-3)
        for e in range(3):
            if mode == "hashed":
                loss = tr.step_hashed(e, first=100, count=1500)
            else:
                idx = torch.tensor(wo.hash_perm(H * W, e, 100, 1500), device="cuda")
                loss = tr.step(idx)
        torch.cuda.synchronize()
        res.append((loss.clone(), tr.flat.clone(), tr.flat_grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        tr.step(torch.arange(0, 100, device="cuda")[::2])          # strided view: refused, not misread
