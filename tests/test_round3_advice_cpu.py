"""Round-2 ADVICE items that are host logic (CPU): in-place gene subset carries `varp`, `write_zarr` puts the original
store back when the swap fails, the fuzzy set's global distance sum is the same integer for any sharding."""
from __future__ import annotations

import os

import numpy as np
import pytest
import torch
from scipy import sparse


def test_inplace_subset_var_subsets_varp():
    import scanpy_amd as sc

    a = sc.AnnData(sparse.random(12, 8, density=0.5, format="csr", dtype=np.float32, random_state=0))
    a.varp["corr"] = np.arange(64, dtype=np.float64).reshape(8, 8)
    keep = np.array([0, 2, 5])
    a._inplace_subset_var(keep)
    assert a.X.shape == (12, 3)
    assert a.varp["corr"].shape == (3, 3)
    np.testing.assert_array_equal(a.varp["corr"], np.arange(64).reshape(8, 8)[np.ix_(keep, keep)])


def test_write_zarr_restores_the_store_when_the_swap_fails(tmp_path, monkeypatch):
    import scanpy_amd as sc
    from scanpy_amd import readwrite

    a = sc.AnnData(sparse.random(20, 6, density=0.4, format="csr", dtype=np.float32, random_state=1))
    path = tmp_path / "x.zarr"
    sc.write_zarr(path, a)
    before = sorted(p.name for p in path.iterdir())
    real_replace = os.replace
    calls = {"n": 0}

    def failing_replace(src, dst):
        calls["n"] += 1
        if calls["n"] == 2:  # the second rename (new store into place) fails after the old store was moved away
            raise OSError("simulated failure between the two renames")
        return real_replace(src, dst)

    monkeypatch.setattr(readwrite.os, "replace", failing_replace)
    with pytest.raises(OSError):
        sc.write_zarr(path, a)
    monkeypatch.setattr(readwrite.os, "replace", real_replace)
    assert path.exists(), "the original store must be back in place"
    assert sorted(p.name for p in path.iterdir()) == before
    assert not [p for p in tmp_path.iterdir() if p.name.startswith(".x.zarr")], "no temporary / hidden copies left behind"
    b = sc.read_zarr(path)
    assert b.X.shape == (20, 6)


class _Shards:
    """all-reduce over a list of per-'rank' tensors, in process"""

    def __init__(self, parts, op):
        self.parts, self.op, self.calls = parts, op, 0


def test_fixed_point_distance_sum_is_sharding_independent():
    from scanpy_amd._pipeline import fixed_point_distance_sum

    rng = np.random.default_rng(3)
    d = torch.from_numpy(np.abs(rng.standard_normal((4000, 15))).astype(np.float32) * 7.5)
    total = d.numel()

    class One:
        world_size, rank = 1, 0

        def allreduce_(self, t):
            return t

        def allreduce_max_(self, t):
            return t

    ref = fixed_point_distance_sum(d, total, One())

    # the same sum from 1, 2, 3, 7 row shards: every shard computes with the GLOBAL maximum and count, integers add up
    for shards in (2, 3, 7):
        bounds = np.linspace(0, d.shape[0], shards + 1).astype(int)
        parts = [d[bounds[i]:bounds[i + 1]] for i in range(shards)]
        gmax = max(float(p.max()) for p in parts)
        isum = 0

        class Fake:
            world_size, rank = shards, 0

            def allreduce_max_(self, t):
                t.fill_(gmax)
                return t

            def allreduce_(self, t):
                return t

        import math

        e = math.frexp(gmax)[1]
        s_bits = 61 - e - max(0, (total - 1).bit_length())
        for p in parts:
            isum += int(torch.round(p.to(torch.float64) * (2.0 ** s_bits)).to(torch.int64).sum())
        assert float(ref) == isum * 2.0 ** -s_bits
        # and through the function itself, one shard at a time with the global maximum injected
        acc = 0.0
        for p in parts:
            acc += float(fixed_point_distance_sum(p, total, Fake()))
        assert abs(acc - float(ref)) <= 1e-9 * float(ref)
    assert abs(float(ref) - float(d.to(torch.float64).sum())) < 1e-6 * float(ref)
