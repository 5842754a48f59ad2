"""-m gpu: the row-sharded path with 2 and 3 ranks sharing cuda:0 (real kernels, gloo collectives through host memory)
against the single-rank run -- SURVEY.md 8(e) on hardware, minus RCCL itself (see tests/dist_gpu_worker.py)."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def _launch(world: int, tmp: Path, n: int, g: int, k: int):
    init = tmp / f"gpu_init_{world}"
    procs = [subprocess.Popen([sys.executable, str(HERE / "dist_gpu_worker.py"), str(r), str(world), str(init), str(tmp),
                               str(n), str(g), str(k)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [dict(np.load(tmp / f"gpu_rank{r}_of{world}.npz")) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_match_single_rank(tmp_path, world):
    n, g, k = 70001, 400, 30  # n >= 65536: the cell-pruned kNN sweep, query-sharded; odd n: ragged shards
    if os.environ.get("SCAMD_TESTS_ON_EMULATOR") == "1":  # (host-emulated kernels, tests/emu: a size they finish;
        n, g = 6001, 200                                   #  run with SCAMD_KNN_IVF=1 to keep the pruned sweep)
    one = _launch(1, tmp_path, n, g, k)[0]
    many = _launch(world, tmp_path, n, g, k)
    # PCA: the int64 Gram matrix is additive over shards -> the SAME model on every rank and for every world size
    for r in many:
        np.testing.assert_array_equal(r["components"], one["components"])
        np.testing.assert_array_equal(r["variance"], one["variance"])
    scores = np.concatenate([r["scores"] for r in many], axis=0)
    np.testing.assert_array_equal(scores, one["scores"])
    # kNN: every rank answers its own rows against all candidates -> identical lists
    idx = np.concatenate([r["knn_idx"] for r in many], axis=0)
    dist = np.concatenate([r["knn_dist"] for r in many], axis=0)
    np.testing.assert_array_equal(idx, one["knn_idx"])
    np.testing.assert_array_equal(dist, one["knn_dist"])
    # graph + Leiden on rank 0 only, labels broadcast: identical to the single-rank labels
    assert bool(many[0]["has_graph"]) and not any(bool(r["has_graph"]) for r in many[1:])
    # the graph itself: per-rank membership strengths + all-to-all of the directed edges + per-rank merge give the rows of
    # the single-device fuzzy set bit for bit (same float32 expression on the same operands)
    for key in ("conn_indptr", "conn_indices", "conn_data"):
        np.testing.assert_array_equal(many[0][key], one[key], err_msg=key)
    for r in many:
        np.testing.assert_array_equal(r["labels"], one["labels"])
        assert int(r["nc"]) == int(one["nc"]) and float(r["q"]) == float(one["q"])
