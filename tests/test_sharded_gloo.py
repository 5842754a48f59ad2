"""world_size-2 (gloo, CPU) runs of the row-sharded path against the single-process run.

Covers SURVEY.md 8(e): contiguous balanced row blocks, PCA with float64 all-reduces of g x b panels, all-gather
of the embedding and of the kNN lists, graph + Leiden on rank 0, label broadcast."""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

import numpy as np

from scanpy_amd._pipeline import shard_bounds

HERE = Path(__file__).resolve().parent


def _launch(world: int, tmp: Path, n: int, g: int, k: int, mode: str):
    init = tmp / f"init_{mode}_{world}"
    procs = [subprocess.Popen([sys.executable, str(HERE / "dist_worker.py"), str(r), str(world), str(init), str(tmp),
                               str(n), str(g), str(k), mode]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [dict(np.load(tmp / f"rank{r}_of{world}.npz")) for r in range(world)]


def test_shard_bounds_partition():
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_pca_two_ranks_matches_one(tmp_path):
    n, g, k = 1501, 300, 12  # odd n: the two shards differ in length
    one = _launch(1, tmp_path, n, g, k, "pca")[0]
    two = _launch(2, tmp_path, n, g, k, "pca")
    scores = np.concatenate([r["scores"] for r in two], axis=0)
    assert scores.shape == one["scores"].shape
    for r in two:  # every rank holds the same global model
        np.testing.assert_allclose(r["components"], two[0]["components"], rtol=0, atol=0)
        np.testing.assert_allclose(r["variance"], one["variance"], rtol=1e-9)
        np.testing.assert_allclose(r["ratio"], one["ratio"], rtol=1e-9)
        np.testing.assert_allclose(r["mean"], one["mean"], rtol=1e-12, atol=1e-15)
    # the all-reduce changes the float64 summation order only: loadings agree far below the 1e-4 parity bar
    assert np.abs(two[0]["components"] - one["components"]).max() < 1e-8
    assert np.abs(scores - one["scores"]).max() < 1e-5


def test_full_path_two_ranks_matches_one(tmp_path):
    from sklearn.metrics import adjusted_rand_score

    n, g, k = 1201, 240, 10
    one = _launch(1, tmp_path, n, g, k, "path")[0]
    two = _launch(2, tmp_path, n, g, k, "path")
    assert bool(two[0]["has_graph"]) and not bool(two[1]["has_graph"])  # graph + Leiden live on rank 0 only
    idx = np.concatenate([r["knn_idx"] for r in two], axis=0)
    dist = np.concatenate([r["knn_dist"] for r in two], axis=0)
    assert idx.shape == one["knn_idx"].shape
    # each rank answered exactly its own row block, against ALL candidates
    assert (idx[:, 0] == np.arange(n)).all()
    same = (np.sort(idx, axis=1) == np.sort(one["knn_idx"], axis=1)).all(axis=1).mean()
    assert same > 0.999
    np.testing.assert_allclose(dist, one["knn_dist"], rtol=1e-4, atol=1e-5)
    # labels: identical on both ranks (broadcast), same clustering as the single-process run
    np.testing.assert_array_equal(two[0]["labels"], two[1]["labels"])
    assert adjusted_rand_score(two[0]["labels"], one["labels"]) > 0.99
    assert int(two[1]["nc"]) == int(two[0]["nc"]) and abs(float(two[1]["q"]) - float(two[0]["q"])) < 1e-12
    # the graph rank 0 assembled from the two ranks' rows (membership strengths per shard, all-to-all of the directed
    # edges, local merge) is the single-process graph: same pattern, same values
    from oracle import connectivities as oc

    ref, _, _ = oc.fuzzy_simplicial_set(idx, dist.astype(np.float32), n, k)  # from the two ranks' own kNN lists
    ref.sort_indices()
    np.testing.assert_array_equal(two[0]["conn_indptr"], ref.indptr)
    np.testing.assert_array_equal(two[0]["conn_indices"], ref.indices)
    np.testing.assert_allclose(two[0]["conn_data"], ref.data, rtol=0, atol=1e-6)


def test_full_path_two_ranks_streaming_their_blocks_from_a_zarr_store(tmp_path, monkeypatch):
    """out of core + sharded (SURVEY.md 8(e) + 8(f).4): each rank opens the store `backed='r'` and streams only its own
    row block through the chunked PCA; the result equals the in-memory two-rank run bit for bit"""
    import scanpy_amd as sc
    from scanpy_amd import readwrite as rw
    from scanpy_amd.datasets import synthetic_planted

    n, g, k = 1201, 240, 10
    monkeypatch.setattr(rw, "CHUNK_ELEMS", 4001)
    monkeypatch.setattr(rw, "CHUNKS_PER_SHARD", 3)
    x, _ = synthetic_planted(n, g, n_types=12, seed=5)
    sc.write_zarr(tmp_path / "store.zarr", sc.AnnData(x))
    mem = _launch(2, tmp_path, n, g, k, "path")
    disk = _launch(2, tmp_path, n, g, k, "path_backed")
    for a, b in zip(mem, disk):
        for key in ("scores", "knn_idx", "knn_dist", "labels", "q", "nc"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=key)
