"""bench.py's `full_size_properties` on a result produced by the CPU chain (sklearn PCA / kNN, oracle fuzzy set and
Leiden): every gate green; then one corruption per gate, which must be named.  No GPU: the function only reads host
arrays."""
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
from scipy import sparse
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from oracle import connectivities as oconn  # noqa: E402
from oracle import knn as oknn  # noqa: E402
from oracle import leiden as ol  # noqa: E402
from oracle import pca as opca  # noqa: E402


@pytest.fixture(scope="module")
def chain():
    n, g, k = 3000, 300, 15
    x, _ = bench.make_matrix(n, g, 0, "planted")
    ref = opca.pca_reference(x, 20)
    x_pca, comps, var = ref["X_pca"], ref["components"], ref["variance"]
    idx, dist, _ = oknn.knn_sklearn(np.asarray(x_pca, dtype=np.float32), k)
    # the product returns float64 distances of the float32 embedding: give the checker the same
    ei, ed = oknn.knn_exact_f64(np.asarray(x_pca, dtype=np.float32), np.arange(n), k)
    conn, _, _ = oconn.fuzzy_simplicial_set(ei, ed, n, k)
    labels, q = ol.leiden(conn, seed=0)
    res = SimpleNamespace(x_pca=np.asarray(x_pca, dtype=np.float32), components=np.asarray(comps), variance=np.asarray(var),
                          knn_indices=ei.astype(np.int32), knn_distances=ed, conn_indptr=conn.indptr.astype(np.int64),
                          conn_indices=conn.indices.astype(np.int32), conn_data=conn.data.astype(np.float32),
                          labels=labels.astype(np.int32), modularity=q, n_communities=int(labels.max()) + 1)
    return res, x, n, k


def _run(res, x, n, k):
    return bench.full_size_properties(res, x, n, k, n_sample=256)


def test_clean_result_passes_every_gate(chain):
    res, x, n, k = chain
    out = _run(res, x, n, k)
    assert out["failed_gates"] == [], out
    assert out["knn"]["rows_differing_beyond_ties"] == 0 and out["knn"]["self_first_rows"] == 1.0
    assert out["connectivities"]["sample_max_abs"] <= 1e-6 and out["connectivities"]["sample_entries"] > 1000
    assert out["leiden"]["disconnected_communities"] == 0 and out["leiden"]["modularity_abs_err"] < 1e-9
    assert out["leiden"]["improving_moves"] == 0 and out["leiden"]["mergeable_pairs"] == 0  # (the oracle's stable partition)
    assert out["pca"]["scores_sample_rel_err"] < 1e-4


def test_approximate_run_reports_recall_instead_of_gating_exactness(chain):
    """`bench.py --knn-nprobe`: lists that are exact among SOME rows only -- here every row's farthest listed neighbour
    replaced by a far-away cell with its true distance -- pass the kNN gates, and the sampled recall says what was lost"""
    res, x, n, k = chain
    idx, dist = res.knn_indices.copy(), res.knn_distances.copy()
    far = np.array([next(c for c in ((i + n // 2 + t) % n for t in range(n)) if c not in idx[i, :-1]) for i in range(n)], dtype=np.int32)
    idx[:, -1] = far
    dist[:, -1] = np.sqrt(((res.x_pca.astype(np.float64) - res.x_pca[far].astype(np.float64)) ** 2).sum(1))
    dist[:, -1] = np.maximum(dist[:, -1], dist[:, -2])  # (keeps the rows ascending whatever cell was picked)
    out = bench.full_size_properties(_copy(res, knn_indices=idx, knn_distances=dist), x, n, k, n_sample=256, approximate=True)
    assert "knn_rows_differing_beyond_ties" not in out["failed_gates"], out["knn"]
    assert out["knn"]["approximate"] and abs(out["knn"]["sample_recall"] - (k - 2) / (k - 1)) < 0.03
    exact = bench.full_size_properties(res, x, n, k, n_sample=256, approximate=True)
    assert exact["knn"]["sample_recall"] == 1.0 and exact["failed_gates"] == []


def _copy(res, **kw):
    d = dict(vars(res))
    d.update(kw)
    return SimpleNamespace(**d)


def test_each_corruption_is_named(chain):
    res, x, n, k = chain
    rows = np.sort(np.random.default_rng(123).choice(n, size=256, replace=False))  # the function's own sample
    # a wrong neighbour in a sampled row
    idx = res.knn_indices.copy()
    i = int(rows[3])
    far = int(np.setdiff1d(np.arange(n), idx[i])[-1])
    idx[i, -1] = far
    assert "knn_max_rel_distance_err" in _run(_copy(res, knn_indices=idx), x, n, k)["failed_gates"]  # (distance not the pair's)
    dist = res.knn_distances.copy()
    dist[i, -1] = np.sqrt(((res.x_pca[i].astype(np.float64) - res.x_pca[far].astype(np.float64)) ** 2).sum())
    assert "knn_rows_differing_beyond_ties" in _run(_copy(res, knn_indices=idx, knn_distances=dist), x, n, k)["failed_gates"]
    # one connectivity value off by 1e-3 (also breaks symmetry)
    data = res.conn_data.copy()
    data[res.conn_indptr[int(rows[0])]] += np.float32(1e-3)
    got = _run(_copy(res, conn_data=data), x, n, k)["failed_gates"]
    assert "conn_asymmetry" in got and "conn_sample_max_abs" in got
    # two communities merged into one label: connected or not, the reported modularity no longer matches
    lab = res.labels.copy()
    lab[lab == lab.max()] = 0
    got = _run(_copy(res, labels=lab, n_communities=int(lab.max()) + 1), x, n, k)["failed_gates"]
    assert "modularity_abs_err" in got
    # a cell moved into a community it has no edge to
    lab = res.labels.copy()
    conn_rows = np.split(res.conn_indices, res.conn_indptr[1:-1])
    for v in range(n):
        other = np.setdiff1d(np.unique(res.labels), np.unique(res.labels[conn_rows[v]]))
        if other.size:
            lab[v] = other[0]
            break
    got = _run(_copy(res, labels=lab), x, n, k)["failed_gates"]
    assert "disconnected_communities" in got
    # a community cut in two halves (relabelled consistently, modularity recomputed): the halves are mergeable with a gain
    lab = res.labels.copy()
    big = np.flatnonzero(lab == 0)
    lab[big[: big.size // 2]] = lab.max() + 1
    got = _run(_copy(res, labels=lab, n_communities=int(lab.max()) + 1, modularity=ol.modularity(
        sparse.csr_matrix((res.conn_data, res.conn_indices, res.conn_indptr), shape=(n, n)), lab)), x, n, k)["failed_gates"]
    assert "leiden_mergeable_pairs" in got and "leiden_improving_moves" in got
    # loadings scaled: not orthonormal, scores no longer theirs
    got = _run(_copy(res, components=res.components * 1.01), x, n, k)["failed_gates"]
    assert "pca_orthonormality_err" in got and "pca_scores_sample_rel_err" in got
