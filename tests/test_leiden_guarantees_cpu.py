"""The CPU Leiden oracle against what the Leiden paper guarantees of any correct implementation (CPU only).

igraph / leidenalg are absent and the reference's tests hold no golden partition (a stochastic optimiser), so the oracle
(oracle/leiden.c) cannot be compared with the reference's output.  It CAN be held to the properties the published
algorithm proves (oracle/leiden_guarantees.py: Traag, Waltman & van Eck 2019, "Guarantees"): after a stable iteration
(n_iterations = -1) no single vertex move and no merge of two communities improves the quality, and every community is
connected.  The checkers themselves are tested by what they must reject: an unfinished run and a perturbed partition.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
from scipy import sparse
from scipy.sparse.csgraph import connected_components

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import connectivities as oconn  # noqa: E402
from oracle import knn as oknn  # noqa: E402
from oracle import leiden as ol  # noqa: E402
from oracle import leiden_guarantees as lg  # noqa: E402


def _graph(n, n_c, spread, seed):
    r = np.random.default_rng(seed)
    c = r.normal(size=(n_c, 10)) * spread
    x = (c[r.integers(0, n_c, n)] + r.normal(size=(n, 10))).astype(np.float32)
    idx, dist = oknn.knn_exact_f64(x, np.arange(n), 15)
    conn, _, _ = oconn.fuzzy_simplicial_set(idx, dist, n, 15)
    return conn


def _communities_connected(conn, labels) -> bool:
    conn = sparse.csr_matrix(conn)
    rows = np.repeat(np.arange(conn.shape[0]), np.diff(conn.indptr))
    same = labels[rows] == labels[conn.indices]
    inner = sparse.csr_matrix((same.astype(np.int8), conn.indices.copy(), conn.indptr.copy()), shape=conn.shape)  # (eliminate_zeros works in place)
    inner.eliminate_zeros()
    return connected_components(inner, directed=False)[0] == int(labels.max()) + 1


GRAPHS = {"separated": (3000, 12, 4.0, 1), "overlapping": (4000, 30, 1.5, 2), "structure-less": (3000, 1, 1.0, 3)}


@pytest.fixture(scope="module", params=list(GRAPHS))
def graph(request):
    return request.param, _graph(*GRAPHS[request.param])


@pytest.mark.parametrize("resolution", [1.0, 0.5, 2.0])
def test_oracle_stable_partition_is_node_optimal_separated_and_connected(graph, resolution):
    name, conn = graph
    for seed in (0, 1):
        memb, q = ol.leiden(conn, resolution=resolution, n_iterations=-1, seed=seed)
        im = lg.improving_moves(conn, memb, resolution=resolution)
        mp = lg.mergeable_pairs(conn, memb, resolution=resolution)
        assert im["count"] == 0, (name, resolution, seed, im)
        assert mp["count"] == 0, (name, resolution, seed, mp)
        assert _communities_connected(conn, memb), (name, resolution, seed)
        assert abs(q - ol.modularity(conn, memb, resolution=resolution)) < 1e-9


def test_checkers_reject_what_they_must():
    """an unfinished run on an ambiguous graph is not node optimal (the paper guarantees that for stable iterations only);
    a partition with 1 % of its vertices moved at random has improving moves; two halves of one community are mergeable"""
    conn = _graph(*GRAPHS["structure-less"])
    memb2, _ = ol.leiden(conn, n_iterations=1, seed=0)
    assert lg.improving_moves(conn, memb2)["count"] > 0
    assert lg.mergeable_pairs(conn, memb2)["count"] == 0  # (separation holds after every iteration)
    conn_s = _graph(*GRAPHS["separated"])
    memb, _ = ol.leiden(conn_s, n_iterations=-1, seed=0)
    r = np.random.default_rng(0)
    moved = memb.copy()
    pick = r.choice(memb.size, memb.size // 100, replace=False)
    moved[pick] = (moved[pick] + 1 + r.integers(0, memb.max(), pick.size)) % (memb.max() + 1)
    out = lg.improving_moves(conn_s, moved)
    assert out["count"] >= pick.size // 2 and out["max_gain"] > 0
    split = memb.copy()
    big = np.flatnonzero(memb == 0)
    split[big[: big.size // 2]] = memb.max() + 1
    assert lg.mergeable_pairs(conn_s, split)["count"] >= 1


def test_gains_are_differences_of_modularity():
    """the checker's gain of a move / a merge equals the difference of the oracle's modularity, to rounding"""
    conn = _graph(*GRAPHS["overlapping"])
    memb, _ = ol.leiden(conn, n_iterations=1, seed=0)
    im = lg.improving_moves(conn, memb)
    if im["count"]:
        v = im["worst_vertex"]
        # the best single move of v, found by trying every neighbouring community and a fresh one
        q0 = ol.modularity(conn, memb)
        best = -np.inf
        row = sparse.csr_matrix(conn)[v].indices
        for c in set(memb[row].tolist()) | {int(memb.max()) + 1}:
            if c == memb[v]:
                continue
            trial = memb.copy()
            trial[v] = c
            best = max(best, ol.modularity(conn, trial) - q0)
        assert abs(best - im["max_gain"]) < 1e-9
    split = memb.copy()
    big = np.flatnonzero(memb == 0)
    split[big[: big.size // 2]] = memb.max() + 1
    mp = lg.mergeable_pairs(conn, split)
    gain = ol.modularity(conn, memb) - ol.modularity(conn, split)
    assert mp["max_gain"] >= gain - 1e-9  # (the undone split is one of the candidate merges)
