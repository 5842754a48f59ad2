"""CPU pinning of the Leiden oracle (oracle/leiden.c).  PARITY UNPINNED at label level (igraph / leidenalg absent, the
reference ships no golden labels); what CAN be pinned without them:
  * oracle modularity == networkx's `modularity` (an independent implementation of the same formula) and the
    known-answer cases of the reference's tests/test_metrics.py:250-283;
  * the optimiser: planted partitions recovered (ARI), never worse than networkx's Louvain, deterministic per seed,
    seed-sensitive, resolution-monotone (tests/test_clustering.py:67-102 semantics)."""
from __future__ import annotations

import networkx as nx
import numpy as np
import pytest
from scipy import sparse
from sklearn.metrics import adjusted_rand_score

from oracle import leiden as ol


def _nx_graph(adj):
    """Every stored entry (i, j) once as an undirected edge i <= j (the matrix is symmetric)."""
    coo = sparse.triu(sparse.csr_matrix(adj)).tocoo()
    g = nx.Graph()
    g.add_nodes_from(range(adj.shape[0]))
    g.add_weighted_edges_from(zip(coo.row.tolist(), coo.col.tolist(), coo.data.tolist()))
    return g


def _communities(labels):
    return [set(np.flatnonzero(labels == c).tolist()) for c in np.unique(labels)]


@pytest.mark.parametrize("resolution", [0.5, 1.0, 2.0])
def test_modularity_equals_networkx(pbmc68k, resolution):
    adj = pbmc68k["connectivities"]
    g = _nx_graph(adj)
    rng = np.random.default_rng(0)
    for labels in (pbmc68k["louvain_codes"].astype(np.int32), rng.integers(0, 7, adj.shape[0]).astype(np.int32),
                   np.zeros(adj.shape[0], dtype=np.int32), np.arange(adj.shape[0], dtype=np.int32)):
        q = ol.modularity(adj, labels, resolution=resolution)
        assert abs(q - nx.community.modularity(g, _communities(labels), weight="weight", resolution=resolution)) < 1e-12


def test_modularity_known_answers():
    """tests/test_metrics.py:250-283"""
    two_blocks = sparse.csr_matrix(np.array([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1]], dtype=np.float64))
    assert abs(ol.modularity(two_blocks, np.array([0, 0, 1, 1], dtype=np.int32)) - 0.5) < 1e-12
    full = sparse.csr_matrix(np.ones((4, 4)) - np.eye(4))
    assert abs(ol.modularity(full, np.zeros(4, dtype=np.int32))) < 1e-12


def _planted(n_blocks, size, p_in, p_out, seed):
    rng = np.random.default_rng(seed)
    n = n_blocks * size
    truth = np.repeat(np.arange(n_blocks), size)
    same = truth[:, None] == truth[None, :]
    a = np.triu(rng.random((n, n)) < np.where(same, p_in, p_out), 1)
    a = (a + a.T).astype(np.float64)
    return sparse.csr_matrix(a), truth


def test_leiden_recovers_planted_partition_and_beats_louvain():
    adj, truth = _planted(8, 60, 0.3, 0.01, seed=1)
    labels, q = ol.leiden(adj, seed=0)
    assert adjusted_rand_score(truth, labels) > 0.99
    assert abs(q - ol.modularity(adj, labels)) < 1e-12
    g = _nx_graph(adj)
    q_louvain = nx.community.modularity(g, nx.community.louvain_communities(g, weight="weight", seed=0), weight="weight")
    assert q >= q_louvain - 1e-9
    sizes = np.bincount(labels)
    assert (np.diff(sizes) <= 0).all()  # ids by decreasing community size, like leidenalg


def test_leiden_fixture_quality_determinism_and_seed(pbmc68k):
    adj = pbmc68k["connectivities"]
    l0, q0 = ol.leiden(adj, seed=0)
    l1, q1 = ol.leiden(adj, seed=0)
    assert np.array_equal(l0, l1) and q0 == q1  # tests/test_clustering.py:67-83
    l2, _ = ol.leiden(adj, seed=1)
    assert not np.array_equal(l0, l2)  # :86-102
    g = _nx_graph(adj)
    q_louvain = nx.community.modularity(g, nx.community.louvain_communities(g, weight="weight", seed=0), weight="weight")
    assert q0 >= q_louvain - 5e-3
    assert q0 > ol.modularity(adj, pbmc68k["louvain_codes"].astype(np.int32)) - 5e-3  # at least the stored clustering


def test_leiden_resolution_controls_granularity(pbmc68k):
    adj = pbmc68k["connectivities"]
    n_comm = [int(ol.leiden(adj, resolution=r, seed=0)[0].max()) + 1 for r in (0.25, 1.0, 4.0)]
    assert n_comm[0] <= n_comm[1] <= n_comm[2] and n_comm[0] < n_comm[2]
    l2, _ = ol.leiden(adj, n_iterations=2, seed=0)
    assert l2.shape == (700,)
