"""The guarantees the Leiden paper proves for its partitions, as checkers.  Test infrastructure only.

The reference calls `igraph.Graph.community_leiden(objective_function="modularity", weights=..., resolution=...,
n_iterations=-1)` (src/scanpy/tools/_leiden.py:166-196); igraph and leidenalg are absent from this image, so neither the
CPU restatement (oracle/leiden.c) nor the GPU optimiser can be compared with the reference's own output, and a stochastic
optimiser has no golden partition in the reference's tests (tests/test_clustering.py checks parameters, reproducibility
per seed and error paths).  What CAN be pinned is what the published algorithm guarantees of ANY correct implementation
(Traag, Waltman & van Eck, "From Louvain to Leiden: guaranteeing well-connected communities", Sci. Rep. 9, 5233 (2019),
section "Guarantees" and Table 1), stated for the quality function the reference uses -- modularity with resolution g on a
weighted undirected graph, Q = (1 / 2m) sum_C [ e_C - g K_C^2 / 2m ], e_C = twice the internal weight, K_C = total strength:

  * after EVERY iteration: communities are connected (the refinement only merges along edges), and the partition is
    g-separated -- no two communities can be merged with a gain:  E(C, D) - g K_C K_D / 2m <= 0  for all C != D
    (Louvain and Leiden both; it follows from the aggregate-level local moving having converged);
  * after a STABLE iteration (n_iterations = -1 runs until an iteration changes nothing): node optimality -- no single
    vertex can be moved to another (or a new, empty) community with a gain:
        [ k_v(D) - g k_v K_D / 2m ]  -  [ k_v(C \\ v) - g k_v (K_C - k_v) / 2m ]  <= 0   for all v in C, all D.

`connected` is checked by bench.py's full-size properties and the GPU tests already; this module adds the other two.
Gains are returned in units of Q (divided by m), so that a tolerance means the same on every graph.
"""
from __future__ import annotations

import numpy as np
from scipy import sparse


def _prep(adj, labels, objective: str = "modularity", node_weights=None):
    """-> (a, lab, nu, norm, onehot, ntot, two_m): vertex weights nu and the divisor of the penalty term -- strengths and 2m for
    modularity, ones (or igraph's `node_weights`) and 1 for CPM (igraph `objective_function='CPM'`: gain of joining D =
    k_v(D) - g n_v N_D)"""
    a = sparse.csr_matrix(adj).astype(np.float64)
    a.setdiag(0)  # (the reference's graphs have no self loops; the optimisers ignore them)
    a.eliminate_zeros()
    labels = np.asarray(labels)
    _, lab = np.unique(labels, return_inverse=True)
    n, nc = a.shape[0], int(lab.max()) + 1
    k = np.asarray(a.sum(axis=1)).ravel()
    two_m = float(k.sum())
    if objective.lower() == "cpm":
        nu, norm = (np.ones(n) if node_weights is None else np.asarray(node_weights, dtype=np.float64)), 1.0
    elif objective.lower() == "modularity":
        nu, norm = k, two_m
    else:
        raise ValueError(f"objective {objective!r}")
    onehot = sparse.csr_matrix((np.ones(n), (np.arange(n), lab)), shape=(n, nc))
    ntot = np.asarray(onehot.T @ nu).ravel()
    return a, lab, nu, norm, onehot, ntot, two_m


def quality(adj, labels, *, resolution: float = 1.0, objective: str = "modularity", node_weights=None) -> float:
    """(1 / 2m) sum_C [ e_C - g N_C^2 / norm ]: the modularity, or igraph's CPM quality, of a partition"""
    a, lab, nu, norm, onehot, ntot, two_m = _prep(adj, labels, objective, node_weights)
    e = np.asarray((onehot.T @ a @ onehot).diagonal()).ravel()
    return float((e - resolution * ntot * ntot / norm).sum() / two_m)


def improving_moves(adj, labels, *, resolution: float = 1.0, tol: float = 1e-12, objective: str = "modularity", node_weights=None):
    """vertices that a single move to a neighbouring community (or to a community of their own) would improve.

    -> dict(count, fraction, max_gain (in units of Q), worst_vertex)"""
    a, lab, k, norm, onehot, ktot, two_m = _prep(adj, labels, objective, node_weights)
    n = a.shape[0]
    g = resolution
    w = (a @ onehot).tocsr()  # w[v, D] = k_v(D), stored for the communities v has an edge to
    w.sort_indices()
    rows = np.repeat(np.arange(n), np.diff(w.indptr))
    cols = w.indices
    own = cols == lab[rows]
    k_own = np.zeros(n)
    k_own[rows[own]] = w.data[own]  # k_v(C \ v): no self loops
    stay = k_own - g * k * (ktot[lab] - k) / norm
    gain = w.data - g * k[rows] * ktot[cols] / norm - stay[rows]
    gain[own] = -np.inf
    best = np.full(n, -np.inf)
    np.maximum.at(best, rows, gain)
    best = np.maximum(best, 0.0 - stay)  # a community of its own: k_v(empty) = 0, K = 0
    best /= two_m / 2.0
    bad = best > tol
    worst = int(np.argmax(best)) if n else -1
    return {"count": int(bad.sum()), "fraction": float(bad.mean()) if n else 0.0, "max_gain": float(best.max()) if n else 0.0,
            "worst_vertex": worst}


def mergeable_pairs(adj, labels, *, resolution: float = 1.0, tol: float = 1e-12, objective: str = "modularity", node_weights=None):
    """pairs of communities whose merge would improve the quality (g-separation violated).

    -> dict(count, max_gain (in units of Q), n_communities)"""
    a, lab, k, norm, onehot, ktot, two_m = _prep(adj, labels, objective, node_weights)
    e = (onehot.T @ a @ onehot).tocoo()  # E(C, D) for C != D (each unordered pair twice)
    off = e.row < e.col
    gain = (e.data[off] - resolution * ktot[e.row[off]] * ktot[e.col[off]] / norm) / (two_m / 2.0)
    return {"count": int((gain > tol).sum()), "max_gain": float(gain.max()) if gain.size else 0.0,
            "n_communities": int(ktot.size)}
