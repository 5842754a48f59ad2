"""Host-side pieces of scanpy_amd.tl.umap that need no GPU, against the oracle restatement of umap-learn."""
from __future__ import annotations

import numpy as np
import pytest

import scanpy_amd as sc
from oracle import umap as ou
from scanpy_amd.tools import _umap


@pytest.mark.parametrize(("spread", "min_dist"), [(1.0, 0.5), (1.0, 0.1), (2.0, 0.3)])
def test_find_ab_params(spread, min_dist):
    a, b = _umap.find_ab_params(spread, min_dist)
    ao, bo = ou.find_ab_params(spread, min_dist)
    assert abs(a - ao) < 1e-9 and abs(b - bo) < 1e-9
    if (spread, min_dist) == (1.0, 0.1):  # umap-learn's documented defaults: a ~ 1.577, b ~ 0.895
        assert abs(a - 1.577) < 5e-3 and abs(b - 0.895) < 5e-3


@pytest.mark.parametrize("n_epochs", [5, 200, 500])
def test_prune_and_schedule(pbmc68k, n_epochs):
    g = pbmc68k["connectivities"]
    csr, eps = _umap._prune_and_schedule(g, n_epochs)
    go = ou.prune_graph(g, n_epochs).tocsr()
    go.sort_indices()
    assert np.array_equal(csr.indptr, go.indptr) and np.array_equal(csr.indices, go.indices)
    np.testing.assert_allclose(eps, ou.make_epochs_per_sample(go.data, n_epochs), rtol=1e-6)
    assert (eps >= 1.0).all()  # at most one firing per epoch: the kernel's schedule relies on it
    assert abs(csr - csr.T).max() == 0  # still symmetric: the mirrored sample has the same schedule


def test_errors_without_gpu(pbmc68k):
    adata = sc.AnnData(pbmc68k["X"].copy())
    with pytest.raises(ValueError, match="Run `sc.pp.neighbors` first"):
        sc.tl.umap(adata)
    adata.obsp["connectivities"] = pbmc68k["connectivities"]
    adata.uns["neighbors"] = dict(connectivities_key="connectivities", distances_key="distances", params=dict(method="umap"))
    with pytest.raises(ValueError, match="Plot PAGA first"):  # (the reference's message, tools/_utils.py:91-93)
        sc.tl.umap(adata, init_pos="paga")
    with pytest.raises(ValueError, match="Unknown method"):
        sc.tl.umap(adata, method="rapids")


def test_init_pos_from_paga_follows_the_reference(pbmc68k):
    """`init_pos='paga'` (src/scanpy/tools/_umap.py:175-179 -> tools/_utils.py:81-114): cells start around their group's
    PAGA node, shifted half-way towards the most strongly connected group's node and jittered along that direction from
    the group's own spawned generator; isolated groups sit exactly on their node.  Checked against a mask-by-mask
    transcription of that rule, with a dense and a sparse coarse matrix."""
    import pandas as pd
    from scipy import sparse

    n = pbmc68k["X"].shape[0]
    rng = np.random.default_rng(3)
    labels = pd.Categorical(rng.integers(0, 5, n).astype(str))
    pos = rng.standard_normal((5, 2)) * 4.0
    coarse = np.array([[0, .3, .9, 0, 0], [.3, 0, .1, 0, 0], [.9, .1, 0, .5, 0], [0, 0, .5, 0, 0], [0, 0, 0, 0, 0]])  # group 4: isolated
    for conn in (coarse, sparse.csr_matrix(coarse)):
        adata = sc.AnnData(pbmc68k["X"].copy())
        adata.obs["grp"] = labels
        adata.uns["paga"] = dict(pos=pos, groups="grp", connectivities=conn)
        got = _umap.init_pos_from_paga(adata, seed=7)
        want = np.ones((n, 2))
        for i, sub in enumerate(np.random.default_rng(7).spawn(5)):
            mask = np.asarray(labels == labels.categories[i])
            nb = np.nonzero(coarse[i])[0]
            if nb.size:
                d = pos[i] - pos[nb[np.argmax(coarse[i][nb])]]
                want[mask] = pos[i] - 0.5 * d + sub.random((int(mask.sum()), 2)) * d
            else:
                want[mask] = pos[i]
        np.testing.assert_array_equal(got, want)
        assert (got[np.asarray(labels == "4")] == pos[4]).all()
    adata.uns["paga"].pop("pos")
    with pytest.raises(ValueError, match="Plot PAGA first"):
        _umap.init_pos_from_paga(adata)


def test_oracle_schemes_agree_in_quality(pbmc68k):
    """the synchronous (Jacobi) scheme the GPU uses optimises the same objective as the reference's sequential sweep"""
    g = pbmc68k["connectivities"]
    a, b = ou.find_ab_params()
    y_seq = ou.simplicial_set_embedding(g, seed=0, scheme="sequential")
    y_syn = ou.simplicial_set_embedding(g, seed=0, scheme="synchronous")
    ce_seq, ce_syn = ou.fuzzy_cross_entropy(g, y_seq, a, b), ou.fuzzy_cross_entropy(g, y_syn, a, b)
    ce_init = ou.fuzzy_cross_entropy(g, ou.initial_embedding(ou.prune_graph(g, 500), 2, "spectral", np.random.RandomState(0)), a, b)
    assert ce_seq < 0.7 * ce_init and ce_syn < 0.7 * ce_init and abs(ce_syn - ce_seq) < 0.06 * ce_seq


def test_spectral_solver_matches_arpack_on_a_graph_without_clusters():
    """`init_pos='spectral'` (umap-learn: ARPACK on the normalised Laplacian): on a connected graph WITHOUT clusters -- a
    curved sheet, eigenvalues 1 - O(1e-4) -- the Chebyshev-filtered subspace iteration returns the plane ARPACK returns
    (round 4's fifty block power steps returned a random one there, and the layout test against the sequential oracle
    showed it).  The operator is applied with float32 operands, as the device SpMM does."""
    import torch
    from scipy import sparse
    from scipy.sparse.linalg import eigsh

    from oracle import connectivities as oc
    from oracle import knn as oknn

    n = 8000
    rng = np.random.default_rng(9)
    uv = rng.random((n, 2))
    feat = np.stack([uv[:, 0], uv[:, 1], np.sin(3 * uv[:, 0]), np.cos(3 * uv[:, 1]), uv[:, 0] * uv[:, 1],
                     np.sin(2 * (uv[:, 0] + uv[:, 1]))], axis=1)
    x = (feat @ rng.standard_normal((6, 50)) + 0.01 * rng.standard_normal((n, 50))).astype(np.float32)
    idx, dist, _ = oknn.knn_sklearn(x, 15, n_jobs=-1)
    g = sparse.csr_matrix(oc.fuzzy_simplicial_set(idx, dist, n, 15)[0]).astype(np.float64)
    deg = np.asarray(g.sum(axis=1)).ravel()
    s = (sparse.diags(1.0 / np.sqrt(deg)) @ g @ sparse.diags(1.0 / np.sqrt(deg))).tocsr()
    s32 = s.astype(np.float32)
    info = {}
    v = _umap._top_eigenvectors_below_trivial(
        lambda y: torch.from_numpy((s32 @ y.numpy().astype(np.float32)).astype(np.float64)), torch.from_numpy(np.sqrt(deg)), 2, 0,
        info=info).numpy()
    lam, vec = eigsh(s, k=3, which="LA", tol=1e-9)
    ref = vec[:, np.argsort(-lam)[1:3]]
    cosines = np.linalg.svd(np.linalg.qr(v)[0].T @ np.linalg.qr(ref)[0], compute_uv=False)
    assert info["converged"] and info["residual"] < 2e-6 and info["operator_applications"] < 1500, info
    assert cosines.min() > 0.9999, (cosines, info)
    np.testing.assert_allclose(info["ritz_values"], np.sort(lam)[::-1][1:3], atol=1e-6)
