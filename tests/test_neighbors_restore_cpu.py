"""`Neighbors(adata)` restores a stored graph like the reference (src/scanpy/neighbors/__init__.py:416-474): host
logic only, runs without a GPU."""
from __future__ import annotations

import numpy as np

import scanpy_amd as sc


def _adata(pbmc68k, with_params: bool):
    adata = sc.AnnData(pbmc68k["X"].copy())
    adata.obsp["distances"] = pbmc68k["distances"].copy()
    adata.obsp["connectivities"] = pbmc68k["connectivities"].copy()
    adata.uns["neighbors"] = dict(connectivities_key="connectivities", distances_key="distances")
    if with_params:
        adata.uns["neighbors"]["params"] = dict(n_neighbors=10, method="umap")
    return adata


def test_restore_with_params(pbmc68k):
    nb = sc.Neighbors(_adata(pbmc68k, True))
    assert nb.n_neighbors == 10 and nb.knn is True
    assert nb.distances.shape == (700, 700) and nb.connectivities.nnz == 9992
    assert nb._number_connected_components >= 1


def test_restore_estimates_n_neighbors(pbmc68k):
    """without `params` the reference estimates n_neighbors = nnz(connectivities) / n / 2 (`:446-464`)"""
    nb = sc.Neighbors(_adata(pbmc68k, False))
    assert nb.n_neighbors == int(9992 / 700 / 2)
    adata = _adata(pbmc68k, False)
    del adata.obsp["connectivities"]
    assert sc.Neighbors(adata).n_neighbors == int(pbmc68k["distances"].count_nonzero() / 700)


def test_no_graph():
    nb = sc.Neighbors(sc.AnnData(np.zeros((5, 3), dtype=np.float32)))
    assert nb.distances is None and nb.connectivities is None and nb.n_neighbors is None
