"""`sc.metrics.modularity` (reference tests/test_metrics.py:250-394) through the HIP kernel."""
from __future__ import annotations

from itertools import combinations

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import scanpy_amd

    return scanpy_amd


@pytest.mark.parametrize("is_directed", [False, True])
@pytest.mark.parametrize("use_sparse", [False, True])
def test_modularity_sample_structure(sc, use_sparse, is_directed):
    mat = np.array([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1]])
    adj = sparse.csr_matrix(mat) if use_sparse else mat
    score = sc.metrics.modularity(adj, ["A", "A", "B", "B"], is_directed=is_directed)
    assert 0 <= score <= 1
    assert score == pytest.approx(0.5, rel=1e-9)  # two equal blocks: 1 - 2 * (1/2)^2


def test_modularity_single_community_order_and_empty(sc):
    adj = np.ones((4, 4)) - np.eye(4)
    assert sc.metrics.modularity(adj, ["A"] * 4, is_directed=False) == pytest.approx(0.0, abs=1e-9)
    blocks = np.array([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1]])
    assert sc.metrics.modularity(blocks, ["A", "A", "B", "B"], is_directed=False) == \
        sc.metrics.modularity(blocks, ["B", "B", "A", "A"], is_directed=False)
    assert np.isnan(sc.metrics.modularity(np.zeros((4, 4)), list("ABCD"), is_directed=False))


def test_modularity_errors(sc):
    with pytest.raises(ValueError, match=r"Membership vector size differs"):
        sc.metrics.modularity(np.eye(4), ["A", "A", "B"], is_directed=False)
    with pytest.raises(TypeError, match=r"labels.*array"):
        sc.metrics.modularity(np.eye(3), "col_name", is_directed=False)
    with pytest.raises(TypeError, match=r"is_directed"):
        sc.metrics.modularity(np.eye(3), ["A", "A", "B"], is_directed=None)
    import pandas as pd

    adata = sc.AnnData(np.zeros((3, 2), dtype=np.float32), obs=pd.DataFrame({"label": ["A", "A", "B"]}),
                       obsp=dict(connectivities=sparse.csr_matrix(np.eye(3))), uns=dict(neighbors=dict(params={})))
    with pytest.raises(ValueError, match=r"labels.*string"):
        sc.metrics.modularity(adata, ["A"] * 3, mode="retrieve")
    with pytest.raises(ValueError, match=r"undirected"):
        sc.metrics.modularity(adata, "label", is_directed=True)


def test_modularity_adata_modes_and_oracle(sc, pbmc68k):
    """retrieve == calculate == update (tests/test_metrics.py:311-344) and agreement with the CPU oracle."""
    from oracle import leiden as ol

    adata = sc.AnnData(pbmc68k["X"])
    sc.pp.pca(adata)
    sc.pp.neighbors(adata)
    sc.tl.leiden(adata, flavor="igraph")
    scores = {"retrieve": sc.metrics.modularity(adata, labels="leiden", mode="retrieve")}
    del adata.uns["leiden"]["modularity"]
    scores["calculate"] = sc.metrics.modularity(adata, labels="leiden", mode="calculate")
    assert "modularity" not in adata.uns["leiden"]
    scores["update"] = sc.metrics.modularity(adata, labels="leiden", mode="update")
    assert adata.uns["leiden"]["modularity"] == scores["update"]
    for s in scores.values():
        assert 0 <= s <= 1
    for (_, a), (_, b) in combinations(scores.items(), 2):
        assert a == pytest.approx(b, abs=1e-9)
    ref = ol.modularity(adata.obsp["connectivities"], adata.obs["leiden"].cat.codes.to_numpy())
    assert scores["calculate"] == pytest.approx(ref, abs=1e-7)
