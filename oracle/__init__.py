"""CPU oracle for the `pca -> neighbors -> leiden` path.  TEST INFRASTRUCTURE ONLY.

Nothing in `scanpy_amd/` may import this package: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and only as the
checker.  Each function cites the reference file:line (relative to /root/reference) whose
arithmetic it restates.

Pinning status (see DESIGN.md "Oracle"):
  * pca             -- PINNED: literally the reference's own sklearn call
                       (src/scanpy/preprocessing/_pca/__init__.py:287-308); checked against
                       tests/test_pca.py:34-59 golden vectors (tests/golden/pca_toy.npz).
  * knn (exact)     -- PINNED: the reference's `transformer='sklearn'` call
                       (src/scanpy/neighbors/__init__.py:754-768) + `_common.py` conventions;
                       checked against tests/test_neighbors.py:27-39.
  * connectivities  -- PINNED: restatement of umap-learn 0.5.x fuzzy_simplicial_set (not in the
                       container) anchored on tests/test_neighbors.py:43-48 and on the bundled
                       pbmc68k_reduced fixture (stored distances -> stored connectivities).
  * leiden_guarantees -- checkers for what the Leiden paper proves of a stable partition (node optimality,
                       g-separation); the oracle meets both exactly (tests/test_leiden_guarantees_cpu.py).
  * leiden          -- PARITY UNPINNED at label level: igraph/leidenalg are absent and the
                       reference ships no golden labels (tests/test_clustering.py only pins
                       determinism, NMI>0.9 across flavors, and modularity).  `oracle/leiden.c`
                       restates Traag et al. 2019 with the call contract of
                       src/scanpy/tools/_leiden.py:166-196; modularity is cross-checked against
                       networkx and planted partitions.  The one set of reference-PRODUCED labels in the
                       tree anchors it (tests/test_gpu_parity_hard.py): the `louvain` column of the bundled
                       fixture (src/scanpy/datasets/_datasets.py:349-427) -- NMI of the oracle's partition
                       of the fixture's own graph against it 0.90 (0.93 at matched cluster count), above
                       the reference's cross-implementation bar (tests/test_clustering.py:130-163).
  * compare.py      -- the north-star gates (loadings 1e-4, kNN sets equal beyond ties, connectivities 1e-5,
                       ARI 0.99) as functions, used by tests/ and by bench.py's `parity` leg.
"""
