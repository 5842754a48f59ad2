from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pca_toy():
    return dict(np.load(GOLDEN / "pca_toy.npz"))


@pytest.fixture(scope="session")
def neighbors_toy():
    return dict(np.load(GOLDEN / "neighbors_toy.npz"))


@pytest.fixture(scope="session")
def pbmc68k():
    from scipy import sparse

    f = dict(np.load(GOLDEN / "pbmc68k_reduced.npz"))
    out = {k: f[k] for k in ("X", "X_pca", "louvain_codes", "bulk_labels_codes", "highly_variable")}
    out["n_neighbors"] = int(f["n_neighbors"][0])
    for name in ("counts", "distances", "connectivities"):
        out[name] = sparse.csr_matrix(
            (f[f"{name}_data"], f[f"{name}_indices"], f[f"{name}_indptr"]), shape=tuple(f[f"{name}_shape"])
        )
    return out
