from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"

import os  # noqa: E402

if os.environ.get("SCAMD_TESTS_ON_EMULATOR") == "1":
    # the `-m gpu` tests against the host-emulated kernel library (tests/emu/README.md); never set in a normal run
    sys.path.insert(0, str(Path(__file__).resolve().parent / "emu"))
    import patch_torch  # noqa: E402

    patch_torch.activate()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests` on a host without a GPU skips the `gpu` tests instead of failing in their set-up"""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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
    out["obs_n_counts"] = f["obs_n_counts"]
    for name in ("counts", "distances", "connectivities"):
        out[name] = sparse.csr_matrix(
            (f[f"{name}_data"], f[f"{name}_indices"], f[f"{name}_indptr"]), shape=tuple(f[f"{name}_shape"])
        )
    # `.raw.X` of sc.datasets.pbmc68k_reduced() (src/scanpy/datasets/_datasets.py:407-425): counts normalised by the
    # stored size factors, log1p, rounded to 3 digits, one documented tie-break
    size = out["obs_n_counts"].astype(np.float32) / np.float32(1e4)
    raw = out["counts"].astype(np.float32)
    raw.data /= np.repeat(size, np.diff(raw.indptr))
    raw.data = np.round(np.log1p(raw.data), 3)
    raw[357, 715] = 4.019
    out["raw_X"] = raw.tocsr()
    return out


@pytest.fixture(scope="session")
def hvg_golden():
    return dict(np.load(GOLDEN / "hvg_golden.npz"))


@pytest.fixture(scope="session")
def scale_toy():
    return dict(np.load(GOLDEN / "scale_toy.npz"))
