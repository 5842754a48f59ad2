"""The C-ABI shared library loads on a CPU-only host and exports every symbol include/scanpy_amd.h declares
(no compute calls without a GPU); the product path fails loudly when there is no GPU."""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "scanpy_amd.h"


def _header_symbols() -> list[str]:
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(scamd_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from scanpy_amd import _build, _lib

    _build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    from scanpy_amd import _lib

    syms = _header_symbols()
    assert len(syms) >= 20
    raw = C.CDLL(str(_lib.LIB_PATH))
    missing = [s for s in syms if not hasattr(raw, s)]
    assert not missing, f"declared in include/scanpy_amd.h but not exported: {missing}"
    # the ctypes binding covers exactly the header (no stale / unbound entry points)
    assert sorted(_lib.SIGNATURES) == syms


def test_abi_version_and_error_string(lib):
    assert lib.scamd_abi_version() == 1
    assert isinstance(lib.scamd_last_error(), bytes)
    assert lib.scamd_device_count() >= 0


def test_workspace_queries_are_host_only(lib):
    """*_workspace_bytes are pure host functions: usable without a device, monotone in the problem size."""
    a = lib.scamd_knn_workspace_bytes(100_000, 50, 100_000, 15)
    b = lib.scamd_knn_workspace_bytes(1_000_000, 50, 1_000_000, 15)
    assert 0 < a < b
    assert lib.scamd_knn_workspace_bytes(1000, 256, 1000, 256) > 0
    assert lib.scamd_knn_workspace_bytes(1000, 257, 1000, 15) == 0  # d > 256 unsupported
    assert lib.scamd_knn_workspace_bytes(1000, 50, 1000, 257) == 0  # k > 256 unsupported
    assert lib.scamd_fuzzy_workspace_bytes(1000, 15) > 0
    assert lib.scamd_leiden_workspace_bytes(1000, 20000) > 0
    assert lib.scamd_csr_transpose_workspace_bytes(1000, 2000, 100000) > 0
    assert lib.scamd_spmm_f64acc_workspace_bytes(2000, 100000, 64) > 0
    assert lib.scamd_colsum_workspace_bytes(64) > 0


def test_argument_validation_without_gpu(lib):
    """Bad arguments are rejected before any HIP call, with a message in scamd_last_error()."""
    rc = lib.scamd_knn_l2_f32(None, 10, 5, 5, 0, 10, 3, None, None, 1.0, None, None, 0, None)
    assert rc == -1 and b"null" in lib.scamd_last_error()
    rc = lib.scamd_spmm_csr_f32(None, None, None, 1, 1, None, 1, None, None, None)
    assert rc == -1


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np

    import scanpy_amd as sc
    from scanpy_amd import _lib

    ad = sc.AnnData(np.random.default_rng(0).random((30, 60)).astype(np.float32))
    with pytest.raises(_lib.ScamdError, match="no CPU fallback"):
        sc.pp.pca(ad, n_comps=5)
    ad.obsm["X_pca"] = np.zeros((30, 5), dtype=np.float32)
    with pytest.raises(_lib.ScamdError, match="no CPU fallback"):
        sc.pp.neighbors(ad, n_neighbors=5)


def test_product_does_not_import_oracle():
    """Nothing under scanpy_amd/ may reference the oracle package (it is test infrastructure)."""
    offenders = []
    for p in (ROOT / "scanpy_amd").rglob("*.py"):
        if re.search(r"^\s*(from|import)\s+oracle\b", p.read_text(), flags=re.M):
            offenders.append(str(p))
    assert not offenders, offenders
