"""ctypes wrapper of oracle/leiden.c (CPU Leiden restatement).  Test infrastructure only.

Call contract follows src/scanpy/tools/_leiden.py:166-196: symmetric adjacency (both directions
stored), float64 weights, resolution, n_iterations (<0: until stable), integer seed, beta = 0.01.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
from scipy import sparse

HERE = Path(__file__).resolve().parent
LIB = HERE / "_lib" / "liboracle_leiden.so"


def build() -> Path:
    if not LIB.exists() or LIB.stat().st_mtime < (HERE / "leiden.c").stat().st_mtime:
        subprocess.run(["make", "-s", "-C", str(HERE)], check=True)
    return LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(str(build()))
        lib.oracle_leiden.restype = C.c_int
        lib.oracle_leiden.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                      C.c_int, C.c_uint64, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        lib.oracle_modularity.restype = C.c_double
        lib.oracle_modularity.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        _lib = lib
    return _lib


def _parts(adj):
    adj = sparse.csr_matrix(adj)
    indptr = np.ascontiguousarray(adj.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(adj.indices, dtype=np.int32)
    w = np.ascontiguousarray(adj.data, dtype=np.float64)
    return adj.shape[0], indptr, indices, w


def leiden(adj, *, resolution: float = 1.0, n_iterations: int = -1, seed: int = 0, beta: float = 0.01):
    """-> (membership int32 [n] ordered by decreasing size, modularity)."""
    n, indptr, indices, w = _parts(adj)
    memb = np.empty(n, dtype=np.int32)
    q = C.c_double(0)
    nc = C.c_int32(0)
    rc = _load().oracle_leiden(n, indptr.ctypes.data, indices.ctypes.data, w.ctypes.data, float(resolution),
                               float(beta), int(n_iterations), int(seed) & (2**64 - 1), memb.ctypes.data,
                               C.byref(q), C.byref(nc))
    if rc != 0:
        raise RuntimeError(f"oracle_leiden failed: {rc}")
    return memb, float(q.value)


def modularity(adj, membership, *, resolution: float = 1.0) -> float:
    n, indptr, indices, w = _parts(adj)
    memb = np.ascontiguousarray(membership, dtype=np.int32)
    return float(_load().oracle_modularity(n, indptr.ctypes.data, indices.ctypes.data, w.ctypes.data,
                                           memb.ctypes.data, float(resolution)))
