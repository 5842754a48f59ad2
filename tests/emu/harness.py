"""TEST INFRASTRUCTURE: numpy-level callers of the C ABI of the HOST-EMULATED kernel library
(tests/emu/_build/*/libscanpy_amd_emu.so).  The emulated library takes host pointers where the product takes device
pointers; prototypes come from scanpy_amd._lib.SIGNATURES (the same table the product binds with)."""
from __future__ import annotations

import ctypes as C
import os
import sys
from functools import lru_cache
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from scanpy_amd._lib import SIGNATURES  # noqa: E402


@lru_cache(maxsize=2)
def load(asan: bool = False) -> C.CDLL:
    sys.path.insert(0, str(HERE))
    import build as emu_build

    lib = C.CDLL(str(emu_build.build(asan=asan)))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for name in ("emu_launches", "emu_partial_collectives", "emu_mixed_collectives", "emu_reads_of_inactive_lanes"):
        getattr(lib, name).restype = C.c_longlong
    lib.emu_user_counter.restype, lib.emu_user_counter.argtypes = C.c_longlong, (C.c_int,)
    lib.emu_set_dma_late.restype, lib.emu_set_dma_late.argtypes = None, (C.c_int,)
    return lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _check(lib, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: rc={rc}: {lib.scamd_last_error().decode()}")


def _ws(nbytes: int):
    # exact size, its own allocation: an over-run past the workspace is an over-run of a heap block
    return np.full(max(int(nbytes), 1), 0xAB, dtype=np.uint8)


def stats(lib) -> dict:
    return {"launches": lib.emu_launches(), "partial_collectives": lib.emu_partial_collectives(),
            "mixed_collectives": lib.emu_mixed_collectives(), "reads_of_inactive_lanes": lib.emu_reads_of_inactive_lanes()}


def user_counters(lib, n: int = 16) -> list:
    """the event counters the kernels bump under SCAMD_EMU (emu_runtime.cpp: emu_user_counters)"""
    return [int(lib.emu_user_counter(i)) for i in range(n)]


def fuzzy_simplicial_set(lib, idx, dist):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    n, k = idx.shape
    cap = 2 * n * (k - 1)
    indptr = np.empty(n + 1, dtype=np.int64)
    indices = np.empty(cap, dtype=np.int32)
    data = np.empty(cap, dtype=np.float32)
    sigma = np.empty(n, dtype=np.float32)
    rho = np.empty(n, dtype=np.float32)
    ws = _ws(lib.scamd_fuzzy_workspace_bytes(n, k))
    nnz = C.c_int64(0)
    rc = lib.scamd_fuzzy_simplicial_set_f32(_p(idx), _p(dist), n, k, _p(indptr), _p(indices), _p(data), cap, _p(sigma), _p(rho),
                                            C.byref(nnz), _p(ws), ws.size, None)
    _check(lib, rc, "fuzzy")
    m = int(nnz.value)
    return indptr, indices[:m].copy(), data[:m].copy(), sigma, rho


def leiden(lib, adj, *, resolution=1.0, n_iterations=-1, beta=0.01, seed=0, initial_membership=None, objective=0, node_weights=None):
    adj = adj.tocsr()
    adj.sort_indices()
    n = adj.shape[0]
    indptr = np.ascontiguousarray(adj.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(adj.indices, dtype=np.int32)
    w = np.ascontiguousarray(adj.data, dtype=np.float32)
    memb = np.empty(n, dtype=np.int32)
    q = C.c_double(0)
    nc = C.c_int32(0)
    ws = _ws(lib.scamd_leiden_workspace_bytes(n, adj.nnz))
    if node_weights is not None:
        init = None if initial_membership is None else np.ascontiguousarray(initial_membership, dtype=np.int32)
        nw = np.ascontiguousarray(node_weights, dtype=np.float32)
        rc = lib.scamd_leiden_csr_nw_f32(_p(indptr), _p(indices), _p(w), n, adj.nnz, float(resolution), int(n_iterations), float(beta),
                                         int(seed), int(objective), _p(nw), None if init is None else _p(init), _p(memb),
                                         C.byref(q), C.byref(nc), _p(ws), ws.size, None)
    elif objective:
        init = None if initial_membership is None else np.ascontiguousarray(initial_membership, dtype=np.int32)
        rc = lib.scamd_leiden_csr_ex_f32(_p(indptr), _p(indices), _p(w), n, adj.nnz, float(resolution), int(n_iterations), float(beta),
                                         int(seed), int(objective), None if init is None else _p(init), _p(memb), C.byref(q),
                                         C.byref(nc), _p(ws), ws.size, None)
    elif initial_membership is not None:
        init = np.ascontiguousarray(initial_membership, dtype=np.int32)
        rc = lib.scamd_leiden_csr_init_f32(_p(indptr), _p(indices), _p(w), n, adj.nnz, float(resolution), int(n_iterations),
                                           float(beta), int(seed), _p(init), _p(memb), C.byref(q), C.byref(nc), _p(ws), ws.size, None)
    else:
        rc = lib.scamd_leiden_csr_f32(_p(indptr), _p(indices), _p(w), n, adj.nnz, float(resolution), int(n_iterations), float(beta),
                                      int(seed), _p(memb), C.byref(q), C.byref(nc), _p(ws), ws.size, None)
    _check(lib, rc, "leiden")
    return memb, float(q.value), int(nc.value)


def knn(lib, x, k, *, q_begin=0, n_query=None, cert_scale=1.0, nprobe=0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    nq = n if n_query is None else n_query
    idx = np.empty((nq, k), dtype=np.int32)
    dist = np.empty((nq, k), dtype=np.float64)
    ws = _ws(lib.scamd_knn_workspace_bytes(n, d, nq, k))
    nfb = C.c_int64(0)
    if nprobe:
        rc = lib.scamd_knn_l2_ivf_f32(_p(x), n, d, d, q_begin, nq, k, int(nprobe), _p(idx), _p(dist), C.byref(nfb), _p(ws), ws.size, None)
    else:
        rc = lib.scamd_knn_l2_f32(_p(x), n, d, d, q_begin, nq, k, _p(idx), _p(dist), float(cert_scale), C.byref(nfb), _p(ws), ws.size, None)
    _check(lib, rc, "knn")
    return idx, dist, int(nfb.value)


def leiden_split(lib, adj, membership):
    """scamd_leiden_debug_split_f32 -> (membership after the split, components - communities)"""
    adj = adj.tocsr()
    adj.sort_indices()
    n = adj.shape[0]
    indptr = np.ascontiguousarray(adj.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(adj.indices, dtype=np.int32)
    w = np.ascontiguousarray(adj.data, dtype=np.float32)
    memb = np.ascontiguousarray(membership, dtype=np.int32).copy()
    ns = C.c_int32(0)
    ws = _ws(lib.scamd_leiden_workspace_bytes(n, adj.nnz))
    rc = lib.scamd_leiden_debug_split_f32(_p(indptr), _p(indices), _p(w), n, adj.nnz, _p(memb), C.byref(ns), _p(ws), ws.size, None)
    _check(lib, rc, "leiden split")
    return memb, int(ns.value)


def leiden_stats(lib) -> dict:
    out = (C.c_int32 * 12)()
    lib.scamd_leiden_last_stats(out, 12)
    keys = ("iterations", "launches", "host_round_trips", "polish_full_sweeps", "polish_rounds", "polish_moves",
            "polish_skipped_proven", "levels_first_iteration", "lm_sweeps", "lm_sweep_algorithmic_MB", "polish_splits", "ended_by_iteration_cap")
    return dict(zip(keys, (int(v) for v in out)))


def pca_csr(lib, x, n_comps, *, zero_center=True, seed=0, tol=2e-8):
    """scamd_pca_csr_f32 on a scipy CSR float32 matrix -> dict(scores, components, variance, variance_ratio, mean, info)"""
    x = x.tocsr()
    x.sort_indices()
    n, g = x.shape
    indptr = np.ascontiguousarray(x.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(x.indices, dtype=np.int32)
    data = np.ascontiguousarray(x.data, dtype=np.float32)
    scores = np.empty((n, n_comps), dtype=np.float32)
    comps = np.empty((n_comps, g), dtype=np.float64)
    var = np.empty(n_comps, dtype=np.float64)
    ratio = np.empty(n_comps, dtype=np.float64)
    mean = np.empty(g, dtype=np.float64)
    info = np.zeros(8, dtype=np.int32)
    ws = _ws(lib.scamd_pca_csr_workspace_bytes(n, g, n_comps))
    rc = lib.scamd_pca_csr_f32(_p(indptr), _p(indices), _p(data), n, g, x.nnz, n_comps, int(zero_center), int(seed), float(tol),
                               _p(scores), _p(comps), _p(var), _p(ratio), _p(mean), _p(info), _p(ws), ws.size, None)
    _check(lib, rc, "pca_csr")
    return dict(scores=scores, components=comps, variance=var, variance_ratio=ratio, mean=mean, info=info)


def csr_gram(lib, x, scale_bits=None):
    x = x.tocsr()
    x.sort_indices()
    n, g = x.shape
    indptr = np.ascontiguousarray(x.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(x.indices, dtype=np.int32)
    data = np.ascontiguousarray(x.data, dtype=np.float32)
    ws = _ws(lib.scamd_csr_gram_workspace_bytes(n, g))
    absmax = C.c_float(0)
    rc = lib.scamd_csr_gram_f32(_p(indptr), _p(indices), _p(data), n, g, x.nnz, 0, None, 0, None, C.byref(absmax), _p(ws), ws.size, None)
    _check(lib, rc, "gram absmax")
    if scale_bits is None:
        scale_bits = int(np.floor(62 - np.log2(n * float(absmax.value) ** 2))) - 1
    g_pad = (g + 127) // 128 * 128
    gram = np.zeros((g_pad, g_pad), dtype=np.int64)
    colsum = np.zeros(g_pad, dtype=np.int64)
    rc = lib.scamd_csr_gram_f32(_p(indptr), _p(indices), _p(data), n, g, x.nnz, scale_bits, _p(gram), g_pad, _p(colsum), None, _p(ws), ws.size, None)
    _check(lib, rc, "gram")
    return gram, colsum, scale_bits, float(absmax.value)


def modularity(lib, adj, membership, resolution=1.0):
    adj = adj.tocsr()
    n = adj.shape[0]
    indptr = np.ascontiguousarray(adj.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(adj.indices, dtype=np.int32)
    w = np.ascontiguousarray(adj.data, dtype=np.float32)
    memb = np.ascontiguousarray(membership, dtype=np.int32)
    q = C.c_double(0)
    ws = _ws(lib.scamd_leiden_workspace_bytes(n, adj.nnz))
    rc = lib.scamd_modularity_csr_f32(_p(indptr), _p(indices), _p(w), n, adj.nnz, _p(memb), float(resolution), C.byref(q), _p(ws), ws.size, None)
    _check(lib, rc, "modularity")
    return float(q.value)
