"""Thin torch-tensor wrappers over the C ABI (include/scanpy_amd.h).  No arithmetic happens here."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._device import ptr, require_gpu, stream_ptr, workspace_pool

import os as _os

# SCAMD_POISON_WORKSPACE=1: the shared scratch buffer is filled with 0xAB bytes before every C call, so that a kernel
# which reads scratch it did not write fails the same way on every box (round 3: such a read showed up as a device
# fault on some boxes only)
_POISON = _os.environ.get("SCAMD_POISON_WORKSPACE") == "1"
# SCAMD_GUARD=1 (debug; round 4's hunt for the unreproduced device faults): every output tensor and every workspace handed
# to a C entry point sits between two 4 KiB guards filled with a canary byte; after the call (and a stream sync) the guards
# are read back -- a kernel that WRITES past either end of a buffer it was given raises here, naming the entry point.
_GUARD = _os.environ.get("SCAMD_GUARD") == "1"
_GUARD_BYTES = 4096
_CANARY = 0xC5
_guards: list = []  # (label, raw uint8 buffer, payload bytes) of the call in progress


def _guarded(nbytes: int, dev: torch.device, label: str) -> torch.Tensor:
    pay = (int(nbytes) + 255) // 256 * 256
    raw = torch.empty(pay + 2 * _GUARD_BYTES, dtype=torch.uint8, device=dev)
    raw[:_GUARD_BYTES] = _CANARY
    raw[_GUARD_BYTES + int(nbytes):] = _CANARY  # (the slack up to the 256-byte boundary belongs to the guard)
    _guards.append((label, raw, int(nbytes)))
    return raw[_GUARD_BYTES:_GUARD_BYTES + int(nbytes)]


def _empty(shape, *, dtype, device):
    if not _GUARD:
        return torch.empty(shape, dtype=dtype, device=device)
    shape = (int(shape),) if not isinstance(shape, (tuple, list, torch.Size)) else tuple(int(v) for v in shape)
    numel = 1
    for v in shape:
        numel *= v
    item = torch.empty((), dtype=dtype).element_size()
    return _guarded(numel * item, device, f"{dtype} {shape}").view(dtype).view(shape)


def _zeros(shape, *, dtype, device):
    t = _empty(shape, dtype=dtype, device=device)
    if _GUARD:
        t.zero_()
        return t
    return torch.zeros(shape, dtype=dtype, device=device)


def _check(rc: int, what: str = "") -> None:
    # the guard list is taken and cleared FIRST: when the C entry point reports an error the guarded buffers must not
    # stay referenced (device memory held) nor be attributed to the next, unrelated call
    pending = list(_guards)
    _guards.clear()
    _lib.check(rc, what)
    if _GUARD and pending:
        torch.cuda.synchronize()
        bad = []
        for label, raw, nbytes in pending:
            lo_ok = bool((raw[:_GUARD_BYTES] == _CANARY).all())
            hi_ok = bool((raw[_GUARD_BYTES + nbytes:] == _CANARY).all())
            if not (lo_ok and hi_ok):
                bad.append(f"{label} ({nbytes} B): {'below' if not lo_ok else ''}{' above' if not hi_ok else ''}")
        if bad:
            raise _lib.ScamdError(f"SCAMD_GUARD: {what} wrote outside " + "; ".join(bad))


def _ws(nbytes: int, dev: torch.device):
    if _GUARD:
        buf = _guarded(max(int(nbytes), 1), dev, "workspace")
        if _POISON:
            buf.fill_(0xAB)
        return buf, C.c_size_t(buf.numel())
    buf = workspace_pool.get(nbytes, dev)
    if _POISON:  # debug: every entry point must initialise what it reads (a dirty workspace is the normal case)
        buf.fill_(0xAB)
    return buf, C.c_size_t(buf.numel())


def mfma_selftest() -> None:
    require_gpu()
    _check(_lib.load().scamd_selftest_mfma_layout(stream_ptr()), "mfma selftest")


def knn(x: torch.Tensor, k: int, *, q_begin: int = 0, n_query: int | None = None, cert_scale: float = 1.0,
        nprobe: int | None = None):
    """x [n, d] float32 (device).  -> (idx int32 [nq, k], dist float64 [nq, k], n_fallback).

    nprobe: None / 0 = the exact search (scamd_knn_l2_f32); > 0 = the approximate IVF mode (scamd_knn_l2_ivf_f32): every
    query sees the rows of the `nprobe` cells nearest to its own cell only."""
    dev = require_gpu()
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_cuda
    x = x.contiguous()
    n, d = x.shape
    nq = n - q_begin if n_query is None else n_query
    idx = _empty((nq, k), dtype=torch.int32, device=dev)
    dist = _empty((nq, k), dtype=torch.float64, device=dev)
    need = lib.scamd_knn_workspace_bytes(n, d, nq, k)
    if need == 0 and nq > 0:
        raise _lib.ScamdError(f"knn: unsupported shape d={d} (max 256) / k={k} (max 256)")
    ws, wsz = _ws(need, dev)
    nfb = C.c_int64(0)
    if nprobe:
        if cert_scale != 1.0:
            raise ValueError("knn: cert_scale is a test knob of the exact entry point")
        rc = lib.scamd_knn_l2_ivf_f32(ptr(x), n, d, x.stride(0), q_begin, nq, k, int(nprobe), ptr(idx), ptr(dist),
                                      C.byref(nfb), ptr(ws), wsz, stream_ptr())
        _check(rc, "scamd_knn_l2_ivf_f32")
        if lib.scamd_knn_last_nprobe() == 0:
            import warnings

            warnings.warn(f"knn: the approximate mode (nprobe={int(nprobe)}) does not take this shape (n={n}, d={d}, k={k}: it "
                          "needs n >= 4096, d <= 64, k <= 24); the search was answered EXACTLY.", UserWarning, stacklevel=3)
    else:
        rc = lib.scamd_knn_l2_f32(ptr(x), n, d, x.stride(0), q_begin, nq, k, ptr(idx), ptr(dist),
                                  float(cert_scale), C.byref(nfb), ptr(ws), wsz, stream_ptr())
        _check(rc, "scamd_knn_l2_f32")
    return idx, dist, int(nfb.value)


def knn_cert_factors(engine: int = 1):
    """the certificate's error-bound factors (cert_k, cert_k2, key_slack) of an engine, in units of u = 2^-24"""
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    _lib.load().scamd_knn_cert_factors(int(engine), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def knn_debug_b3_scores(x: torch.Tensor, q0: int, nq: int, c0: int, nc: int):
    """TEST entry: raw scores ||c||^2 - 2 q.c (centred frame) of the bf16 engine for queries [q0, q0 + nq) x candidates
    [c0, c0 + nc) (multiples of 32), through the select kernel's own packing and MFMA chain.
    -> (scores float32 [nq, nc], mu float32 [d], cmax float)"""
    dev = require_gpu()
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_cuda
    x = x.contiguous()
    n, d = x.shape
    out = _empty((nq, nc), dtype=torch.float32, device=dev)
    mu = _empty(128, dtype=torch.float32, device=dev)
    cmax = _empty(1, dtype=torch.float32, device=dev)
    ws, wsz = _ws(lib.scamd_knn_debug_b3_scores_workspace_bytes(n), dev)
    rc = lib.scamd_knn_debug_b3_scores_f32(ptr(x), n, d, x.stride(0), q0, nq, c0, nc, ptr(out), ptr(mu), ptr(cmax),
                                           ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_knn_debug_b3_scores_f32")
    return out, mu[:d], float(cmax.item())


def fuzzy_simplicial_set(knn_idx: torch.Tensor, knn_dist: torch.Tensor):
    """-> (indptr int64 [n+1], indices int32 [nnz], data float32 [nnz], sigma [n], rho [n])."""
    dev = require_gpu()
    lib = _lib.load()
    n, k = knn_idx.shape
    knn_idx = knn_idx.to(torch.int32).contiguous()
    knn_dist = knn_dist.to(torch.float32).contiguous()
    cap = 2 * n * (k - 1)
    indptr = _empty(n + 1, dtype=torch.int64, device=dev)
    indices = _empty(cap, dtype=torch.int32, device=dev)
    data = _empty(cap, dtype=torch.float32, device=dev)
    sigma = _empty(n, dtype=torch.float32, device=dev)
    rho = _empty(n, dtype=torch.float32, device=dev)
    ws, wsz = _ws(lib.scamd_fuzzy_workspace_bytes(n, k), dev)
    nnz = C.c_int64(0)
    rc = lib.scamd_fuzzy_simplicial_set_f32(ptr(knn_idx), ptr(knn_dist), n, k, ptr(indptr), ptr(indices), ptr(data),
                                            cap, ptr(sigma), ptr(rho), C.byref(nnz), ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_fuzzy_simplicial_set_f32")
    m = int(nnz.value)
    return indptr, indices[:m], data[:m], sigma, rho


def fuzzy_weights(knn_idx: torch.Tensor, knn_dist: torch.Tensor, row_begin: int, n_total: int, sum_all: torch.Tensor):
    """membership strengths of a rank's own rows (row-sharded fuzzy set, step 1).  knn_idx holds global row ids;
    sum_all = device float64 [1], the sum of all n_total * k distances.  -> w float32 [n_local, k] (0 = absent)"""
    dev = require_gpu()
    n, k = knn_idx.shape
    knn_idx = knn_idx.to(torch.int32).contiguous()
    knn_dist = knn_dist.to(torch.float32).contiguous()
    sum_all = sum_all.to(torch.float64).contiguous()
    w = _empty((n, k), dtype=torch.float32, device=dev)
    cnt = _empty(max(n, 1), dtype=torch.int32, device=dev)
    _check(_lib.load().scamd_fuzzy_weights_f32(ptr(knn_idx), ptr(knn_dist), n, k, int(row_begin), int(n_total),
                                                   ptr(sum_all), ptr(w), None, None, ptr(cnt), stream_ptr()),
               "scamd_fuzzy_weights_f32")
    return w


def fuzzy_merge_rows(knn_idx: torch.Tensor, w: torch.Tensor, in_indptr: torch.Tensor, in_src: torch.Tensor,
                     in_w: torch.Tensor):
    """symmetrised rows of a rank (row-sharded fuzzy set, step 3) from its out-edges (knn_idx, w) and the in-edges the
    other ranks sent, sorted by (row, source).  -> (indptr int64 [n_local + 1], indices int32 (global), data float32)"""
    dev = require_gpu()
    lib = _lib.load()
    n, k = knn_idx.shape
    knn_idx = knn_idx.to(torch.int32).contiguous()
    w = w.to(torch.float32).contiguous()
    in_indptr = in_indptr.to(torch.int64).contiguous()
    in_src = in_src.to(torch.int32).contiguous()
    in_w = in_w.to(torch.float32).contiguous()
    cap = n * (k - 1) + int(in_src.numel())
    indptr = _empty(n + 1, dtype=torch.int64, device=dev)
    indices = _empty(max(cap, 1), dtype=torch.int32, device=dev)
    data = _empty(max(cap, 1), dtype=torch.float32, device=dev)
    ws, wsz = _ws(lib.scamd_fuzzy_merge_workspace_bytes(n, cap), dev)
    nnz = C.c_int64(0)
    rc = lib.scamd_fuzzy_merge_rows_f32(ptr(knn_idx), ptr(w), n, k, ptr(in_indptr), ptr(in_src), ptr(in_w), ptr(indptr),
                                        ptr(indices), ptr(data), cap, C.byref(nnz), ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_fuzzy_merge_rows_f32")
    return indptr, indices[: nnz.value], data[: nnz.value]


def _knn_graph(entry: str, knn_idx: torch.Tensor, knn_dist: torch.Tensor | None):
    """shared driver of the gauss / jaccard connectivity kernels -> (indptr int64, indices int32, data float32)"""
    dev = require_gpu()
    lib = _lib.load()
    n, k = knn_idx.shape
    knn_idx = knn_idx.to(torch.int32).contiguous()
    cap = 2 * n * (k - 1)
    indptr = _empty(n + 1, dtype=torch.int64, device=dev)
    indices = _empty(cap, dtype=torch.int32, device=dev)
    data = _empty(cap, dtype=torch.float32, device=dev)
    ws, wsz = _ws(lib.scamd_fuzzy_workspace_bytes(n, k), dev)
    nnz = C.c_int64(0)
    if entry == "gauss":
        knn_dist = knn_dist.to(torch.float32).contiguous()
        rc = lib.scamd_gauss_connectivities_f32(ptr(knn_idx), ptr(knn_dist), n, k, ptr(indptr), ptr(indices), ptr(data), cap,
                                                C.byref(nnz), ptr(ws), wsz, stream_ptr())
    else:
        rc = lib.scamd_jaccard_connectivities_f32(ptr(knn_idx), n, k, ptr(indptr), ptr(indices), ptr(data), cap,
                                                  C.byref(nnz), ptr(ws), wsz, stream_ptr())
    _check(rc, f"scamd_{entry}_connectivities_f32")
    m = int(nnz.value)
    return indptr, indices[:m], data[:m]


def gauss_connectivities(knn_idx: torch.Tensor, knn_dist: torch.Tensor):
    return _knn_graph("gauss", knn_idx, knn_dist)


def jaccard_connectivities(knn_idx: torch.Tensor):
    return _knn_graph("jaccard", knn_idx, None)


def csr_transpose(indptr: torch.Tensor, indices: torch.Tensor, data: torch.Tensor, n: int, g: int):
    dev = require_gpu()
    lib = _lib.load()
    nnz = data.numel()
    t_indptr = _empty(g + 1, dtype=torch.int64, device=dev)
    t_indices = _empty(nnz, dtype=torch.int32, device=dev)
    t_data = _empty(nnz, dtype=torch.float32, device=dev)
    ws, wsz = _ws(lib.scamd_csr_transpose_workspace_bytes(n, g, nnz), dev)
    rc = lib.scamd_csr_transpose_f32(ptr(indptr), ptr(indices), ptr(data), n, g, nnz, ptr(t_indptr), ptr(t_indices),
                                     ptr(t_data), ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_csr_transpose_f32")
    return t_indptr, t_indices, t_data


def csr_row_stats(indptr: torch.Tensor, data: torch.Tensor, n_rows: int):
    dev = require_gpu()
    s = _empty(n_rows, dtype=torch.float64, device=dev)
    q = _empty(n_rows, dtype=torch.float64, device=dev)
    _check(_lib.load().scamd_csr_row_stats_f32(ptr(indptr), ptr(data), n_rows, ptr(s), ptr(q), stream_ptr()),
               "scamd_csr_row_stats_f32")
    return s, q


def csr_absmax(indptr, indices, data, n: int, g: int) -> float:
    """max |x| over the stored entries (phase 1 of the Gram computation)."""
    dev = require_gpu()
    lib = _lib.load()
    ws, wsz = _ws(lib.scamd_csr_gram_workspace_bytes(n, g), dev)
    mx = C.c_float(0.0)
    rc = lib.scamd_csr_gram_f32(ptr(indptr), ptr(indices), ptr(data), n, g, data.numel(), 0, None, 0, None,
                                C.byref(mx), ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_csr_gram_f32 (absmax)")
    return float(mx.value)


def csr_gram(indptr, indices, data, n: int, g: int, scale_bits: int):
    """-> (gram int64 [g_pad, g_pad], colsum int64 [g_pad]) in fixed point (x 2^scale_bits), g_pad = ceil128(g)."""
    dev = require_gpu()
    lib = _lib.load()
    gp = (g + 127) // 128 * 128
    gram = _empty((gp, gp), dtype=torch.int64, device=dev)
    colsum = _empty(gp, dtype=torch.int64, device=dev)
    ws, wsz = _ws(lib.scamd_csr_gram_workspace_bytes(n, g), dev)
    rc = lib.scamd_csr_gram_f32(ptr(indptr), ptr(indices), ptr(data), n, g, data.numel(), int(scale_bits), ptr(gram),
                                gp, ptr(colsum), None, ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_csr_gram_f32")
    return gram, colsum


def spmm(indptr, indices, data, n: int, g: int, b: torch.Tensor, shift: torch.Tensor | None = None) -> torch.Tensor:
    """Y[n, l] = A B - 1 shift^T, float32."""
    dev = require_gpu()
    b = b.to(torch.float32).contiguous()
    assert b.shape[0] == g
    l = b.shape[1]
    y = _empty((n, l), dtype=torch.float32, device=dev)
    if shift is not None:
        shift = shift.to(torch.float32).contiguous()
    _check(_lib.load().scamd_spmm_csr_f32(ptr(indptr), ptr(indices), ptr(data), n, g, ptr(b), l, ptr(shift),
                                              ptr(y), stream_ptr()), "scamd_spmm_csr_f32")
    return y


def spmm_f64acc(indptr, indices, data, n_rows: int, b: torch.Tensor, scale: torch.Tensor | None = None,
                colsum: torch.Tensor | None = None) -> torch.Tensor:
    """W[n_rows, l] float64 = A B (float64 accumulation) - scale colsum^T."""
    dev = require_gpu()
    lib = _lib.load()
    assert b.dtype == torch.float32 and b.is_contiguous()
    l = b.shape[1]
    nnz = data.numel()
    w = _empty((n_rows, l), dtype=torch.float64, device=dev)
    ws, wsz = _ws(lib.scamd_spmm_f64acc_workspace_bytes(n_rows, nnz, l), dev)
    rc = lib.scamd_spmm_csr_f32_f64acc(ptr(indptr), ptr(indices), ptr(data), n_rows, nnz, ptr(b), l, ptr(scale),
                                       ptr(colsum), ptr(w), ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_spmm_csr_f32_f64acc")
    return w


def dense_debug(op: int, in0: torch.Tensor, in1: torch.Tensor | None = None):
    """building blocks of the dense eigensolver on caller buffers (csrc/dense.hip; tests).  op 1: in0^T in1;
    op 2: CholeskyQR factor of a Gram matrix -> (S, pivot_failed); op 3: Jacobi eigh -> (theta, Y, sweeps)."""
    dev = require_gpu()
    lib = _lib.load()
    in0 = in0.to(torch.float64).contiguous()
    flag = C.c_int(0)
    if op == 1:
        in1 = in1.to(torch.float64).contiguous()
        kdim, m = in0.shape
        n = in1.shape[1]
        out = _empty((m, n), dtype=torch.float64, device=dev)
        rc = lib.scamd_dense_debug_f64(1, ptr(in0), ptr(in1), m, n, kdim, ptr(out), None, C.byref(flag), stream_ptr())
        _check(rc, "scamd_dense_debug_f64")
        return out
    m = in0.shape[0]
    out0 = _empty((m, m) if op == 2 else (m,), dtype=torch.float64, device=dev)
    out1 = _empty((m, m), dtype=torch.float64, device=dev)
    rc = lib.scamd_dense_debug_f64(op, ptr(in0), None, m, m, m, ptr(out0), ptr(out1), C.byref(flag), stream_ptr())
    _check(rc, "scamd_dense_debug_f64")
    return (out0, int(flag.value)) if op == 2 else (out0, out1, int(flag.value))


def eigh_topk(a: torch.Tensor, k: int, *, seed: int = 0, tol: float = 2e-8):
    """Top-k eigenpairs of the symmetric PSD float64 matrix a [g, g] -> (lam [k] descending, v [g, k], info dict).
    Raises ScamdError (SCAMD_EUNSUPPORTED) outside the device solver's range or when it does not converge."""
    dev = require_gpu()
    lib = _lib.load()
    assert a.dtype == torch.float64 and a.dim() == 2 and a.shape[0] == a.shape[1] and a.is_cuda
    a = a.contiguous()
    g = a.shape[0]
    lam = _empty(k, dtype=torch.float64, device=dev)
    v = _empty((g, k), dtype=torch.float64, device=dev)
    info = (C.c_int32 * 12)()
    need = lib.scamd_eigh_topk_workspace_bytes(g, k)
    ws, wsz = _ws(need, dev)
    rc = lib.scamd_eigh_topk_f64(ptr(a), g, a.stride(0), k, int(seed) & (2**64 - 1), float(tol), ptr(lam), ptr(v), info,
                                 ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_eigh_topk_f64")
    resid = C.cast(C.byref(info, 16), C.POINTER(C.c_double))[0]
    return lam, v, {"n_outer": info[0], "n_gemm": info[1], "block_size": info[2], "chol_retries": info[3], "residual": resid}


def pca_csr(indptr, indices, data, n: int, g: int, n_comps: int, *, zero_center: bool = True, seed: int = 0,
            tol: float = 2e-8):
    """`scamd_pca_csr_f32`: the whole Gram-route PCA of a resident CSR matrix in one C call.
    -> (scores f32 [n, k], components f64 [k, g], variance [k], variance_ratio [k], mean [g], info dict)"""
    dev = require_gpu()
    lib = _lib.load()
    k = int(n_comps)
    scores = _empty((n, k), dtype=torch.float32, device=dev)
    comps = _empty((k, g), dtype=torch.float64, device=dev)
    var = _empty(k, dtype=torch.float64, device=dev)
    ratio = _empty(k, dtype=torch.float64, device=dev)
    mean = _empty(g, dtype=torch.float64, device=dev)
    info = (C.c_int32 * 12)()
    need = lib.scamd_pca_csr_workspace_bytes(n, g, k)
    ws, wsz = _ws(need, dev)
    rc = lib.scamd_pca_csr_f32(ptr(indptr), ptr(indices), ptr(data), n, g, data.numel(), k, 1 if zero_center else 0,
                               int(seed) & (2**64 - 1), float(tol), ptr(scores), ptr(comps), ptr(var), ptr(ratio), ptr(mean),
                               info, ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_pca_csr_f32")
    resid = C.cast(C.byref(info, 16), C.POINTER(C.c_double))[0]
    return scores, comps, var, ratio, mean, {"n_outer": info[0], "n_gemm": info[1], "block_size": info[2],
                                             "chol_retries": info[3], "residual": resid, "scale_bits": info[6]}


def pca_solve_gram(gram_q: torch.Tensor, colsum_q: torch.Tensor, n_total: int, g: int, scale_bits: int, n_comps: int, *,
                   zero_center: bool = True, seed: int = 0, tol: float = 2e-8):
    """`scamd_pca_solve_gram_f64`: the dense half of the Gram route in one C call (no torch arithmetic): int64 Gram matrix
    [gp, gp] + column sums -> (components f64 [k, g], loadings f32 [g, k], shift f32 [k], variance [k], ratio [k],
    mean f64 [g], eigenvalues f64 [k], info)."""
    dev = require_gpu()
    lib = _lib.load()
    k = int(n_comps)
    assert gram_q.dtype == torch.int64 and gram_q.is_contiguous() and colsum_q.dtype == torch.int64
    comps = _empty((k, g), dtype=torch.float64, device=dev)
    v32 = _empty((g, k), dtype=torch.float32, device=dev)
    shift = _empty(k, dtype=torch.float32, device=dev)
    var = _empty(k, dtype=torch.float64, device=dev)
    ratio = _empty(k, dtype=torch.float64, device=dev)
    mean = _empty(g, dtype=torch.float64, device=dev)
    lam = _empty(k, dtype=torch.float64, device=dev)
    info = (C.c_int32 * 12)()
    ws, wsz = _ws(lib.scamd_pca_solve_gram_workspace_bytes(g, k), dev)
    rc = lib.scamd_pca_solve_gram_f64(ptr(gram_q), gram_q.shape[1], ptr(colsum_q), int(n_total), g, int(scale_bits), k,
                                      1 if zero_center else 0, int(seed) & (2**64 - 1), float(tol), ptr(comps), ptr(v32),
                                      ptr(shift), ptr(var), ptr(ratio), ptr(mean), ptr(lam), info, ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_pca_solve_gram_f64")
    resid = C.cast(C.byref(info, 16), C.POINTER(C.c_double))[0]
    return comps, v32, shift, var, ratio, mean, lam, {"n_outer": info[0], "n_gemm": info[1], "block_size": info[2],
                                                       "chol_retries": info[3], "residual": resid}


def spectral_embedding(indptr: torch.Tensor, indices: torch.Tensor, weights: torch.Tensor, n: int, dim: int, *, seed: int = 0,
                       tol: float = 2e-6, max_outer: int = 60, max_degree: int = 64):
    """`scamd_spectral_embedding_f32`: the `dim` eigenvectors of D^-1/2 A D^-1/2 below the trivial one, float64 [n, dim] on the
    device, + info (outer_iterations, operator_applications, residual, converged, ritz_values)."""
    dev = require_gpu()
    lib = _lib.load()
    indptr = indptr.to(torch.int64).contiguous()
    indices = indices.to(torch.int32).contiguous()
    weights = weights.to(torch.float32).contiguous()
    nnz = weights.numel()
    out = _empty((n, dim), dtype=torch.float64, device=dev)
    need = lib.scamd_spectral_embedding_workspace_bytes(n, nnz, int(dim))
    if need == 0:
        raise _lib.ScamdError(f"spectral_embedding: unsupported shape n={n} dim={dim} (at most 10 components)")
    ws, wsz = _ws(need, dev)
    info = (C.c_double * 8)()
    rc = lib.scamd_spectral_embedding_f32(ptr(indptr), ptr(indices), ptr(weights), n, nnz, int(dim), int(seed) & (2**64 - 1),
                                          float(tol), int(max_outer), int(max_degree), ptr(out), info, ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_spectral_embedding_f32")
    return out, {"outer_iterations": int(info[0]), "operator_applications": int(info[1]), "residual": float(info[2]),
                 "converged": bool(info[3] > 0.5), "ritz_values": [float(info[4 + j]) for j in range(min(dim, 4))]}


def colsum(y: torch.Tensor) -> torch.Tensor:
    dev = require_gpu()
    lib = _lib.load()
    assert y.dtype == torch.float32 and y.is_contiguous()
    n, l = y.shape
    out = _empty(l, dtype=torch.float64, device=dev)
    ws, wsz = _ws(lib.scamd_colsum_workspace_bytes(l), dev)
    _check(lib.scamd_colsum_f32_f64(ptr(y), n, l, ptr(out), ptr(ws), wsz, stream_ptr()), "scamd_colsum_f32_f64")
    return out


def leiden(indptr: torch.Tensor, indices: torch.Tensor, weights: torch.Tensor, n: int, *, resolution: float = 1.0,
           n_iterations: int = -1, beta: float = 0.01, seed: int = 0, initial_membership: torch.Tensor | None = None,
           objective: str = "modularity", node_weights: torch.Tensor | None = None):
    """Symmetric CSR graph on device -> (membership int32 [n] on device, modularity, n_communities).
    `initial_membership` (int32 [n] on the device, ids in [0, n)): start from that partition instead of singletons.
    `objective`: 'modularity' (resolution normalised by 2m, vertex weight = strength) or 'cpm' (igraph's CPM: vertex weight
    1, resolution as given; the returned float is then the resolution-1 modularity of the partition).
    `node_weights` (float32 [n] in [0, 1e6], CPM only): igraph's `node_weights` -- a community pays resolution x (sum of its
    members' weights)^2."""
    dev = require_gpu()
    lib = _lib.load()
    indptr = indptr.to(torch.int64).contiguous()
    indices = indices.to(torch.int32).contiguous()
    weights = weights.to(torch.float32).contiguous()
    nnz = weights.numel()
    memb = _empty(n, dtype=torch.int32, device=dev)
    ws, wsz = _ws(lib.scamd_leiden_workspace_bytes(n, nnz), dev)
    q = C.c_double(0.0)
    nc = C.c_int32(0)
    init = None
    if initial_membership is not None:
        init = initial_membership.to(device=dev, dtype=torch.int32).contiguous()
        if init.numel() != n:
            raise ValueError(f"initial_membership has {init.numel()} entries for {n} vertices")
    if objective.lower() not in ("modularity", "cpm"):
        raise ValueError(f"objective={objective!r}: 'modularity' or 'cpm'")
    if node_weights is not None:
        if objective.lower() != "cpm":
            raise NotImplementedError("node_weights with the modularity objective (its vertex weights are the strengths)")
        nw = node_weights.to(device=dev, dtype=torch.float32).contiguous()
        if nw.numel() != n:
            raise ValueError(f"node_weights has {nw.numel()} entries for {n} vertices")
        rc = lib.scamd_leiden_csr_nw_f32(ptr(indptr), ptr(indices), ptr(weights), n, nnz, float(resolution),
                                         int(n_iterations), float(beta), int(seed) & (2**64 - 1), 1, ptr(nw), ptr(init),
                                         ptr(memb), C.byref(q), C.byref(nc), ptr(ws), wsz, stream_ptr())
        _check(rc, "scamd_leiden_csr_nw_f32")
        return memb, float(q.value), int(nc.value)
    if objective.lower() == "cpm":
        rc = lib.scamd_leiden_csr_ex_f32(ptr(indptr), ptr(indices), ptr(weights), n, nnz, float(resolution),
                                         int(n_iterations), float(beta), int(seed) & (2**64 - 1), 1, ptr(init), ptr(memb),
                                         C.byref(q), C.byref(nc), ptr(ws), wsz, stream_ptr())
        _check(rc, "scamd_leiden_csr_ex_f32")
        return memb, float(q.value), int(nc.value)
    if init is not None:
        rc = lib.scamd_leiden_csr_init_f32(ptr(indptr), ptr(indices), ptr(weights), n, nnz, float(resolution),
                                           int(n_iterations), float(beta), int(seed) & (2**64 - 1), ptr(init), ptr(memb),
                                           C.byref(q), C.byref(nc), ptr(ws), wsz, stream_ptr())
        _check(rc, "scamd_leiden_csr_init_f32")
        return memb, float(q.value), int(nc.value)
    rc = lib.scamd_leiden_csr_f32(ptr(indptr), ptr(indices), ptr(weights), n, nnz, float(resolution),
                                  int(n_iterations), float(beta), int(seed) & (2**64 - 1), ptr(memb), C.byref(q),
                                  C.byref(nc), ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_leiden_csr_f32")
    return memb, float(q.value), int(nc.value)


def leiden_last_stats() -> dict:
    """Diagnostics of this thread's last `leiden` call (scamd_leiden_last_stats)."""
    out = (C.c_int32 * 16)()
    _lib.load().scamd_leiden_last_stats(out, 16)
    keys = ("iterations", "launches", "host_round_trips", "polish_full_sweeps", "polish_rounds", "polish_moves",
            "polish_skipped_proven", "levels_first_iteration", "lm_sweeps", "lm_sweep_algorithmic_MB", "polish_splits",
            "ended_by_iteration_cap", "polish_ended_by_round_cap", "iteration_cap", "device_copies", "device_fills")
    return dict(zip(keys, (int(v) for v in out)))


def modularity(indptr: torch.Tensor, indices: torch.Tensor, weights: torch.Tensor, n: int, membership: torch.Tensor,
               *, resolution: float = 1.0) -> float:
    dev = require_gpu()
    lib = _lib.load()
    indptr = indptr.to(torch.int64).contiguous()
    indices = indices.to(torch.int32).contiguous()
    weights = weights.to(torch.float32).contiguous()
    membership = membership.to(torch.int32).contiguous()
    nnz = weights.numel()
    ws, wsz = _ws(lib.scamd_leiden_workspace_bytes(n, nnz), dev)
    q = C.c_double(0.0)
    rc = lib.scamd_modularity_csr_f32(ptr(indptr), ptr(indices), ptr(weights), n, nnz, ptr(membership),
                                      float(resolution), C.byref(q), ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_modularity_csr_f32")
    return float(q.value)


# ---- upstream normalisation chain (csrc/preprocess.hip) ---------------------------------------------------------
def pp_row_sums(indptr, indices, data, n: int, col_skip: torch.Tensor | None = None) -> torch.Tensor:
    dev = require_gpu()
    out = _empty(n, dtype=torch.float32, device=dev)
    rc = _lib.load().scamd_pp_row_sums_f32(ptr(indptr), ptr(indices), ptr(data), n, data.numel(), ptr(col_skip), ptr(out), stream_ptr())
    _check(rc, "scamd_pp_row_sums_f32")
    return out


def pp_row_count_positive(indptr, data, n: int) -> torch.Tensor:
    dev = require_gpu()
    out = _empty(n, dtype=torch.int32, device=dev)
    _check(_lib.load().scamd_pp_row_count_positive_f32(ptr(indptr), ptr(data), n, data.numel(), ptr(out), stream_ptr()),
               "scamd_pp_row_count_positive_f32")
    return out


def pp_count_high(indptr, indices, data, n: int, g: int, row_total: torch.Tensor, max_fraction: float) -> torch.Tensor:
    dev = require_gpu()
    counts = _empty(g, dtype=torch.int32, device=dev)
    rc = _lib.load().scamd_pp_count_high_f32(ptr(indptr), ptr(indices), ptr(data), n, g, data.numel(), ptr(row_total),
                                             float(max_fraction), ptr(counts), stream_ptr())
    _check(rc, "scamd_pp_count_high_f32")
    return counts


def pp_row_divide_(indptr, data, n: int, factor: torch.Tensor) -> None:
    require_gpu()
    assert factor.dtype == torch.float32 and factor.numel() == n
    _check(_lib.load().scamd_pp_row_divide_f32(ptr(indptr), ptr(data), n, data.numel(), ptr(factor), stream_ptr()),
               "scamd_pp_row_divide_f32")


def pp_log1p_(data: torch.Tensor, base: float | None = None) -> None:
    require_gpu()
    assert data.dtype == torch.float32 and data.is_contiguous()
    _check(_lib.load().scamd_pp_log1p_f32(ptr(data), data.numel(), 0.0 if base is None else float(base), stream_ptr()),
               "scamd_pp_log1p_f32")


def pp_col_stats(indptr, indices, data, n: int, g: int, *, row_mask: torch.Tensor | None = None, expm1_scale: float | None = None,
                 count_positive: bool = True):
    """-> (sum float64 [g], sumsq float64 [g], npos int64 [g] or None) over the masked rows; expm1_scale = s: of
    expm1(x * s)."""
    dev = require_gpu()
    s = _empty(g, dtype=torch.float64, device=dev)
    sq = _empty(g, dtype=torch.float64, device=dev)
    npos = _empty(g, dtype=torch.int64, device=dev) if count_positive else None
    if row_mask is not None:
        assert row_mask.dtype == torch.uint8 and row_mask.numel() == n
    rc = _lib.load().scamd_pp_col_stats_f32(ptr(indptr), ptr(indices), ptr(data), n, g, data.numel(), ptr(row_mask),
                                            0 if expm1_scale is None else 1, 1.0 if expm1_scale is None else float(expm1_scale),
                                            ptr(s), ptr(sq), ptr(npos), stream_ptr())
    _check(rc, "scamd_pp_col_stats_f32")
    return s, sq, npos


def pp_col_stats_clip(indptr, indices, data, n: int, g: int, clip: torch.Tensor, *, row_mask: torch.Tensor | None = None):
    """-> (sum float64 [g], sumsq float64 [g]) of min(x, clip[gene]) over the stored values of the masked rows."""
    dev = require_gpu()
    clip = clip.to(torch.float64).contiguous()
    assert clip.numel() == g
    s = _empty(g, dtype=torch.float64, device=dev)
    sq = _empty(g, dtype=torch.float64, device=dev)
    if row_mask is not None:
        assert row_mask.dtype == torch.uint8 and row_mask.numel() == n
    rc = _lib.load().scamd_pp_col_stats_clip_f32(ptr(indptr), ptr(indices), ptr(data), n, g, data.numel(), ptr(row_mask),
                                                 ptr(clip), ptr(s), ptr(sq), stream_ptr())
    _check(rc, "scamd_pp_col_stats_clip_f32")
    return s, sq


def pp_scale_csr_(indptr, indices, data, n: int, std: torch.Tensor, *, max_value: float | None = None,
                  row_mask: torch.Tensor | None = None) -> None:
    require_gpu()
    assert std.dtype == torch.float64
    rc = _lib.load().scamd_pp_scale_csr_f32(ptr(indptr), ptr(indices), ptr(data), n, data.numel(), ptr(std),
                                            0.0 if max_value is None else float(max_value), 0 if max_value is None else 1,
                                            ptr(row_mask), stream_ptr())
    _check(rc, "scamd_pp_scale_csr_f32")


def pp_scale_dense(indptr, indices, data, n: int, g: int, mean: torch.Tensor, std: torch.Tensor, *,
                   max_value: float | None = None, row_mask: torch.Tensor | None = None,
                   out_dtype: torch.dtype = torch.float64) -> torch.Tensor:
    dev = require_gpu()
    assert mean.dtype == torch.float64 and std.dtype == torch.float64 and out_dtype in (torch.float32, torch.float64)
    out = _empty((n, g), dtype=out_dtype, device=dev)
    rc = _lib.load().scamd_pp_scale_dense_f32(ptr(indptr), ptr(indices), ptr(data), n, g, data.numel(), ptr(mean), ptr(std),
                                              0.0 if max_value is None else float(max_value), 0 if max_value is None else 1,
                                              ptr(row_mask), ptr(out), 1 if out_dtype == torch.float64 else 0, stream_ptr())
    _check(rc, "scamd_pp_scale_dense_f32")
    return out


# ---- UMAP layout (csrc/umap.hip) -----------------------------------------------------------------------------------
def umap_optimize_(indptr, indices, epochs_per_sample, n: int, y: torch.Tensor, *, n_epochs: int, a: float, b: float,
                   gamma: float = 1.0, initial_alpha: float = 1.0, negative_sample_rate: float = 5.0, seed: int = 0) -> None:
    """y [n, dim] float32 (device, contiguous): initial embedding in, optimised embedding out."""
    dev = require_gpu()
    lib = _lib.load()
    assert y.dtype == torch.float32 and y.is_contiguous() and y.shape[0] == n
    assert epochs_per_sample.dtype == torch.float32
    nnz, dim = int(indices.numel()), int(y.shape[1])
    need = lib.scamd_umap_workspace_bytes(n, nnz, dim)
    if need == 0:
        raise _lib.ScamdError(f"umap: unsupported shape n={n} n_components={dim} (supported: 1..8)")
    ws, wsz = _ws(need, dev)
    rc = lib.scamd_umap_optimize_f32(ptr(indptr), ptr(indices), ptr(epochs_per_sample), n, nnz, dim, int(n_epochs),
                                     float(a), float(b), float(gamma), float(initial_alpha), float(negative_sample_rate),
                                     int(seed) & (2**64 - 1), ptr(y), ptr(ws), wsz, stream_ptr())
    _check(rc, "scamd_umap_optimize_f32")
