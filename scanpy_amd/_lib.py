"""ctypes binding of libscanpy_amd.so (the C ABI declared in include/scanpy_amd.h).

The product path has NO CPU fallback: if the shared library is missing or no GPU is visible the
calls raise.  `load()` only dlopens (works on a CPU-only host, used by the symbol-export test).
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "_lib" / "libscanpy_amd.so"

_i64, _i32, _f64, _sz, _vp, _u64 = C.c_int64, C.c_int, C.c_double, C.c_size_t, C.c_void_p, C.c_uint64

# name -> (restype, argtypes); must list every symbol of include/scanpy_amd.h
SIGNATURES = {
    "scamd_abi_version": (_i32, []),
    "scamd_last_error": (C.c_char_p, []),
    "scamd_device_count": (_i32, []),
    "scamd_knn_workspace_bytes": (_sz, [_i64, _i32, _i64, _i32]),
    "scamd_knn_l2_f32": (_i32, [_vp, _i64, _i32, _i64, _i64, _i64, _i32, _vp, _vp, _f64, C.POINTER(_i64), _vp, _sz, _vp]),
    "scamd_knn_l2_ivf_f32": (_i32, [_vp, _i64, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _vp, C.POINTER(_i64), _vp, _sz, _vp]),
    "scamd_knn_last_select_ms": (C.c_float, []),
    "scamd_knn_last_select_pairs": (_f64, []),
    "scamd_knn_last_select_prepass_pairs": (_f64, []),
    "scamd_knn_last_select_engine": (_i32, []),
    "scamd_knn_last_second_tier_queries": (_i32, []),
    "scamd_knn_last_nprobe": (_i32, []),
    "scamd_knn_last_coarse": (_i32, []),
    "scamd_knn_cert_factors": (None, [_i32, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64)]),
    "scamd_knn_debug_b3_scores_workspace_bytes": (_sz, [_i64]),
    "scamd_knn_debug_b3_scores_f32": (_i32, [_vp, _i64, _i32, _i64, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scamd_fuzzy_workspace_bytes": (_sz, [_i64, _i32]),
    "scamd_fuzzy_simplicial_set_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp, C.POINTER(_i64), _vp, _sz, _vp]),
    "scamd_fuzzy_weights_f32": (_i32, [_vp, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "scamd_fuzzy_merge_workspace_bytes": (_sz, [_i64, _i64]),
    "scamd_fuzzy_merge_rows_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(_i64), _vp, _sz, _vp]),
    "scamd_gauss_connectivities_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, C.POINTER(_i64), _vp, _sz, _vp]),
    "scamd_jaccard_connectivities_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _i64, C.POINTER(_i64), _vp, _sz, _vp]),
    "scamd_csr_row_stats_f32": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "scamd_csr_transpose_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "scamd_csr_transpose_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scamd_spmm_csr_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp]),
    "scamd_spmm_f64acc_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "scamd_spmm_csr_f32_f64acc": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scamd_csr_gram_workspace_bytes": (_sz, [_i64, _i64]),
    "scamd_csr_gram_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _vp, C.POINTER(C.c_float), _vp, _sz, _vp]),
    "scamd_eigh_topk_workspace_bytes": (_sz, [_i64, _i32]),
    "scamd_eigh_topk_f64": (_i32, [_vp, _i64, _i64, _i32, _u64, _f64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scamd_dense_debug_f64": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, C.POINTER(_i32), _vp]),
    "scamd_pca_csr_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "scamd_pca_csr_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _u64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scamd_pca_solve_gram_workspace_bytes": (_sz, [_i64, _i32]),
    "scamd_pca_solve_gram_f64": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _u64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scamd_spectral_embedding_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "scamd_spectral_embedding_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _u64, _f64, _i32, _i32, _vp, C.POINTER(_f64), _vp, _sz, _vp]),
    "scamd_colsum_workspace_bytes": (_sz, [_i32]),
    "scamd_colsum_f32_f64": (_i32, [_vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "scamd_leiden_workspace_bytes": (_sz, [_i64, _i64]),
    "scamd_leiden_csr_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _f64, _u64, _vp, C.POINTER(_f64), C.POINTER(_i32), _vp, _sz, _vp]),
    "scamd_leiden_csr_init_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _f64, _u64, _vp, _vp, C.POINTER(_f64), C.POINTER(_i32), _vp, _sz, _vp]),
    "scamd_leiden_csr_ex_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _f64, _u64, _i32, _vp, _vp, C.POINTER(_f64), C.POINTER(_i32), _vp, _sz, _vp]),
    "scamd_leiden_csr_nw_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _f64, _u64, _i32, _vp, _vp, _vp, C.POINTER(_f64), C.POINTER(_i32), _vp, _sz, _vp]),
    "scamd_leiden_last_stats": (None, [C.POINTER(_i32), _i32]),
    "scamd_leiden_debug_split_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, C.POINTER(_i32), _vp, _sz, _vp]),
    "scamd_modularity_csr_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _f64, C.POINTER(_f64), _vp, _sz, _vp]),
    "scamd_pp_row_sums_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "scamd_pp_row_count_positive_f32": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "scamd_pp_count_high_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, C.c_float, _vp, _vp]),
    "scamd_pp_row_divide_f32": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "scamd_pp_log1p_f32": (_i32, [_vp, _i64, _f64, _vp]),
    "scamd_pp_col_stats_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i32, _f64, _vp, _vp, _vp, _vp]),
    "scamd_pp_col_stats_clip_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "scamd_pp_scale_csr_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _f64, _i32, _vp, _vp]),
    "scamd_pp_scale_dense_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _f64, _i32, _vp, _vp, _i32, _vp]),
    "scamd_umap_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "scamd_umap_optimize_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _f64, _f64, _f64, _f64, _f64, _u64, _vp, _vp, _sz, _vp]),
    "scamd_lzf_decompress": (_i64, [_vp, _sz, _vp, _sz]),
    "scamd_unshuffle": (_i32, [_vp, _vp, _sz, _i32]),
    "scamd_selftest_mfma_layout": (_i32, [_vp]),
}


class ScamdError(RuntimeError):
    pass


@lru_cache(maxsize=1)
def load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ScamdError(
            f"{LIB_PATH} not found: build it with `python -m scanpy_amd._build` "
            "(scanpy_amd has no CPU fallback for the pca/neighbors/leiden kernels)"
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.scamd_abi_version() != 1:
        raise ScamdError("libscanpy_amd.so ABI version mismatch; rebuild")
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().scamd_last_error().decode(errors="replace")
        raise ScamdError(f"{what or 'libscanpy_amd'} failed (code {rc}): {msg}")
