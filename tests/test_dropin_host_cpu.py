"""The drop-in front-ends (sc.pp.pca / sc.pp.neighbors / sc.tl.leiden: signatures, slots, params, errors, key_added,
restrict_to, transformer routes) on a machine WITHOUT a GPU: the kernel layer is replaced by the CPU stand-ins the
gloo tests use (tests/dist_worker.py: sklearn brute kNN, oracle fuzzy set, oracle Leiden; tests/stub_backend.py for the
PCA data passes), so what runs here is the product's host logic, including the real PCA solver.  The same test bodies
run against the HIP kernels in tests/test_gpu_pipeline.py."""
from __future__ import annotations

import numpy as np
import pytest
import torch

import test_gpu_pipeline as gp
from dist_worker import _patch_kernels
from stub_backend import CpuStubBackend


@pytest.fixture(autouse=True)
def _cpu_kernel_layer(monkeypatch):
    from scanpy_amd import _device, _kernels
    from scanpy_amd.preprocessing import _pca_solver

    saved = {name: getattr(_kernels, name) for name in ("knn", "fuzzy_simplicial_set", "leiden", "modularity")}
    monkeypatch.setattr(_device, "require_gpu", lambda: torch.device("cpu"))
    monkeypatch.setattr(_pca_solver, "GpuBackend", CpuStubBackend)
    _patch_kernels()

    def modularity(indptr, indices, weights, n, membership, *, resolution=1.0):
        from scipy import sparse

        from oracle import leiden as ol

        adj = sparse.csr_matrix((weights.numpy(), indices.numpy(), indptr.numpy()), shape=(n, n))
        return ol.modularity(adj, membership.numpy(), resolution=resolution)

    _kernels.modularity = modularity
    yield
    for name, fn in saved.items():
        setattr(_kernels, name, fn)


@pytest.fixture(scope="module")
def sc():
    import scanpy_amd

    return scanpy_amd


@pytest.mark.parametrize("fmt", ["csr", "dense"])
def test_pca_transform_golden(sc, pca_toy, fmt):
    gp.test_pca_transform_golden(sc, pca_toy, fmt)


def test_pca_no_zero_center_golden(sc, pca_toy):
    gp.test_pca_no_zero_center_golden(sc, pca_toy)


def test_pca_randomized_sparse_warns(sc, pca_toy):
    gp.test_pca_randomized_sparse_warns(sc, pca_toy)


def test_pca_shapes_and_errors(sc, pca_toy):
    gp.test_pca_shapes_and_errors(sc, pca_toy)


def test_pca_real_counts_layer_and_mask(sc, pbmc68k):
    gp.test_pca_real_counts_layer_and_mask(sc, pbmc68k)


def test_pca_key_added_and_copy(sc, pca_toy):
    gp.test_pca_key_added_and_copy(sc, pca_toy)


def test_neighbors_toy_golden(sc, neighbors_toy):
    gp.test_neighbors_toy_golden(sc, neighbors_toy)


def test_neighbors_key_added_use_rep_n_pcs(sc, pbmc68k):
    gp.test_neighbors_key_added_use_rep_n_pcs(sc, pbmc68k)


def test_neighbors_transformer_plugin_route(sc, pbmc68k):
    gp.test_neighbors_transformer_plugin_route(sc, pbmc68k)


def test_neighbors_precomputed_distances(sc, pbmc68k):
    gp.test_neighbors_precomputed_distances(sc, pbmc68k)


def test_neighbors_auto_pca_fallback(sc, pbmc68k):
    gp.test_neighbors_auto_pca_fallback(sc, pbmc68k)


def test_leiden_basic_and_params(sc, pbmc68k):
    gp.test_leiden_basic_and_params(sc, pbmc68k)


def test_leiden_errors(sc, pbmc68k):
    gp.test_leiden_errors(sc, pbmc68k)


def test_leiden_restrict_to_and_keys(sc, pbmc68k):
    gp.test_leiden_restrict_to_and_keys(sc, pbmc68k)


@pytest.mark.parametrize("chunk_size", [333, 2000])
def test_pca_chunked_equals_one_shot(sc, pbmc68k, chunk_size, monkeypatch):
    gp.test_pca_chunked_equals_one_shot(sc, pbmc68k, chunk_size, "0", monkeypatch)
