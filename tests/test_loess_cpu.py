"""LOESS as R / scikit-misc compute it (scanpy_amd/preprocessing/_loess.py, oracle/loess.py) and
`highly_variable_genes(flavor='seurat_v3' | 'seurat_v3_paper')` on top of it.

Known answers: Seurat's vst table for pbmc3k (reference tests/_scripts/seurat_hvg_v3.csv.gz -> tests/golden/
loess_seurat_v3.npz): `variance.expected` is 10 ** the fitted values of R's loess(log10(variance) ~ log10(mean),
span = 0.3).  The reference's own test of the flavor (tests/test_highly_variable_genes.py:424-458) compares
scikit-misc's fit with these numbers at rtol 2e-5; here they are met at 1e-12.  The rest of the flavor (clipping,
ranks, batches) is checked against the numpy restatement in oracle/preprocess.py on the bundled counts -- the
reference's end-to-end goldens need pbmc3k, which is a download."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas as pd
import pytest
from scipy import sparse

import scanpy_amd as sc
from oracle import loess as oloess
from oracle import preprocess as op
from scanpy_amd.preprocessing import _csr_device
from scanpy_amd.preprocessing._loess import _kd_vertices, loess_fit
from tests.stub_backend import CpuStubPPBackend

GOLD = np.load(Path(__file__).parent / "golden" / "loess_seurat_v3.npz")


@pytest.fixture(autouse=True)
def _cpu_backend(monkeypatch):
    monkeypatch.setattr(_csr_device, "default_backend", lambda: CpuStubPPBackend())


def test_loess_reproduces_seurats_expected_variance():
    x, y = np.log10(GOLD["mean"]), np.log10(GOLD["variance"])
    fit = loess_fit(x, y, span=0.3, degree=2)
    assert np.abs(10 ** fit / GOLD["variance_expected"] - 1).max() < 1e-12
    # what a plain pointwise local regression would give instead: two orders of magnitude outside the reference's
    # own tolerance -- the k-d tree + Hermite blending is part of the definition
    verts = _kd_vertices(np.sort(x), int(np.floor(x.size * 0.3 * 0.2)))
    assert verts.size == 30 and np.isin(verts[1:-1], x).all()  # vertices sit AT data values


def test_oracle_loess_agrees_on_a_subsample():
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(GOLD["mean"].size, 3000, replace=False))
    x, y = np.log10(GOLD["mean"][pick]), np.log10(GOLD["variance"][pick])
    np.testing.assert_allclose(loess_fit(x, y), oloess.loess(x, y), rtol=0, atol=1e-9)
    np.testing.assert_allclose(loess_fit(x, y, span=0.5, degree=1), oloess.loess(x, y, span=0.5, degree=1), rtol=0,
                               atol=1e-9)
    with pytest.raises(ValueError, match="too few"):
        loess_fit(x[:5], y[:5], span=0.3)
    with pytest.raises(ValueError, match="one length"):
        loess_fit(x, y[:-1])


def _counts(pbmc68k):
    raw = pbmc68k["raw_X"]
    return sparse.csr_matrix(np.rint(raw.toarray() if sparse.issparse(raw) else raw).astype(np.float32))


@pytest.mark.parametrize("fmt", ["csr", "dense"])
def test_seurat_v3_single_batch_equals_the_restatement(pbmc68k, fmt):
    x = _counts(pbmc68k)
    a = sc.AnnData(x if fmt == "csr" else x.toarray())
    sc.pp.highly_variable_genes(a, flavor="seurat_v3", n_top_genes=200)
    ref = op.highly_variable_genes_seurat_v3(x, n_top_genes=200)
    assert a.uns["hvg"] == {"flavor": "seurat_v3"}
    assert list(a.var.columns) == ["highly_variable", "highly_variable_rank", "means", "variances", "variances_norm"]
    np.testing.assert_allclose(a.var["means"], ref["means"], rtol=1e-6)
    np.testing.assert_allclose(a.var["variances"], ref["variances"], rtol=1e-5)
    np.testing.assert_allclose(a.var["variances_norm"], ref["variances_norm"], rtol=1e-5, equal_nan=True)
    assert int(a.var["highly_variable"].sum()) == 200
    assert (a.var["highly_variable"].to_numpy() != ref["highly_variable"].to_numpy()).sum() <= 2  # ties at the cut
    ranks = a.var["highly_variable_rank"].to_numpy()
    assert np.isnan(ranks[~a.var["highly_variable"].to_numpy()]).all() and np.nanmax(ranks) == 199
    assert a.var["variances_norm"].dtype == np.float64


@pytest.mark.parametrize("flavor", ["seurat_v3", "seurat_v3_paper"])
def test_seurat_v3_batches_and_return_frame(pbmc68k, flavor):
    x = _counts(pbmc68k)
    batch = np.array(["b%d" % (i % 3) for i in range(x.shape[0])])
    a = sc.AnnData(x)
    a.obs["batch"] = batch
    df = sc.pp.highly_variable_genes(a, flavor=flavor, n_top_genes=150, batch_key="batch", inplace=False)
    ref = op.highly_variable_genes_seurat_v3(x, n_top_genes=150, batch=batch, flavor=flavor)
    np.testing.assert_allclose(df["variances_norm"], ref["variances_norm"], rtol=1e-5, equal_nan=True)
    np.testing.assert_array_equal(df["highly_variable_nbatches"].to_numpy(), ref["highly_variable_nbatches"].to_numpy())
    assert (df["highly_variable"].to_numpy() != ref["highly_variable"].to_numpy()).sum() <= 2
    assert "gene_name" in df.columns and "highly_variable_nbatches" in df.columns
    single = sc.pp.highly_variable_genes(a, flavor=flavor, n_top_genes=150, inplace=False)
    assert "highly_variable_nbatches" not in single.columns  # dropped without a batch key (`:305-306`)
    sub = sc.AnnData(x)
    sc.pp.highly_variable_genes(sub, flavor=flavor, n_top_genes=150, subset=True)
    assert sub.shape == (x.shape[0], 150)


def test_seurat_v3_warns_on_non_counts_and_handles_constant_genes(pbmc68k):
    """tests/test_highly_variable_genes.py:494-512"""
    x = _counts(pbmc68k)
    a = sc.AnnData((x * 0.5).tocsr().astype(np.float32))
    with pytest.warns(UserWarning, match="expects raw count data, but non-integers were found"):
        sc.pp.highly_variable_genes(a, flavor="seurat_v3", n_top_genes=50)
    d = x.toarray()
    d[:, :5] = 0  # constant (all-zero) genes: variance 0, excluded from the trend, never selected
    b = sc.AnnData(sparse.csr_matrix(d))
    sc.pp.highly_variable_genes(b, flavor="seurat_v3", n_top_genes=50, check_values=False)
    assert not b.var["highly_variable"].iloc[:5].any() and int(b.var["highly_variable"].sum()) == 50


def test_clip_col_sums_stub_and_value_check_tensor_ops():
    """The numpy stand-in of `clip_col_sums` against the reference formula (`clip_square_sum`,
    _highly_variable_genes.py:75-115; the product's kernel `scamd_pp_col_stats_clip_f32` is compared with the same
    formula in tests/test_gpu_preprocess.py), and `GpuPPBackend.nonnegative_integers` -- plain tensor ops -- run on CPU
    tensors."""
    import torch

    rng = np.random.default_rng(1)
    x = sparse.random(300, 40, density=0.3, format="csr", random_state=1, dtype=np.float32)
    x.data = np.rint(x.data * 20).astype(np.float32)
    x.eliminate_zeros()
    stub = CpuStubPPBackend()
    ms = stub.upload(x)
    dm = _csr_device.DeviceMatrix("csr", x.shape, torch.from_numpy(x.indptr.astype(np.int64)),
                                  torch.from_numpy(x.indices.astype(np.int32)), torch.from_numpy(x.data.copy()), x)
    clip = rng.random(40) * 10
    mask = rng.random(300) < 0.5
    for rm in (None, mask):
        d = x.toarray().astype(np.float64)
        if rm is not None:
            d = d[rm]
        c = np.minimum(d, clip[None, :]) * (d != 0)
        want_sq, want_s = (c * c).sum(axis=0), c.sum(axis=0)
        got = stub.clip_col_sums(ms, clip, row_mask=rm)
        np.testing.assert_allclose(got[0], want_sq, rtol=1e-12)
        np.testing.assert_allclose(got[1], want_s, rtol=1e-12)
    assert _csr_device.GpuPPBackend.nonnegative_integers(None, dm) is True
    dm.data[3] = 0.5
    assert _csr_device.GpuPPBackend.nonnegative_integers(None, dm) is False
    dm.data[3] = -0.0
    assert _csr_device.GpuPPBackend.nonnegative_integers(None, dm) is False  # signbit, like the reference


def test_seurat_v3_streams_a_backed_count_matrix(pbmc68k, tmp_path, monkeypatch):
    """all three sweeps of the flavor are sums over cells, so an on-disk count matrix is streamed by row chunks"""
    from scanpy_amd.preprocessing import _highly_variable_genes as hvg

    x = _counts(pbmc68k)
    a = sc.AnnData(x)
    a.obs["batch"] = pd.Categorical(np.arange(x.shape[0]) % 2)
    sc.write_h5ad(tmp_path / "c.h5ad", a)
    b = sc.read_h5ad(tmp_path / "c.h5ad", backed="r")
    orig = hvg._StreamedColStats.__init__
    monkeypatch.setattr(hvg._StreamedColStats, "__init__", lambda self, be, x, step=111: orig(self, be, x, step))
    for kw in (dict(), dict(batch_key="batch")):
        ra = sc.pp.highly_variable_genes(a, flavor="seurat_v3", n_top_genes=120, inplace=False, **kw)
        rb = sc.pp.highly_variable_genes(b, flavor="seurat_v3", n_top_genes=120, inplace=False, **kw)
        np.testing.assert_allclose(rb["variances_norm"], ra["variances_norm"], rtol=1e-9, equal_nan=True)
        assert (rb["highly_variable"].to_numpy() != ra["highly_variable"].to_numpy()).sum() <= 2
    assert b.X.is_backed
