"""-m gpu: scanpy_amd.pp.normalize_total / log1p / highly_variable_genes / scale through the HIP kernels of
csrc/preprocess.hip (C ABI), against the reference's goldens and the oracle.  Same assertions as the CPU host-logic
tests (tests/test_preprocess_host_cpu.py), which run them on a numpy stand-in for the device passes."""
from __future__ import annotations

import numpy as np
import pytest
from scipy import sparse

from oracle import preprocess as op
from tests import test_preprocess_host_cpu as host

pytestmark = pytest.mark.gpu

TYPES = host.TYPES


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_chain_goldens(pbmc68k, hvg_golden, typ):
    host.check_chain_against_goldens(pbmc68k, hvg_golden, typ)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_normalize_total(typ):
    host.check_normalize_total(typ)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_scale(scale_toy, typ):
    host.check_scale(scale_toy, typ)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_random_against_oracle(typ):
    host.check_random_against_oracle(typ)


def _counts(n, g, seed, density=0.05):
    rng = np.random.default_rng(seed)
    x = sparse.random(n, g, density=density, format="csr", dtype=np.float32, random_state=seed)
    x.data = np.ceil(rng.lognormal(0.5, 1.0, size=x.nnz)).astype(np.float32)
    return x


@pytest.mark.parametrize(("n", "g"), [(20000, 2000), (3000, 5000)], ids=["lds-tables", "global-atomics"])
def test_kernels_at_size_against_oracle(n, g):
    """raw kernels (both column-table variants: g <= 4096 in LDS, beyond in global memory), ragged rows, empty rows"""
    import torch

    from scanpy_amd import _kernels as K

    x = _counts(n, g, seed=n)
    x.data[x.indptr[17]:x.indptr[18]] = 0  # a cell whose stored values are all zero
    x = sparse.vstack([x[:100], sparse.csr_matrix((3, g), dtype=np.float32), x[100:]]).tocsr()  # three empty rows
    n = x.shape[0]
    dev = torch.device("cuda")
    ip = torch.from_numpy(x.indptr.astype(np.int64)).to(dev)
    ix = torch.from_numpy(x.indices.astype(np.int32)).to(dev)
    dt = torch.from_numpy(x.data.copy()).to(dev)
    sums = K.pp_row_sums(ip, ix, dt, n).cpu().numpy()
    ref = np.asarray(x.astype(np.float64).sum(axis=1)).ravel().astype(np.float32)
    assert np.array_equal(sums, ref)  # float64 accumulation of integers-valued float32: exact
    hi = K.pp_count_high(ip, ix, dt, n, g, torch.from_numpy(sums).to(dev), 0.05).cpu().numpy()
    rows = np.repeat(np.arange(n), np.diff(x.indptr))
    assert np.array_equal(hi, np.bincount(x.indices[x.data > np.float32(0.05) * sums[rows]], minlength=g))
    sums2 = K.pp_row_sums(ip, ix, dt, n, torch.from_numpy(hi.astype(np.int32)).to(dev)).cpu().numpy()
    keep = hi[x.indices] == 0
    assert np.array_equal(sums2, np.bincount(rows[keep], weights=x.data[keep].astype(np.float64), minlength=n).astype(np.float32))
    xo, fo, _ = op.normalize_total(x, target_sum=1e4)
    K.pp_row_divide_(ip, dt, n, torch.from_numpy((sums / np.float32(1e4)).astype(np.float32)).to(dev))
    np.testing.assert_allclose(dt.cpu().numpy(), xo.data, rtol=1e-6)
    K.pp_log1p_(dt)
    xo = op.log1p(xo)
    np.testing.assert_allclose(dt.cpu().numpy(), xo.data, rtol=2e-6, atol=1e-7)
    # from here on the host reference starts from the device's own values (log1pf and numpy's log1p differ in the last
    # ulp of float32; the statistics below are checked to float64 accuracy)
    xo = sparse.csr_matrix((dt.cpu().numpy(), x.indices, x.indptr), shape=x.shape)
    mask = (np.arange(n) % 3 != 0)
    for row_mask in (None, mask):
        sub = xo if row_mask is None else xo[row_mask]
        tm = None if row_mask is None else torch.from_numpy(row_mask.astype(np.uint8)).to(dev)
        s, sq, npos = (t.cpu().numpy() for t in K.pp_col_stats(ip, ix, dt, n, g, row_mask=tm))
        s64 = sub.astype(np.float64)
        np.testing.assert_allclose(s, np.asarray(s64.sum(axis=0)).ravel(), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(sq, np.asarray(s64.multiply(s64).sum(axis=0)).ravel(), rtol=1e-12, atol=1e-12)
        assert np.array_equal(npos, np.asarray((sub > 0).sum(axis=0)).ravel())
        s, sq, _ = (t.cpu().numpy() for t in K.pp_col_stats(ip, ix, dt, n, g, row_mask=tm, expm1_scale=1.0))
        e = sub.copy()
        e.data = np.expm1(e.data)
        e64 = e.astype(np.float64)
        np.testing.assert_allclose(s, np.asarray(e64.sum(axis=0)).ravel(), rtol=5e-6)
        np.testing.assert_allclose(sq, np.asarray(e64.multiply(e64).sum(axis=0)).ravel(), rtol=1e-5)
    # scale, both modes, masked, clipped
    mean, var = op.mean_var(xo[mask])
    std = np.sqrt(var)
    std[std == 0] = 1
    tm = torch.from_numpy(mask.astype(np.uint8)).to(dev)
    tmean, tstd = torch.from_numpy(mean).to(dev), torch.from_numpy(std).to(dev)
    dense = K.pp_scale_dense(ip, ix, dt, n, g, tmean, tstd, max_value=4.0, row_mask=tm).cpu().numpy()
    ref, _, _ = op.scale(xo, zero_center=True, max_value=4.0, mask_obs=mask)
    assert dense.dtype == np.float64
    np.testing.assert_allclose(dense, ref, rtol=1e-12, atol=1e-12)
    d2 = dt.clone()
    K.pp_scale_csr_(ip, ix, d2, n, tstd, max_value=1.5, row_mask=tm)
    ref, _, _ = op.scale(xo, zero_center=False, max_value=1.5, mask_obs=mask)
    np.testing.assert_allclose(d2.cpu().numpy(), ref.data, rtol=1e-6)


def test_empty_matrix_and_bad_arguments():
    import torch

    from scanpy_amd import _kernels as K
    from scanpy_amd import _lib

    dev = torch.device("cuda")
    ip = torch.zeros(1, dtype=torch.int64, device=dev)
    e32 = torch.zeros(0, dtype=torch.float32, device=dev)
    ei = torch.zeros(0, dtype=torch.int32, device=dev)
    assert K.pp_row_sums(ip, ei, e32, 0).numel() == 0
    s, sq, npos = K.pp_col_stats(ip, ei, e32, 0, 5)
    assert float(s.abs().sum()) == 0 and int(npos.sum()) == 0
    with pytest.raises(_lib.ScamdError, match="base"):
        K.pp_log1p_(torch.ones(4, dtype=torch.float32, device=dev), base=1.0)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_filters(typ):
    host.check_filters(typ)


@pytest.mark.parametrize("typ", TYPES, ids=lambda t: t.__name__)
def test_rep_mutation(typ):
    host.check_rep_mutation(typ)


def test_hvg_reference_semantics():
    host.check_hvg_reference_semantics()


def test_hvg_keeps_the_matrix(pbmc68k):
    host.check_hvg_keeps_the_matrix(pbmc68k)


def test_col_stats_clip_kernel_matches_numpy():
    """`scamd_pp_col_stats_clip_f32` (the `clip_square_sum` sweep of flavor='seurat_v3'): per-gene sums of min(x, clip)"""
    import torch
    from scipy import sparse

    from scanpy_amd import _kernels as K

    rng = np.random.default_rng(4)
    for n, g in ((3000, 500), (700, 5000)):  # LDS table / global-atomic variant (g > 4096)
        x = sparse.random(n, g, density=0.05, random_state=7, format="csr", dtype=np.float32)
        x.data = np.rint(x.data * 40).astype(np.float32)
        clip = rng.uniform(2.0, 30.0, size=g)
        mask = (rng.random(n) < 0.6).astype(np.uint8)
        ip = torch.from_numpy(x.indptr.astype(np.int64)).cuda()
        ix = torch.from_numpy(x.indices.astype(np.int32)).cuda()
        dv = torch.from_numpy(x.data).cuda()
        for m in (None, mask):
            s, sq = K.pp_col_stats_clip(ip, ix, dv, n, g, torch.from_numpy(clip).cuda(),
                                        row_mask=None if m is None else torch.from_numpy(m).cuda())
            xm = x if m is None else x[m.astype(bool)]
            c = xm.tocsc()
            v = np.minimum(c.data.astype(np.float64), np.repeat(clip, np.diff(c.indptr)))
            cols = np.repeat(np.arange(g), np.diff(c.indptr))
            np.testing.assert_allclose(s.cpu().numpy(), np.bincount(cols, weights=v, minlength=g), rtol=1e-12, atol=1e-9)
            np.testing.assert_allclose(sq.cpu().numpy(), np.bincount(cols, weights=v * v, minlength=g), rtol=1e-12, atol=1e-9)
