"""GPU parity tests of the individual C-ABI kernels against the CPU oracle (run with -m gpu)."""
from __future__ import annotations

import numpy as np
import pytest
from scipy import sparse

from helpers import knn_sets_equal_mod_ties
from oracle import connectivities as oc
from oracle import knn as oknn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import torch

    from scanpy_amd import _kernels

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _kernels


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_mfma_layout(K):
    K.mfma_selftest()


def _check_knn(K, x, k, **kw):
    idx, dist, nfb = K.knn(_dev(x), k, **kw)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    n = x.shape[0]
    oi, od, _ = oknn.knn_sklearn(x, k)
    # column 0: self with exact zero
    assert (idx[:, 0] == np.arange(n)).all()
    assert (dist[:, 0] == 0).all()
    # exact float64 truth for the distances of whatever we returned
    x64 = x.astype(np.float64)
    true_d = np.sqrt(((x64[:, None, :] - x64[idx]) ** 2).sum(-1))
    np.testing.assert_allclose(dist, true_d, rtol=1e-12, atol=0)
    assert (np.diff(dist[:, 1:], axis=1) >= 0).all(), "neighbours must be sorted by distance"
    bad, differ = knn_sets_equal_mod_ties(idx[:, 1:], dist[:, 1:], oi[:, 1:], od[:, 1:])
    assert bad == 0, f"{bad} rows differ from the sklearn oracle beyond ties ({differ} incl. ties)"
    np.testing.assert_allclose(dist[:, 1:], od[:, 1:], rtol=2e-6, atol=2e-6 * np.abs(x).max())
    return idx, dist, nfb


def test_knn_toy_golden(K, neighbors_toy):
    x = neighbors_toy["X"].astype(np.float32)
    idx, dist, _ = K.knn(_dev(x), 3)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    d = oknn.sparse_from_indices_distances(idx, dist, keep_self=False)
    np.testing.assert_allclose(d.toarray(), neighbors_toy["distances_euclidean"], rtol=1e-6)


@pytest.mark.parametrize(
    ("n", "d", "k"),
    [(1000, 50, 15), (5000, 10, 30), (3000, 64, 15), (2000, 100, 15), (777, 3, 5), (4000, 50, 100), (300, 128, 15), (2500, 33, 24),
     (2000, 20, 15), (1500, 30, 10), (60000, 50, 15), (130, 50, 15), (33, 8, 4),
     # round 5: d in (128, 256], k in (120, 256] -- the reference takes any (src/scanpy/neighbors/__init__.py:88-103)
     (3000, 150, 15), (2500, 256, 30), (3000, 50, 200), (2000, 200, 256), (1500, 129, 121), (5000, 20, 130)],
)
def test_knn_vs_sklearn(K, n, d, k):
    rng = np.random.default_rng(n + d + k)
    centers = rng.standard_normal((8, d)).astype(np.float32) * 3
    x = (centers[rng.integers(0, 8, n)] + rng.standard_normal((n, d))).astype(np.float32)
    _check_knn(K, x, k)


def test_knn_large_offset_norms(K):
    """Points far from the origin: float32 expansion loses digits, certificate/fallback must cope."""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((3000, 50)) + 300.0).astype(np.float32)
    _, _, nfb = _check_knn(K, x, 15)
    print("fallback queries:", nfb)


def test_knn_duplicates(K):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((500, 20)).astype(np.float32)
    x = np.vstack([x, x[:100], x[:50]])  # exact duplicates (and triplicates)
    idx, dist, _ = K.knn(_dev(x), 10)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    x64 = x.astype(np.float64)
    true_d = np.sqrt(((x64[:, None, :] - x64[idx]) ** 2).sum(-1))
    np.testing.assert_allclose(dist, true_d, rtol=1e-12)
    ei, ed = oknn.knn_exact_f64(x, np.arange(len(x)), 11)
    # drop self from the exact list (self may not be first among duplicates)
    for r in range(len(x)):
        keep = ei[r] != r
        er = ed[r][keep][:9] if keep.sum() >= 9 else ed[r][1:10]
        np.testing.assert_allclose(dist[r, 1:], er, rtol=1e-12, atol=1e-12)


def test_knn_forced_fallback_matches(K):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((1500, 50)).astype(np.float32)
    i1, d1, n1 = K.knn(_dev(x), 15)
    i2, d2, n2 = K.knn(_dev(x), 15, cert_scale=1e30)
    assert n2 == 1500 and n1 < 1500
    np.testing.assert_array_equal(i1.cpu().numpy(), i2.cpu().numpy())
    np.testing.assert_array_equal(d1.cpu().numpy(), d2.cpu().numpy())


def test_knn_query_shard(K):
    rng = np.random.default_rng(13)
    x = rng.standard_normal((2100, 50)).astype(np.float32)
    i_all, d_all, _ = K.knn(_dev(x), 15)
    i_s, d_s, _ = K.knn(_dev(x), 15, q_begin=700, n_query=900)
    np.testing.assert_array_equal(i_all[700:1600].cpu().numpy(), i_s.cpu().numpy())
    np.testing.assert_array_equal(d_all[700:1600].cpu().numpy(), d_s.cpu().numpy())


def test_knn_full_size_sampled_exact(K):
    """BASELINE size (1M x 50, k = 15): the GPU result of 1500 sampled queries equals a float64 brute force over all
    1M rows computed on the host; every row is its own first neighbour; distances ascend."""
    import torch

    from scanpy_amd.datasets import blobs_embedding

    n, k = 1_000_000, 15
    x, _ = blobs_embedding(n, 50, seed=3)
    idx, dist, nfb = K.knn(_dev(x), k)
    assert nfb < n // 1000
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert (idx[:, 0] == np.arange(n)).all() and (dist[:, 0] == 0).all()
    assert (np.diff(dist, axis=1) >= 0).all()
    rng = np.random.default_rng(0)
    qs = np.sort(rng.choice(n, 1500, replace=False))
    x64 = x.astype(np.float64)
    xn = (x64 * x64).sum(1)
    for s in range(0, len(qs), 250):
        q = qs[s : s + 250]
        d2 = xn[q][:, None] + xn[None, :] - 2.0 * (x64[q] @ x64.T)
        d2[np.arange(len(q)), q] = -1.0  # self first
        part = np.argpartition(d2, k, axis=1)[:, : k + 4]
        exact = ((x64[q][:, None, :] - x64[part]) ** 2).sum(-1)
        exact[part == q[:, None]] = -1.0
        order = np.lexsort((part, exact), axis=1)[:, :k]
        ref_idx = np.take_along_axis(part, order, axis=1)
        ref_d = np.sqrt(np.maximum(np.take_along_axis(exact, order, axis=1), 0.0))
        same = (np.sort(idx[q], axis=1) == np.sort(ref_idx, axis=1)).all(axis=1)
        assert same.mean() > 0.999, same.mean()  # (the rest: ties at the k-th distance)
        np.testing.assert_allclose(dist[q][same][:, 1:], ref_d[same][:, 1:], rtol=1e-9)


def test_knn_tiny(K):
    x = np.array([[0.0, 0.0], [1.0, 0.0]], dtype=np.float32)
    idx, dist, _ = K.knn(_dev(x), 3)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert idx[0].tolist() == [0, 1, -1] and idx[1].tolist() == [1, 0, -1]
    assert dist[0, 1] == 1.0 and np.isinf(dist[0, 2])


# ---------------------------------------------------------------------------------------------------
def _fuzzy_vs_oracle(K, idx, dist, k, atol=2e-6):
    n = idx.shape[0]
    indptr, indices, data, sigma, rho = K.fuzzy_simplicial_set(_dev(idx.astype(np.int32)), _dev(dist.astype(np.float32)))
    got = sparse.csr_matrix((data.cpu().numpy(), indices.cpu().numpy(), indptr.cpu().numpy()), shape=(n, n))
    ref, rs, rr = oc.fuzzy_simplicial_set(idx, dist, n, k)
    assert got.has_canonical_format or (got.sort_indices() is None)
    np.testing.assert_array_equal(rho.cpu().numpy(), rr)
    np.testing.assert_allclose(sigma.cpu().numpy(), rs, rtol=1e-6)
    assert (np.diff(got.indices.astype(np.int64))[np.setdiff1d(np.arange(got.nnz - 1), got.indptr[1:-1] - 1)] > 0).all(), "sorted columns"
    assert got.nnz == ref.nnz, (got.nnz, ref.nnz)
    np.testing.assert_array_equal(got.indptr, ref.indptr)
    np.testing.assert_array_equal(got.indices, ref.indices)
    np.testing.assert_allclose(got.data, ref.data, rtol=0, atol=atol)
    assert abs(got - got.T).max() == 0, "connectivities must be exactly symmetric"
    return got


def test_fuzzy_toy_golden(K, neighbors_toy):
    x = neighbors_toy["X"]
    idx, dist, _ = oknn.knn_sklearn(x, 3)
    got = _fuzzy_vs_oracle(K, idx, dist, 3)
    np.testing.assert_allclose(got.toarray(), neighbors_toy["connectivities_umap"], rtol=1e-5, atol=1e-6)


def test_fuzzy_fixture_golden(K, pbmc68k):
    d = pbmc68k["distances"]
    k = pbmc68k["n_neighbors"]
    idx, dist = oknn.indices_distances_from_sparse(d, k)
    order = np.argsort(dist, axis=1, kind="stable")
    idx, dist = np.take_along_axis(idx, order, 1), np.take_along_axis(dist, order, 1)
    got = _fuzzy_vs_oracle(K, idx, dist, k)
    ref = pbmc68k["connectivities"].copy()
    ref.sort_indices()
    np.testing.assert_array_equal(got.indices, ref.indices)
    np.testing.assert_allclose(got.data, ref.data, atol=1e-5)


@pytest.mark.parametrize(("n", "k"), [(20000, 15), (3000, 30), (500, 5)])
def test_fuzzy_synthetic(K, n, k):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 20)).astype(np.float32)
    x[: n // 50] = x[n // 50 : 2 * (n // 50)]  # some exact duplicates -> zero distances
    idx, dist, _ = K.knn(_dev(x), k)
    _fuzzy_vs_oracle(K, idx.cpu().numpy(), dist.cpu().numpy(), k)


def test_fuzzy_hub_rows(K):
    """Rows of the symmetric graph far longer than a wave (three hubs that are neighbours of most points: hand-made kNN
    lists, the kernel does not look at geometry): the per-row sort ranks them in register chunks of 64; pattern and
    weights against the oracle."""
    rng = np.random.default_rng(11)
    n, k = 3000, 10
    idx = np.empty((n, k), dtype=np.int32)
    for i in range(n):
        others = rng.choice(n - 4, size=k - 2, replace=False) + 3  # never a hub, maybe i itself
        others = others[others != i][: k - 2]
        while others.size < k - 2:
            c = int(rng.integers(3, n))
            if c != i and c not in others:
                others = np.append(others, c)
        hub = int(rng.integers(0, 3))
        idx[i] = [i, hub if hub != i else (hub + 1) % 3, *others]
    dist = np.sort(rng.random((n, k)).astype(np.float32) + 0.1, axis=1)
    dist[:, 0] = 0.0
    got = _fuzzy_vs_oracle(K, idx, dist.astype(np.float64), k)
    assert np.diff(got.indptr).max() > 500, np.diff(got.indptr).max()


# ---------------------------------------------------------------------------------------------------
def _rand_csr(n, g, density, seed):
    rng = np.random.default_rng(seed)
    m = sparse.random(n, g, density=density, format="csr", dtype=np.float32, random_state=rng,
                      data_rvs=lambda s: np.log1p(np.exp(rng.standard_normal(s))).astype(np.float32))
    m.sort_indices()
    return m


def _csr_dev(m):
    return _dev(m.indptr.astype(np.int64)), _dev(m.indices.astype(np.int32)), _dev(m.data.astype(np.float32))


@pytest.mark.parametrize(("n", "g", "density"), [(5000, 2000, 0.05), (1234, 77, 0.3), (20000, 500, 0.02), (3, 5, 0.9)])
def test_csr_transpose_and_stats(K, n, g, density):
    m = _rand_csr(n, g, density, n + g)
    ip, ix, dv = _csr_dev(m)
    t_ip, t_ix, t_dv = K.csr_transpose(ip, ix, dv, n, g)
    ref = m.tocsc()
    ref.sort_indices()
    np.testing.assert_array_equal(t_ip.cpu().numpy(), ref.indptr)
    np.testing.assert_array_equal(t_ix.cpu().numpy(), ref.indices)
    np.testing.assert_array_equal(t_dv.cpu().numpy(), ref.data)
    s, q = K.csr_row_stats(t_ip, t_dv, g)
    d64 = m.astype(np.float64)
    np.testing.assert_allclose(s.cpu().numpy(), np.asarray(d64.sum(0)).ravel(), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(q.cpu().numpy(), np.asarray(d64.multiply(d64).sum(0)).ravel(), rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize(("n", "g", "l"), [(5000, 2000, 64), (1000, 300, 50), (2000, 100, 128), (700, 765, 7)])
def test_spmm(K, n, g, l):
    m = _rand_csr(n, g, 0.05, n + l)
    rng = np.random.default_rng(l)
    b = rng.standard_normal((g, l)).astype(np.float32)
    shift = rng.standard_normal(l).astype(np.float32)
    ip, ix, dv = _csr_dev(m)
    y = K.spmm(ip, ix, dv, n, g, _dev(b), _dev(shift)).cpu().numpy()
    ref = m.astype(np.float64) @ b.astype(np.float64) - shift.astype(np.float64)[None, :]
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)
    # transposed product with float64 accumulation on the CSC copy
    t_ip, t_ix, t_dv = K.csr_transpose(ip, ix, dv, n, g)
    yd = _dev(y.astype(np.float32))
    cs = K.colsum(yd)
    np.testing.assert_allclose(cs.cpu().numpy(), y.astype(np.float64).sum(0), rtol=1e-12, atol=1e-9)
    mu = rng.standard_normal(g)
    w = K.spmm_f64acc(t_ip, t_ix, t_dv, g, yd, _dev(mu), cs).cpu().numpy()
    ref_w = m.astype(np.float64).T @ y.astype(np.float64) - np.outer(mu, y.astype(np.float64).sum(0))
    np.testing.assert_allclose(w, ref_w, rtol=1e-11, atol=1e-9)


def test_spmm_long_rows(K):
    """Rows longer than one segment (the CSC copy of a tall matrix)."""
    m = _rand_csr(50, 30000, 0.4, 3)  # 12k entries per row
    rng = np.random.default_rng(0)
    b = rng.standard_normal((30000, 64)).astype(np.float32)
    ip, ix, dv = _csr_dev(m)
    w = K.spmm_f64acc(ip, ix, dv, 50, _dev(b)).cpu().numpy()
    np.testing.assert_allclose(w, m.astype(np.float64) @ b.astype(np.float64), rtol=1e-12, atol=1e-10)
    y = K.spmm(ip, ix, dv, 50, 30000, _dev(b)).cpu().numpy()
    np.testing.assert_allclose(y, m.astype(np.float64) @ b.astype(np.float64), rtol=1e-4, atol=1e-3)


@pytest.mark.gpu
# (1234, 77, 0.3) / (3000, 300, 0.2): more than 8 entries per (row, tile) on average, the round-2 kernel; (2100, 300, 0.05):
# three row blocks, the last one partial; (300, 8500, 0.01): more than 64 gene tiles, the round-2 kernel
@pytest.mark.parametrize(("n", "g", "density"), [(5000, 2000, 0.05), (1234, 77, 0.3), (3000, 300, 0.2), (3, 5, 0.9), (9000, 513, 0.02),
                                                 (2100, 300, 0.05), (300, 8500, 0.01)])
def test_csr_gram_bit_exact(K, n, g, density):
    """scamd_csr_gram_f32: integer (fixed point) Gram matrix and column sums, bit-exact against numpy int64."""
    import torch
    from scipy import sparse

    rng = np.random.default_rng(n + g)
    x = sparse.random(n, g, density=density, random_state=rng, format="csr", dtype=np.float32)
    x.data = np.log1p(np.exp(rng.standard_normal(x.nnz))).astype(np.float32)
    x.sort_indices()
    ip = torch.from_numpy(x.indptr.astype(np.int64)).cuda()
    ix = torch.from_numpy(x.indices.astype(np.int32)).cuda()
    dv = torch.from_numpy(x.data).cuda()
    mx = K.csr_absmax(ip, ix, dv, n, g)
    assert mx == float(np.abs(x.data).max())
    sb = 30
    gq, cq = K.csr_gram(ip, ix, dv, n, g, sb)
    gq, cq = gq.cpu().numpy(), cq.cpu().numpy()
    gp = (g + 127) // 128 * 128
    assert gq.shape == (gp, gp) and (gq[g:] == 0).all() and (gq[:, g:] == 0).all() and (cq[g:] == 0).all()
    sc = 2.0 ** sb
    ref = np.zeros((g, g), dtype=np.int64)
    xd = x.astype(np.float64).toarray()
    for row in xd:
        nz = np.flatnonzero(row)
        ref[np.ix_(nz, nz)] += np.rint(np.outer(row[nz], row[nz] * sc)).astype(np.int64)
    np.testing.assert_array_equal(gq[:g, :g], ref)
    np.testing.assert_array_equal(cq[:g], np.rint(xd * sc).astype(np.int64).sum(axis=0))
    # determinism: the accumulation order is arbitrary, the integers are not
    gq2, _ = K.csr_gram(ip, ix, dv, n, g, sb)
    np.testing.assert_array_equal(gq2.cpu().numpy(), gq)


@pytest.mark.gpu
def test_csr_gram_rows_with_crowded_tiles(K):
    """A matrix that is sparse on average (the packed kernels take it) with rows that crowd a 128-gene tile: 17 .. 128 entries in
    one tile finish from the CSR arrays, next to rows with 9 .. 16 (the exchanged-roles rounds) and empty rows."""
    import torch
    from scipy import sparse

    rng = np.random.default_rng(11)
    n, g, sb = 2500, 300, 30
    x = sparse.random(n, g, density=0.03, random_state=rng, format="lil", dtype=np.float32)
    for r in rng.choice(n, size=120, replace=False):
        t0 = 128 * int(rng.integers(0, 3))
        width = min(128, g - t0)
        cnt = int(rng.integers(9, width + 1))
        cols = t0 + rng.choice(width, size=cnt, replace=False)
        x[r, cols] = 1.0
    x = x.tocsr()
    x[rng.choice(n, size=50, replace=False)] = 0  # empty rows
    x.eliminate_zeros()
    x.data = np.log1p(np.exp(rng.standard_normal(x.nnz))).astype(np.float32)
    x.sort_indices()
    assert x.nnz <= 8 * n * 3  # the packed kernels' side of the dispatch
    ip = torch.from_numpy(x.indptr.astype(np.int64)).cuda()
    ix = torch.from_numpy(x.indices.astype(np.int32)).cuda()
    dv = torch.from_numpy(x.data).cuda()
    gq, cq = K.csr_gram(ip, ix, dv, n, g, sb)
    gq, cq = gq.cpu().numpy(), cq.cpu().numpy()
    xd = x.astype(np.float64).toarray()
    ref = np.zeros((g, g), dtype=np.int64)
    for row in xd:
        nz = np.flatnonzero(row)
        ref[np.ix_(nz, nz)] += np.rint(np.outer(row[nz], row[nz] * 2.0**sb)).astype(np.int64)
    np.testing.assert_array_equal(gq[:g, :g], ref)
    np.testing.assert_array_equal(cq[:g], np.rint(xd * 2.0**sb).astype(np.int64).sum(axis=0))


@pytest.mark.gpu
def test_csr_gram_products_beyond_the_fast_rounding(K):
    """Products of 2^51 and more leave the range of the fast kernel's 2^52 rounding trick: it raises a device flag, the
    sums are cleared and the exact instantiation (llrint) runs -- all stream-ordered.  A few huge values among ordinary
    ones, bit-exact against Python integers; a second, all-small call on the same workspace must not see a stale flag."""
    import torch
    from fractions import Fraction
    from scipy import sparse

    rng = np.random.default_rng(5)
    n, g, sb = 700, 300, 36
    x = sparse.random(n, g, density=0.08, random_state=rng, format="csr", dtype=np.float32)
    x.data = np.log1p(np.exp(rng.standard_normal(x.nnz))).astype(np.float32)
    big = rng.choice(x.nnz, size=25, replace=False)
    x.data[big] = rng.uniform(300.0, 900.0, size=25).astype(np.float32)  # 300^2 * 2^36 = 6e15 > 2^51
    x.sort_indices()

    def run(m):
        ip = torch.from_numpy(m.indptr.astype(np.int64)).cuda()
        ix = torch.from_numpy(m.indices.astype(np.int32)).cuda()
        dv = torch.from_numpy(m.data).cuda()
        gq, cq = K.csr_gram(ip, ix, dv, n, g, sb)
        return gq.cpu().numpy()[:g, :g], cq.cpu().numpy()[:g]

    def exact(m):
        ref = [[0] * g for _ in range(g)]
        for r in range(n):
            lo, hi = m.indptr[r], m.indptr[r + 1]
            cols, vals = m.indices[lo:hi], m.data[lo:hi]
            for ja, va in zip(cols, vals):
                for jb, vb in zip(cols, vals):
                    # the kernel rounds fl64(va * fl64(vb * 2^S)) to nearest-even: both products are exact in float64
                    ref[ja][jb] += round(Fraction(float(va)) * Fraction(float(vb)) * 2 ** sb)
        return np.array(ref, dtype=object)

    gq, _ = run(x)
    assert float(np.abs(x.data).max()) ** 2 * 2.0 ** sb > 2.0 ** 51
    ref = exact(x)
    assert (gq.astype(object) == ref).all()
    small = x.copy()
    small.data = np.minimum(small.data, 8.0).astype(np.float32)
    gs, _ = run(small)
    assert (gs.astype(object) == exact(small)).all()


@pytest.mark.parametrize("d", [50, 20])
def test_knn_cell_pruned_equals_brute_force(K, monkeypatch, d):
    """the exact cell-pruned sweep returns what the brute-force sweep returns (bitwise: both end in the same float64
    re-rank), for all queries and for a shard of queries [q_begin, q_begin + n_query) -- the multi-GPU call pattern"""
    from scanpy_amd.datasets import blobs_embedding

    n = 12000
    x, _ = blobs_embedding(n, d, n_types=9, seed=21)
    xd = _dev(x)
    monkeypatch.setenv("SCAMD_KNN_IVF", "0")
    i0, d0, _ = K.knn(xd, 15)
    monkeypatch.setenv("SCAMD_KNN_IVF", "1")
    i1, d1, _ = K.knn(xd, 15)
    np.testing.assert_array_equal(i0.cpu().numpy(), i1.cpu().numpy())
    np.testing.assert_array_equal(d0.cpu().numpy(), d1.cpu().numpy())
    for qb, nq in ((0, 1000), (3000, 5000), (11000, 1000), (4097, 129)):
        i2, d2, _ = K.knn(xd, 15, q_begin=qb, n_query=nq)
        np.testing.assert_array_equal(i0[qb:qb + nq].cpu().numpy(), i2.cpu().numpy())
        np.testing.assert_array_equal(d0[qb:qb + nq].cpu().numpy(), d2.cpu().numpy())


def test_knn_cell_pruned_query_shards_at_default_size(K):
    """n >= 65536 takes the cell-pruned sweep by default; two half shards of queries (what two ranks compute) equal
    the single-call result"""
    from scanpy_amd.datasets import blobs_embedding

    n = 70000
    x, _ = blobs_embedding(n, 50, seed=22)
    xd = _dev(x)
    i0, d0, _ = K.knn(xd, 15)
    half = n // 2
    for qb, nq in ((0, half), (half, n - half)):
        i1, d1, _ = K.knn(xd, 15, q_begin=qb, n_query=nq)
        np.testing.assert_array_equal(i0[qb:qb + nq].cpu().numpy(), i1.cpu().numpy())
        np.testing.assert_array_equal(d0[qb:qb + nq].cpu().numpy(), d1.cpu().numpy())
