"""The certificate that makes the bf16 kNN engine exact, made falsifiable ("next round" item 1).

The search the reference runs is sklearn's exact brute force (src/scanpy/neighbors/__init__.py:754-768); ours filters with
three bf16 MFMA products per pair and certifies every query in float64 against an error bound
(csrc/knn.hip: knn_rerank_rows_kernel).  A bound that is too small stays invisible on friendly data -- round 3 shipped one
2-4x under-priced for a while and every test stayed green.  Three kinds of test close that hole:

(a) inputs the centring cannot remove (clusters at +-3000 along different axes, rows with one huge coordinate, near-duplicate
    pairs far from the origin, and the split's own worst-case values) at n >= 100k, so the pruned sweep runs, against a
    float64 brute force;
(b) a MUTATION test: the same worst-case input with cert_scale = 0.25 and the second tier switched off must produce rows
    that differ from the float64 answer -- if it did not, the inputs would not be adversarial enough to tell a sound
    certificate from an unsound one -- while cert_scale = 1 gives the exact answer;
(c) the bound itself, MEASURED: raw scores of the engine (the select kernel's own packing, operand construction and MFMA
    chain, scamd_knn_debug_b3_scores_f32) against float64 over > 1e8 pairs; max |error| / bound must stay below 1 and is
    printed (it also measures the accumulate rounding inside v_mfma_f32_32x32x16_bf16 instead of assuming it).
"""
from __future__ import annotations

import numpy as np
import pytest

from oracle import compare as cmp
from oracle import knn as oknn

pytestmark = pytest.mark.gpu

U = 2.0 ** -24
# worst case of the hi + lo split (found by exhaustive search over all float32 mantissas, see _split_error below):
# a = 2^15 (1 + 2^-8 + 2^-17): hi = bf16(a) rounds UP, lo = -2^-8 exactly, residual 2^-17 -- the pair (a, a) loses
# 1.988 * 2^-16 * a^2 of its product; the decoy b = 2^15 (1 + 2^-8 + 2^-22) paired with a loses 1.498 * 2^-16 * a * b.
A0 = np.float32(32768.0 * (1.0 + 2.0 ** -8 + 2.0 ** -17))
B0 = np.float32(32768.0 * (1.0 + 2.0 ** -8 + 2.0 ** -22))


@pytest.fixture(scope="module")
def K():
    from scanpy_amd import _kernels

    return _kernels


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bf16_rn(x):
    b = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(np.float32)


def _split_error(q, c):
    """q * c - (qh ch + qh cl + ql ch) for float32 scalars: what the three bf16 products drop (host model of the split)"""
    q, c = np.float32(q), np.float32(c)
    qh, ch = _bf16_rn(q), _bf16_rn(c)
    ql, cl = _bf16_rn(np.float32(q - qh)), _bf16_rn(np.float32(c - ch))
    qh, ql, ch, cl = (float(v) for v in (qh, ql, ch, cl))
    return float(q) * float(c) - (qh * ch + qh * cl + ql * ch)


def test_host_model_of_the_split_worst_case():
    """the constants the adversarial inputs are built from (no GPU needed, but kept beside its users)"""
    eaa = _split_error(A0, A0) / (float(A0) ** 2) * 65536
    eab = _split_error(A0, B0) / (float(A0) * float(B0)) * 65536
    assert 1.95 < eaa < 2.0 and 1.45 < eab < 1.55, (eaa, eab)


def _quant(v, q=64.0):
    """multiples of 1 / q: column sums of mirrored data are then EXACT in float64, i.e. the mean is exactly 0"""
    return (np.round(np.asarray(v, dtype=np.float64) * q) / q).astype(np.float32)


def _worst_case_clumps(n_total=131072, n_clumps=128, n_a=24, n_b=40, seed=0):
    """Mirrored data set (x and -x: column means exactly 0, the image holds the rows as they are) of
    * clumps: n_a "true neighbour" rows with coordinate 0 = A0 and n_b decoys with coordinate 0 = B0, offset by 24 along
      axis 1 -- the decoys are 576 FARTHER in squared distance, but the engine sees the A-A pairs 65.7k farther than they
      are and the A-B pairs only 49.5k: it fills every A query's list with decoys;
    * background: four clusters at +-3000 along axes 2 and 3.
    -> (x float32 [n_total, 50], rows of the A members)"""
    rng = np.random.default_rng(seed)
    d = 50
    rows, a_rows = [], []
    pos = 0
    for _ in range(n_clumps):
        centre = _quant(rng.standard_normal(d) * 200.0)
        centre[0] = 0.0
        a = centre[None, :] + _quant(rng.standard_normal((n_a, d)))
        a[:, 0] = A0
        b = centre[None, :] + _quant(rng.standard_normal((n_b, d)))
        b[:, 0] = B0
        b[:, 1] += np.float32(24.0)
        rows += [a, b]
        a_rows.append(np.arange(pos, pos + n_a))
        pos += n_a + n_b
    n_bg = n_total // 2 - pos
    axes = rng.integers(2, 4, n_bg)
    sign = rng.choice([-1.0, 1.0], n_bg)
    bg = _quant(rng.standard_normal((n_bg, d)))
    bg[np.arange(n_bg), axes] += np.float32(3000.0) * sign.astype(np.float32)
    half = np.vstack(rows + [bg]).astype(np.float32)
    x = np.vstack([half, -half])
    a_rows = np.concatenate(a_rows)
    a_rows = np.concatenate([a_rows, a_rows + half.shape[0]])
    perm = rng.permutation(x.shape[0])
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    return np.ascontiguousarray(x[perm]), np.sort(inv[a_rows])


def _adversarial(kind, n, seed):
    rng = np.random.default_rng(seed)
    d = 50
    if kind == "axes4":  # two / four clusters at +-3000 along different axes: no global shift removes the norms
        axes = rng.integers(0, 2, n)
        sign = rng.choice([-1.0, 1.0], n).astype(np.float32)
        x = rng.standard_normal((n, d)).astype(np.float32)
        x[np.arange(n), axes] += np.float32(3000.0) * sign
        return x, np.arange(n)
    if kind == "huge_coordinate":  # 0.5 % of the rows carry one coordinate of 1e4: the largest norm is 1e8, most are ~50
        from scanpy_amd.datasets import blobs_embedding

        x, _ = blobs_embedding(n, d, n_types=16, seed=seed)
        out = rng.choice(n, n // 200, replace=False)
        x[out, rng.integers(0, d, out.size)] = np.float32(1.0e4) * rng.choice([-1.0, 1.0], out.size).astype(np.float32)
        return x, out
    if kind == "near_duplicates":  # pairs 1e-3 apart at norm 3000: their gap is far below 3 * 2^-16 * 2 |q| |c| = 824
        m = n // 2
        axes = rng.integers(0, 3, m)
        base = rng.standard_normal((m, d)).astype(np.float32) * np.float32(4.0)
        base[np.arange(m), axes] += np.float32(3000.0) * rng.choice([-1.0, 1.0], m).astype(np.float32)
        dup = base + (rng.standard_normal((m, d)) * 2e-4).astype(np.float32)
        x = np.vstack([base, dup]).astype(np.float32)
        return x, np.arange(n)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["axes4", "huge_coordinate", "near_duplicates"])
def test_knn_adversarial_norms_pruned_sweep(K, kind):
    """(a): inputs whose norms survive the centring, n >= 100k (the pruned sweep, the bf16 engine); sampled float64 brute force"""
    from scanpy_amd import _lib

    n, k = 100_000, 15
    x, special = _adversarial(kind, n, 77)
    idx, dist, nfb = K.knn(_dev(x), k)
    lib = _lib.load()
    # (huge_coordinate: an outlier in a cell makes that cell's ball cover everything -- nothing is pruned there, which is
    # a cost, not an error)
    assert int(lib.scamd_knn_last_select_engine()) == 1
    assert kind == "huge_coordinate" or float(lib.scamd_knn_last_select_pairs()) < 0.9 * n * n
    rng = np.random.default_rng(5)
    qs = np.unique(np.concatenate([rng.choice(n, 1500, replace=False), rng.choice(special, min(500, special.size), replace=False)]))
    ri, rd = oknn.knn_exact_f64_sample(x, qs, k)
    bad, differ = cmp.knn_rows_differing_beyond_ties(idx.cpu().numpy()[qs], dist.cpu().numpy()[qs], ri, rd)
    t2 = int(lib.scamd_knn_last_second_tier_queries())
    print(f"{kind}: rows differing {differ} (beyond ties {bad}) of {qs.size}; second tier {t2}, float64 scans {nfb} of {n}")
    assert bad == 0
    if kind == "huge_coordinate":
        # the outliers must not drag everybody else's bound up (round 4: the bound uses the norms a missed neighbour can have)
        assert nfb + t2 < n // 20, "one far row inflates every query's certificate again"


def test_knn_certificate_mutation_is_detected(K, monkeypatch):
    """(b): worst-case split values.  With the certificate as shipped the A queries are rejected (their lists hold decoys)
    and redone exactly; with the bound scaled by 0.25 and no second tier the SAME run certifies wrong lists -- the test
    data can tell a sound bound from an unsound one"""
    from scanpy_amd import _lib

    lib = _lib.load()
    k = 15
    x, a_rows = _worst_case_clumps()
    n = x.shape[0]
    assert np.abs(x.astype(np.float64).sum(0)).max() == 0.0  # the image is x itself
    xd = _dev(x)
    ri, rd = oknn.knn_exact_f64_sample(x, a_rows, k)

    monkeypatch.setenv("SCAMD_KNN_TIER2_MIN", str(1 << 30))
    i0, d0, nf0 = K.knn(xd, k, cert_scale=0.25)
    assert int(lib.scamd_knn_last_select_engine()) == 1 and float(lib.scamd_knn_last_select_pairs()) < 0.9 * n * n
    bad0, _ = cmp.knn_rows_differing_beyond_ties(i0.cpu().numpy()[a_rows], d0.cpu().numpy()[a_rows], ri, rd)
    print(f"cert_scale 0.25: {bad0} of {a_rows.size} worst-case queries wrong beyond ties, float64 scans {nf0}")
    assert bad0 >= a_rows.size // 2, "an under-priced certificate went undetected: the inputs are not adversarial enough"

    i1, d1, nf1 = K.knn(xd, k)
    bad1, differ1 = cmp.knn_rows_differing_beyond_ties(i1.cpu().numpy()[a_rows], d1.cpu().numpy()[a_rows], ri, rd)
    print(f"cert_scale 1: {bad1} wrong ({differ1} differ at ties), float64 scans {nf1}")
    assert bad1 == 0 and nf1 >= a_rows.size  # every one of them had to be rejected
    monkeypatch.delenv("SCAMD_KNN_TIER2_MIN")
    i2, d2, nf2 = K.knn(xd, k)  # default path: second tier first
    np.testing.assert_array_equal(i1.cpu().numpy(), i2.cpu().numpy())
    np.testing.assert_array_equal(d1.cpu().numpy(), d2.cpu().numpy())
    # the rest of the rows (background at +-3000) against the float64 brute force as well
    qs = np.sort(np.random.default_rng(9).choice(n, 1500, replace=False))
    bi, bd = oknn.knn_exact_f64_sample(x, qs, k)
    assert cmp.knn_rows_differing_beyond_ties(i2.cpu().numpy()[qs], d2.cpu().numpy()[qs], bi, bd)[0] == 0


def _score_error_ratio(K, x, q0, nq, c0, nc):
    """max over the block of |engine score - exact score| / bound(q, c); exact = |x_q - x_c|^2 - |fl(x_q - mu)|^2 in float64
    (what knn_rerank_rows_kernel compares a threshold with), bound = u (cert_k (|c|^2 + 2 |q||c|) + cert_k2 2 |q||c|)"""
    import torch

    cert_k, cert_k2, _ = K.knn_cert_factors(1)
    xd = _dev(x)
    sc, mu, cmax = K.knn_debug_b3_scores(xd, q0, nq, c0, nc)
    x64 = xd.double()
    mu32 = mu.float()
    img = (xd - mu32[None, :]).double()  # fl(x - mu): float32 subtraction, as the image packing does
    qn = (img[q0:q0 + nq] ** 2).sum(1)
    cn = (img[c0:c0 + nc] ** 2).sum(1)
    worst, worst_global, worst_abs = 0.0, 0.0, 0.0
    for s in range(0, nq, 1024):
        q = x64[q0 + s:q0 + s + 1024]
        c = x64[c0:c0 + nc]
        d2 = (q * q).sum(1)[:, None] + (c * c).sum(1)[None, :] - 2.0 * (q @ c.T)
        exact = d2 - qn[s:s + 1024, None]
        err = (sc[s:s + 1024].double() - exact).abs()
        qc = torch.sqrt(qn[s:s + 1024, None] * cn[None, :])
        bound = U * (cert_k * (cn[None, :] + 2.0 * qc) + cert_k2 * 2.0 * qc)
        boundg = U * (cert_k * (cmax + 2.0 * torch.sqrt(qn[s:s + 1024, None] * cmax)) + cert_k2 * 2.0 * torch.sqrt(qn[s:s + 1024, None] * cmax))
        worst = max(worst, float((err / bound).max()))
        worst_global = max(worst_global, float((err / boundg).max()))
        worst_abs = max(worst_abs, float(err.max()))
    return worst, worst_global, worst_abs


def test_knn_engine_score_error_stays_below_the_certificates_bound(K):
    """(c): > 1e8 pairs per input; the ratio is the number DESIGN 3.1 quotes"""
    from scanpy_amd.datasets import blobs_embedding

    n, nq, nc = 32768, 8192, 16384
    report = {}
    x, _ = blobs_embedding(n, 50, n_types=12, seed=3)
    report["blobs"] = _score_error_ratio(K, x, 0, nq, 8192, nc)
    report["blobs + 3000"] = _score_error_ratio(K, x + np.float32(3000.0), 0, nq, 8192, nc)
    report["noise x 1000"] = _score_error_ratio(K, np.random.default_rng(1).standard_normal((n, 50)).astype(np.float32) * 1000, 0, nq, 8192, nc)
    # every coordinate at the split's worst-case mantissa (random signs and binades): the errors of all 50 products add up
    # (mirrored, x and -x: the column means are exactly 0 and the image holds these mantissas, not fl(x - mu))
    rng = np.random.default_rng(2)
    half = (np.float32(1.0 + 2.0 ** -8 + 2.0 ** -17) * np.exp2(rng.integers(-3, 6, (n // 2, 50))).astype(np.float32)
            * rng.choice([-1.0, 1.0], (n // 2, 50)).astype(np.float32))
    report["worst mantissa, all coordinates"] = _score_error_ratio(K, np.vstack([half, -half]), 0, nq, 8192, nc)
    xc, _ = _worst_case_clumps(n_total=32768, n_clumps=96)
    report["worst-case clumps"] = _score_error_ratio(K, xc, 0, nq, 8192, nc)
    # all products of a pair of the same sign: nothing cancels in q.c (queries and candidates from both orthants)
    report["worst mantissa, two orthants"] = _score_error_ratio(K, np.vstack([np.abs(half), -np.abs(half)]), 12288, nq, 8192, nc)
    for name, (r, rg, ea) in report.items():
        print(f"{name:34s} max |score_b3 - score_f64| / bound = {r:.4f} (against the kernel's global-norm bound {rg:.4f}; max |error| {ea:.4g})")
    assert all(r < 1.0 for r, _, _ in report.values()), report
    # the adversarial sets must actually load the bound (else this test could not see an under-priced factor either)
    assert max(r for r, _, _ in report.values()) > 0.25
