"""Parity where it is not trivially true ("next round" item 1):

* the exact cell-pruned kNN sweep on data far from the origin and on overlapping / anisotropic clusters
  (bitwise equal to the brute-force sweep, index sets equal to the reference's sklearn call);
* Leiden on graphs that are NOT unambiguously separable: the oracle's own seed-to-seed agreement is the noise
  floor, the GPU partition must sit inside the oracle's seed distribution (modularity) and agree with the oracle
  at least as well as the oracle agrees with itself;
* Leiden anchored on the labels the REFERENCE produced: the bundled fixture's `louvain` codes
  (src/scanpy/datasets/_datasets.py:349-427), bar = the reference's own cross-implementation bar NMI > 0.9
  (tests/test_clustering.py:130-163) at matched cluster count;
* the whole chain at BASELINE configs[1] (100k x 2k) against the sklearn / oracle CPU chain, stage by stage.
"""
from __future__ import annotations

import numpy as np
import pytest
from scipy import sparse
from sklearn.metrics import adjusted_rand_score, normalized_mutual_info_score

from oracle import compare as cmp
from oracle import connectivities as oc
from oracle import knn as oknn
from oracle import leiden as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from scanpy_amd import _kernels

    return _kernels


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _graph_dev(adj):
    adj = sparse.csr_matrix(adj)
    adj.sort_indices()
    return _dev(adj.indptr.astype(np.int64)), _dev(adj.indices.astype(np.int32)), _dev(adj.data.astype(np.float32)), adj.shape[0]


# ---------------------------------------------------------------------------------------------------------------------
# pruned kNN sweep: offset data, overlapping and anisotropic clusters

def _overlapping(n, d, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "offset":  # ||x||^2 ~ 4.5e6 >> d^2 ~ 100: thresholds in score space lose 3 digits in float32
        from scanpy_amd.datasets import blobs_embedding

        x, _ = blobs_embedding(n, d, n_types=12, seed=seed)
        return x + np.float32(300.0)
    if kind == "offset_far":
        from scanpy_amd.datasets import blobs_embedding

        x, _ = blobs_embedding(n, d, n_types=12, seed=seed)
        return x + np.float32(3000.0)
    if kind == "overlap":  # centres 1 sigma apart: every cell's ball intersects every other's
        centers = rng.standard_normal((16, d)).astype(np.float32)
        return (centers[rng.integers(0, 16, n)] + rng.standard_normal((n, d))).astype(np.float32)
    if kind == "anisotropic":  # elongated clusters + a continuous trajectory between two of them
        centers = rng.standard_normal((10, d)).astype(np.float32) * 3
        scales = np.exp(rng.uniform(-1.5, 1.5, size=(10, d))).astype(np.float32)
        lab = rng.integers(0, 10, n)
        x = centers[lab] + scales[lab] * rng.standard_normal((n, d)).astype(np.float32)
        t = rng.random(n // 5).astype(np.float32)[:, None]
        x[: n // 5] = centers[0] * (1 - t) + centers[1] * t + 0.2 * rng.standard_normal((n // 5, d)).astype(np.float32)
        return np.ascontiguousarray(x, dtype=np.float32)
    if kind == "noise":  # no structure at all: nothing can be pruned, the rule must never stop early
        return rng.standard_normal((n, d)).astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["offset", "offset_far", "overlap", "anisotropic", "noise"])
def test_knn_cell_pruned_hard_inputs(K, monkeypatch, kind):
    n, d, k = 20000, 50, 15
    x = _overlapping(n, d, 31, kind)
    xd = _dev(x)
    monkeypatch.setenv("SCAMD_KNN_IVF", "0")
    i0, d0, nf0 = K.knn(xd, k)
    monkeypatch.setenv("SCAMD_KNN_IVF", "1")
    # (offset_far: ||x||^2 = 4.5e8, one float32 ulp of a raw score is of the order of the squared neighbour distances;
    # the search centres the image first -- csrc/knn.hip -- so these queries are certified like any other: a ScamdError
    # here would be a regression, it is no longer an accepted outcome)
    i1, d1, nf1 = K.knn(xd, k)
    np.testing.assert_array_equal(i0.cpu().numpy(), i1.cpu().numpy())
    np.testing.assert_array_equal(d0.cpu().numpy(), d1.cpu().numpy())
    # ... and both equal the reference's sklearn call (exact float64 distances of the float32 points)
    ri, rd, _ = oknn.knn_sklearn(x, k, n_jobs=-1)
    if kind.startswith("offset"):
        # sklearn's own expansion ||x||^2 - 2xy + ||y||^2 loses digits this far from the origin: exact float64 oracle
        ri, rd = oknn.knn_exact_f64(x, np.arange(n), k)
    bad, differ = cmp.knn_rows_differing_beyond_ties(i1.cpu().numpy(), d1.cpu().numpy(), ri, rd)
    print(f"{kind}: rows differing {differ} (beyond ties {bad}); float64 fallbacks brute {nf0} / pruned {nf1}")
    assert bad == 0


def test_knn_cell_pruned_offset_at_default_size(K):
    """n >= 65536 takes the pruned sweep by default; offset embedding; sampled float64 brute force on the host"""
    from scanpy_amd.datasets import blobs_embedding

    n, k = 100_000, 15
    x, _ = blobs_embedding(n, 50, n_types=20, seed=5)
    x = x + np.float32(250.0)
    idx, dist, nfb = K.knn(_dev(x), k)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    qs = np.sort(np.random.default_rng(1).choice(n, 2000, replace=False))
    ri, rd = oknn.knn_exact_f64(x, qs, k)
    bad, differ = cmp.knn_rows_differing_beyond_ties(idx[qs], dist[qs], ri, rd)
    print(f"rows differing {differ} (beyond ties {bad}), fallbacks {nfb}")
    assert bad == 0


# ---------------------------------------------------------------------------------------------------------------------
# Leiden on graphs that are not trivially separable

def _knn_graph(x, k=15):
    idx, dist, _ = oknn.knn_sklearn(x, k, n_jobs=-1)
    c, _, _ = oc.fuzzy_simplicial_set(idx, dist, x.shape[0], k)
    return c


@pytest.mark.parametrize("spread", [2.0, 3.0, 4.0, 5.0])
def test_leiden_overlapping_blobs_vs_oracle_seed_distribution(K, spread):
    """blobs whose spread approaches the centre distance (4 sigma per axis -> |c_a - c_b| ~ 40, in-cluster diameter
    ~ 2 * spread * 7): from clean (2.0) to heavily overlapping (5.0).  The oracle's seeds 0..4 define the distribution;
    the GPU result must (a) reach the oracle's modularity range, (b) agree with the oracle's seed-0 partition at least
    as well as the oracle's other seeds do (minus a small margin)."""
    from scanpy_amd.datasets import blobs_embedding

    n = 30000
    x, truth = blobs_embedding(n, 50, n_types=24, spread=spread, seed=int(spread * 10))
    adj = _knn_graph(x)
    ip, ix, w, _ = _graph_dev(adj)
    oracle = [ol.leiden(adj, seed=s) for s in range(5)]
    q_or = np.array([q for _, q in oracle])
    floor = min(adjusted_rand_score(oracle[0][0], m) for m, _ in oracle[1:])
    m, q, nc = K.leiden(ip, ix, w, n, seed=0)
    m = m.cpu().numpy()
    ari_o = adjusted_rand_score(oracle[0][0], m)
    ari_t = adjusted_rand_score(truth, m)
    ari_t_or = adjusted_rand_score(truth, oracle[0][0])
    print(f"spread {spread}: gpu Q {q:.5f} nc {nc} | oracle Q [{q_or.min():.5f}, {q_or.max():.5f}] nc "
          f"{[int(mm.max()) + 1 for mm, _ in oracle]} | ARI gpu-oracle {ari_o:.4f}, oracle seed floor {floor:.4f}, "
          f"ARI truth gpu {ari_t:.4f} oracle {ari_t_or:.4f}")
    assert abs(q - ol.modularity(adj, m)) < 1e-8
    assert q >= q_or.min() - 2e-3 * max(1.0 - q_or.min(), 0.05), "modularity below the oracle's seed distribution"
    assert ari_o >= min(0.99, floor - 0.01)
    assert ari_t >= ari_t_or - 0.02


def test_leiden_fixture_seed_distribution_and_reference_labels(K, pbmc68k):
    """the real pbmc68k_reduced graph (700 cells, the reference's stored connectivities) over seeds 0..9:
    * our modularity sits inside the oracle's seed distribution (not "oracle - 0.01");
    * NMI against the fixture's `louvain` codes -- labels produced by the REFERENCE pipeline on this very graph -- is
      at least the oracle's, and clears the reference's cross-implementation bar (NMI > 0.9,
      tests/test_clustering.py:130-163) once the cluster counts are matched through the resolution;
    * ARI between the oracle's own seeds is reported as the noise floor."""
    adj = pbmc68k["connectivities"].astype(np.float32)
    ref_labels = pbmc68k["louvain_codes"].astype(np.int32)
    n_ref = int(ref_labels.max()) + 1
    ip, ix, w, n = _graph_dev(adj)
    q_ref = ol.modularity(adj, ref_labels)
    gq, oq, g_nmi, o_nmi, g_lab, o_lab = [], [], [], [], [], []
    for s in range(10):
        m, q, _ = K.leiden(ip, ix, w, n, seed=s)
        m = m.cpu().numpy()
        mo, qo = ol.leiden(adj, seed=s)
        gq.append(q)
        oq.append(qo)
        g_lab.append(m)
        o_lab.append(mo)
        g_nmi.append(normalized_mutual_info_score(ref_labels, m))
        o_nmi.append(normalized_mutual_info_score(ref_labels, mo))
    gq, oq = np.array(gq), np.array(oq)
    floor = np.mean([adjusted_rand_score(o_lab[0], o) for o in o_lab[1:]])
    cross = np.mean([adjusted_rand_score(g, o) for g, o in zip(g_lab, o_lab)])
    print(f"reference louvain labels: {n_ref} clusters, Q {q_ref:.5f}")
    print(f"gpu    Q [{gq.min():.5f}, {gq.max():.5f}] mean {gq.mean():.5f}; NMI vs reference labels mean {np.mean(g_nmi):.4f}")
    print(f"oracle Q [{oq.min():.5f}, {oq.max():.5f}] mean {oq.mean():.5f}; NMI vs reference labels mean {np.mean(o_nmi):.4f}")
    print(f"ARI oracle seed 0 vs seeds 1..9 (noise floor) {floor:.4f}; ARI gpu vs oracle same seed {cross:.4f}")
    assert gq.min() >= oq.min() - 1e-3, "every GPU seed inside the oracle's seed distribution"
    assert gq.mean() >= oq.mean() - 1e-3
    assert gq.min() >= q_ref - 1e-3, "at least the modularity of the reference's own labels"
    assert np.mean(g_nmi) >= np.mean(o_nmi) - 0.02
    assert cross >= floor - 0.01  # (round 2: 0.05; measured 0.9842 against a floor of 0.9884)
    # matched cluster count: scan the resolution until the partition has as many clusters as the reference's
    best = None
    for res in np.linspace(0.4, 1.6, 25):
        m, q, nc = K.leiden(ip, ix, w, n, seed=0, resolution=float(res))
        if nc == n_ref:
            nmi = normalized_mutual_info_score(ref_labels, m.cpu().numpy())
            best = max(best or 0.0, nmi)
    print(f"NMI vs reference labels at matched cluster count ({n_ref}): {best}")
    assert best is not None and best > 0.9


def test_leiden_weak_planted_matrix_chain(K):
    """the chain on the `weak` synthetic matrix (overlapping gene programmes: at this size the CPU chain recovers the
    planted types only partially and its own seeds disagree) -- GPU Leiden on the CPU chain's graph"""
    from oracle import pca as opca
    from scanpy_amd.datasets import synthetic_planted

    n = 30000
    for p in (0.2, 0.12):
        x, truth = synthetic_planted(n, 2000, seed=0, p_programme=p)
        xp = np.ascontiguousarray(opca.pca_reference(x, 50)["X_pca"], dtype=np.float32)
        adj = _knn_graph(xp)
        ip, ix, w, _ = _graph_dev(adj)
        oracle = [ol.leiden(adj, seed=s) for s in range(4)]
        floor = min(adjusted_rand_score(oracle[0][0], m) for m, _ in oracle[1:])
        q_min = min(q for _, q in oracle)
        m, q, nc = K.leiden(ip, ix, w, n, seed=0)
        m = m.cpu().numpy()
        ari_o = adjusted_rand_score(oracle[0][0], m)
        print(f"p_programme {p}: gpu Q {q:.5f} nc {nc}; oracle Q min {q_min:.5f} nc {int(oracle[0][0].max()) + 1}; "
              f"ARI gpu-oracle {ari_o:.4f} (oracle seed floor {floor:.4f}); ARI truth gpu "
              f"{adjusted_rand_score(truth, m):.4f} oracle {adjusted_rand_score(truth, oracle[0][0]):.4f}")
        assert q >= q_min - 2e-3
        assert ari_o >= min(0.99, floor - 0.01)


# ---------------------------------------------------------------------------------------------------------------------
# the whole chain at BASELINE configs[1]

def test_chain_100k_vs_cpu_chain():
    """100k x 2k (BASELINE configs[1]): sklearn PCA(arpack) -> sklearn brute kNN -> oracle fuzzy set -> oracle Leiden
    against sc.pp.pca -> sc.pp.neighbors -> sc.tl.leiden, stage by stage and end to end (bench.py's parity block)."""
    import bench

    n, g = 100_000, 2000
    x, truth = bench.make_matrix(n, g, 0, "planted")
    chain = bench.cpu_chain(x, 50, 15)
    par = bench.parity_block(x, truth, chain, 50, 15)
    print({k: v for k, v in par.items() if k != "gates"})
    assert par["failed_gates"] == []
    assert par["pca_loading_err"] < 1e-4
    assert par["knn_rows_differing_beyond_ties"] == 0
    assert par["conn_max_abs"] < 1e-5 and par["conn_same_pattern"]
    assert par["leiden_ari_vs_cpu_chain"] >= 0.99 and par["leiden_ari_stagewise"] >= 0.99
    assert par["ari_vs_truth"]["gpu"] >= 0.99
    assert par["knn_rows_equal_end_to_end"] > 0.999


def test_knn_second_tier_of_the_bf16_engine(K, monkeypatch):
    """pruned search, bf16 engine: queries its (looser) certificate rejects are re-done by the float32 engine before
    anything reaches the float64 scan.  A small threshold margin and an offset embedding (norms far larger than the
    neighbour distances) provoke thousands of rejections; the result must still be the brute-force one, bit for bit."""
    from scanpy_amd import _lib
    from scanpy_amd.datasets import blobs_embedding

    lib = _lib.load()
    n, k = 100_000, 15
    x, _ = blobs_embedding(n, 50, n_types=20, seed=11)
    x = x + np.float32(40.0)
    xd = _dev(x)
    monkeypatch.setenv("SCAMD_KNN_IVF", "0")
    i0, d0, _ = K.knn(xd, k)
    monkeypatch.setenv("SCAMD_KNN_IVF", "1")
    monkeypatch.setenv("SCAMD_KNN_THR_MARGIN", "2")
    monkeypatch.setenv("SCAMD_KNN_TIER2_MIN", "0")
    i1, d1, nf1 = K.knn(xd, k)
    t2 = int(lib.scamd_knn_last_second_tier_queries())
    print(f"second tier: {t2} queries re-done by the float32 engine, {nf1} left for the float64 scan")
    assert int(lib.scamd_knn_last_select_engine()) == 1 and t2 > 0 and nf1 < t2
    np.testing.assert_array_equal(i0.cpu().numpy(), i1.cpu().numpy())
    np.testing.assert_array_equal(d0.cpu().numpy(), d1.cpu().numpy())
    # every query through both tiers and the float64 scan (cert_scale = 1e30 rejects everything twice)
    m = 12000
    i2, d2, n2 = K.knn(xd[:m].contiguous(), k, cert_scale=1e30)
    i3, d3, _ = K.knn(xd[:m].contiguous(), k)
    assert n2 == m and int(lib.scamd_knn_last_select_engine()) == 1
    np.testing.assert_array_equal(i2.cpu().numpy(), i3.cpu().numpy())
    np.testing.assert_array_equal(d2.cpu().numpy(), d3.cpu().numpy())


def test_knn_quantiser_and_prepass_variants_give_the_same_lists(K, monkeypatch):
    """round 4: the quantiser assigns on the bf16 matrix cores and the threshold pre-pass keeps two minima per lane -- both
    only steer the pruning / the insertions.  The result must be bit for bit that of the float32 assignment and of the
    one-minimum pre-pass, and the bf16 assignment must not cost pruning (evaluated pairs within 5 %)."""
    from scanpy_amd import _lib
    from scanpy_amd.datasets import blobs_embedding

    lib = _lib.load()
    x, _ = blobs_embedding(150_000, 50, n_types=24, seed=19)
    xd = _dev(x)
    ref = None
    pairs = {}
    for assign, min2 in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("SCAMD_KNN_ASSIGN_MFMA", assign)
        monkeypatch.setenv("SCAMD_KNN_PREPASS_MIN2", min2)
        i, d, _ = K.knn(xd, 15)
        pairs[(assign, min2)] = float(lib.scamd_knn_last_select_pairs())
        assert int(lib.scamd_knn_last_select_engine()) == 1
        if ref is None:
            ref = (i.cpu().numpy(), d.cpu().numpy())
        else:
            np.testing.assert_array_equal(ref[0], i.cpu().numpy())
            np.testing.assert_array_equal(ref[1], d.cpu().numpy())
    print("evaluated pairs:", pairs)
    assert pairs[("1", "1")] <= 1.05 * pairs[("0", "1")]


def test_knn_forced_fallback_through_the_cell_scan(K, monkeypatch):
    """every query forced through the float64 fallback (cert_scale = 1e30) in cell-pruned mode: the fallback scans only
    the cells whose ball reaches the query's bound -- same lists as the certified run, bit for bit"""
    from scanpy_amd.datasets import blobs_embedding

    monkeypatch.setenv("SCAMD_KNN_IVF", "1")
    for kind in ("blobs", "offset", "noise"):
        n = 12000
        if kind == "noise":
            x = np.random.default_rng(3).standard_normal((n, 50)).astype(np.float32)
        else:
            x, _ = blobs_embedding(n, 50, n_types=9, seed=23)
            if kind == "offset":
                x = x + np.float32(120.0)
        xd = _dev(x)
        i1, d1, n1 = K.knn(xd, 15)
        i2, d2, n2 = K.knn(xd, 15, cert_scale=1e30)
        assert n2 == n and n1 < n // 10
        np.testing.assert_array_equal(i1.cpu().numpy(), i2.cpu().numpy())
        np.testing.assert_array_equal(d1.cpu().numpy(), d2.cpu().numpy())
