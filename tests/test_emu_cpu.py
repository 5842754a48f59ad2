"""Kernels of scanpy_amd/csrc executed on the HOST, lane by lane (tests/emu/README.md): test infrastructure, a second
library built from the same sources -- the product library and `scanpy_amd` are not involved.  Asserted here: the
entry points that run correctly under the emulator agree with the oracle, and no cross-lane operation was executed by
a partial wave."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "emu"))
sys.path.insert(0, str(ROOT))
from oracle import connectivities as oconn  # noqa: E402
from oracle import knn as oknn  # noqa: E402


@pytest.fixture(scope="module")
def emu():
    import build as emu_build
    import harness

    if not Path(emu_build.CLANG).exists():
        pytest.skip("no clang++ to build the host emulation of the kernels")
    return harness, harness.load()


def test_mfma_register_layouts(emu):
    H, lib = emu
    assert lib.scamd_selftest_mfma_layout(None) == 0


@pytest.mark.parametrize(("n", "k"), [(1500, 15), (400, 30), (300, 5), (500, 40)])
def test_fuzzy_set_against_the_oracle(emu, n, k):
    """k <= 16 / <= 32: record lookup (fss_recip_rec_kernel<16|32>), k = 40: the row-walking lookup; duplicates give
    zero distances"""
    H, lib = emu
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, 8)).astype(np.float32)
    x[: n // 20] = x[n // 20 : 2 * (n // 20)]
    idx, dist = oknn.knn_exact_f64(x, np.arange(n), k)
    lib.emu_reset_stats()
    ip, ix, dat, _, _ = H.fuzzy_simplicial_set(lib, idx, dist)
    ref, _, _ = oconn.fuzzy_simplicial_set(idx, dist, n, k)
    assert np.array_equal(ip, ref.indptr) and np.array_equal(ix, ref.indices)
    assert np.abs(dat - ref.data).max() <= 1e-6
    st = H.stats(lib)
    assert st["partial_collectives"] == st["mixed_collectives"] == st["reads_of_inactive_lanes"] == 0, st


def test_fuzzy_set_hub_rows(emu):
    """rows of the symmetric graph longer than the LDS block of fss_sortrows_kernel (one-wave register-chunk ranking)"""
    H, lib = emu
    rng = np.random.default_rng(11)
    n, k = 6000, 10
    idx = np.empty((n, k), dtype=np.int32)
    for i in range(n):
        others = rng.choice(n - 4, size=k - 2, replace=False) + 3
        others = others[others != i][: k - 2]
        while others.size < k - 2:
            c = int(rng.integers(3, n))
            if c != i and c not in others:
                others = np.append(others, c)
        hub = int(rng.integers(0, 2))
        idx[i] = [i, hub if hub != i else (hub + 1) % 2, *others]
    dist = np.sort(rng.random((n, k)).astype(np.float32) + 0.1, axis=1)
    dist[:, 0] = 0.0
    ip, ix, dat, _, _ = H.fuzzy_simplicial_set(lib, idx, dist)
    ref, _, _ = oconn.fuzzy_simplicial_set(idx, dist.astype(np.float64), n, k)
    assert np.diff(ip).max() > 2048, np.diff(ip).max()
    assert np.array_equal(ip, ref.indptr) and np.array_equal(ix, ref.indices)
    assert np.abs(dat - ref.data).max() <= 1e-6


def _blobs(n, d, n_c, seed, spread=4.0):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((n_c, d)) * spread
    return (cent[rng.integers(0, n_c, n)] + rng.standard_normal((n, d))).astype(np.float32)


@pytest.mark.parametrize(("n", "d", "env"), [(1500, 20, {}), (1500, 50, {}), (4500, 50, {"SCAMD_KNN_IVF": "1", "SCAMD_KNN_CELL_ROWS": "512"}),
                                             (1200, 50, {"SCAMD_KNN_B3": "0"})])
def test_knn_both_engines_brute_and_pruned(emu, monkeypatch, n, d, env):
    """float32 MFMA engine (d <= 32 or SCAMD_KNN_B3=0), bf16x3 engine (d in (32, 50]), cell-pruned sweep: index sets and
    float64 distances of a float64 brute force, no query left to the float64 fallback scan"""
    from oracle import compare as cmp

    H, lib = emu
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    x = _blobs(n, d, 8, n + d, spread=3.0)
    lib.emu_reset_stats()
    idx, dist, n_fallback = H.knn(lib, x, 15)
    ei, ed = oknn.knn_exact_f64(x, np.arange(n), 15)
    bad, _ = cmp.knn_rows_differing_beyond_ties(idx, dist, ei, ed)
    assert bad == 0 and n_fallback == 0
    assert np.max(np.abs(dist - ed) / np.maximum(ed, 1e-30)) < 1e-12
    assert lib.scamd_knn_last_select_engine() == (1 if d > 32 and env.get("SCAMD_KNN_B3") != "0" else 0)
    st = H.stats(lib)
    assert st["partial_collectives"] == st["mixed_collectives"] == st["reads_of_inactive_lanes"] == 0, st


@pytest.mark.parametrize("kind", ["blobs", "noise", "offset"])
def test_knn_coarse_first_stage_gives_the_same_lists(emu, monkeypatch, kind):
    """the COARSE sweep of the bf16 engine (round 6: hi.hi product first against a threshold widened by a bound on the two
    dropped products, the other eight MFMAs only for sub-tiles with a coarse survivor -- chosen by the host when the cell
    bounds prune little): the float64 brute force's lists, no query in the fallback, on clustered data, on data without
    structure (every cell swept: the regime it is for) and far from the origin (large norms = a large slack); both landing
    modes of the LDS-DMA ring; and the SAME lists as the plain kernel"""
    from oracle import compare as cmp

    H, lib = emu
    monkeypatch.setenv("SCAMD_KNN_IVF", "1")
    monkeypatch.setenv("SCAMD_KNN_CELL_ROWS", "512")
    n = 4300
    rng = np.random.default_rng(5)
    x = _blobs(n, 50, 8, 91, spread=3.0) if kind != "noise" else rng.standard_normal((n, 50)).astype(np.float32)
    if kind == "offset":
        x = (x + 40.0).astype(np.float32)
    ei, ed = oknn.knn_exact_f64(x, np.arange(n), 15)
    monkeypatch.setenv("SCAMD_KNN_COARSE", "0")
    idx0, dist0, _ = H.knn(lib, x, 15)
    assert lib.scamd_knn_last_coarse() == 0
    monkeypatch.setenv("SCAMD_KNN_COARSE", "1")
    try:
        for late in (1, 0):
            lib.emu_set_dma_late(late)
            lib.emu_reset_stats()
            idx, dist, n_fallback = H.knn(lib, x, 15)
            assert lib.scamd_knn_last_coarse() == 1 and lib.scamd_knn_last_select_engine() == 1
            bad, _ = cmp.knn_rows_differing_beyond_ties(idx, dist, ei, ed)
            assert bad == 0 and n_fallback == 0, (kind, late, bad, n_fallback)
            assert np.array_equal(np.sort(idx, axis=1), np.sort(idx0, axis=1)) and np.array_equal(np.sort(dist, axis=1), np.sort(dist0, axis=1))
            st = H.stats(lib)
            assert st["partial_collectives"] == st["mixed_collectives"] == st["reads_of_inactive_lanes"] == 0, st
    finally:
        lib.emu_set_dma_late(0)


@pytest.mark.parametrize("ivf", ["1", "0"])
def test_knn_lds_dma_ring_with_late_landing(emu, monkeypatch, ivf):
    """the bf16 engine's tile ring (three LDS buffers filled by LDS-DMA, two requests in flight across the barrier): the
    emulator lands every request as LATE as the kernel's counted waits allow (a read placed before its wait sees the old
    bytes) -- the default run of the other tests lands them at once (a request into a buffer still being read corrupts it).
    Both must give the float64 brute force; cells of 512 rows and a 4-entry order table make the blocks cross many cells"""
    from oracle import compare as cmp

    H, lib = emu
    monkeypatch.setenv("SCAMD_KNN_IVF", ivf)
    monkeypatch.setenv("SCAMD_KNN_CELL_ROWS", "512")
    n = 4500 if ivf == "1" else 1700
    x = _blobs(n, 50, 8, 77, spread=3.0)
    ei, ed = oknn.knn_exact_f64(x, np.arange(n), 15)
    try:
        for late in (1, 0):
            lib.emu_set_dma_late(late)
            idx, dist, n_fallback = H.knn(lib, x, 15)
            assert lib.scamd_knn_last_select_engine() == 1
            bad, _ = cmp.knn_rows_differing_beyond_ties(idx, dist, ei, ed)
            assert bad == 0 and n_fallback == 0, (late, bad, n_fallback)
    finally:
        lib.emu_set_dma_late(0)


def test_knn_persistent_launch_takes_every_block_once(emu, monkeypatch):
    """the pruned sweep's persistent launch (knn_select_reg_kernel: a fixed number of workgroups take blocks off the eight
    per-XCD queues and steal from the others once their own is dry): forced at test size with SCAMD_KNN_PERSISTENT=8
    (workgroups) on a layout with >= 64 blocks (the XCD-aware queues) and on one below (block-id order) -- same lists, same
    survivor / insertion counts as the launch of one workgroup per slot, every query answered"""
    from oracle import compare as cmp

    H, lib = emu
    monkeypatch.setenv("SCAMD_KNN_IVF", "1")
    monkeypatch.setenv("SCAMD_KNN_CELL_ROWS", "512")
    for n, launches in ((8500, ("0", "8")), (2500, ("0", "8"))):
        x = _blobs(n, 50, 6, 5, spread=3.0)
        ei, ed = oknn.knn_exact_f64(x, np.arange(n), 15)
        ref = None
        for groups in launches:
            monkeypatch.setenv("SCAMD_KNN_PERSISTENT", groups)
            before = H.user_counters(lib, 2)
            idx, dist, n_fallback = H.knn(lib, x, 15)
            after = H.user_counters(lib, 2)
            bad, _ = cmp.knn_rows_differing_beyond_ties(idx, dist, ei, ed)
            assert bad == 0 and n_fallback == 0, (n, groups, bad, n_fallback)
            got = (idx.tobytes(), dist.tobytes(), after[0] - before[0], after[1] - before[1])
            ref = ref or got
            assert got == ref, (n, groups)


def test_knn_trace_dump_matches_its_parser(emu, monkeypatch, tmp_path):
    """SCAMD_KNN_TRACE: the per-block records the pruned sweep dumps and `tools/knn_trace.py` reads (the measurement behind
    DESIGN 3.1 "What binds it") -- record size, tiles swept == the launch's own count of evaluated pairs, cells per block"""
    sys.path.insert(0, str(ROOT / "tools"))
    import knn_trace

    H, lib = emu
    path = tmp_path / "trace.bin"
    monkeypatch.setenv("SCAMD_KNN_IVF", "1")
    monkeypatch.setenv("SCAMD_KNN_CELL_ROWS", "512")
    monkeypatch.setenv("SCAMD_KNN_TRACE", str(path))
    x = _blobs(4500, 50, 8, 5, spread=3.0)
    H.knn(lib, x, 15)
    tr, cell = knn_trace.parse_trace(path)
    lib.scamd_knn_last_select_pairs.restype = C.c_double
    assert tr.shape[0] == cell.size and tr.shape[0] >= 4500 // 128
    assert int(tr[:, 2].sum()) * 64 * 128 == int(lib.scamd_knn_last_select_pairs())  # tiles x 64 candidates x 128 queries
    assert (tr[:, 7] >= 1).all() and (tr[:, 7] <= 16).all() and cell.min() >= 0 and cell.max() < 16


def test_knn_float64_fallback_scan(emu):
    """cert_scale = 1e30: no query can be certified, every one goes through the float64 scan"""
    from oracle import compare as cmp

    H, lib = emu
    x = _blobs(700, 20, 4, 3)
    idx, dist, n_fallback = H.knn(lib, x, 10, cert_scale=1e30)
    ei, ed = oknn.knn_exact_f64(x, np.arange(700), 10)
    assert n_fallback == 700
    assert cmp.knn_rows_differing_beyond_ties(idx, dist, ei, ed)[0] == 0


@pytest.mark.parametrize("small", ["1", "0"])
def test_leiden_reaches_the_oracles_modularity(emu, monkeypatch, small):
    """every Leiden kernel on the host (n = 1500: two levels through the big kernels, then -- SCAMD_LEIDEN_SMALL=1 -- the
    one-workgroup small levels): reported modularity == modularity of the labels, not below the oracle's own run, identical
    on repetition, no cross-lane operation by a partial wave"""
    from oracle import leiden as ol

    H, lib = emu
    monkeypatch.setenv("SCAMD_LEIDEN_SMALL", small)
    x = _blobs(1500, 10, 12, 0)
    idx, dist = oknn.knn_exact_f64(x, np.arange(1500), 15)
    conn, _, _ = oconn.fuzzy_simplicial_set(idx, dist, 1500, 15)
    lib.emu_reset_stats()
    memb, q, nc = H.leiden(lib, conn, seed=0)
    assert abs(q - ol.modularity(conn, memb)) < 1e-9 and nc == int(memb.max()) + 1
    assert q > ol.leiden(conn, seed=0)[1] - 2e-3
    memb2, q2, _ = H.leiden(lib, conn, seed=0)
    assert q2 == q and np.array_equal(memb, memb2)
    assert abs(H.modularity(lib, conn, memb) - q) < 1e-12
    st = H.stats(lib)
    assert st["partial_collectives"] == st["mixed_collectives"] == st["reads_of_inactive_lanes"] == 0, st
    # the paper's guarantees for a stable partition (oracle/leiden_guarantees.py): no vertex move, no merge improves it
    from oracle import leiden_guarantees as lg

    assert lg.improving_moves(conn, memb)["count"] == 0 and lg.mergeable_pairs(conn, memb)["count"] == 0


def test_leiden_polish_finishes_an_unfinished_partition(emu, monkeypatch):
    """n_iterations = -1 promises a stable partition.  With the outer loop cut after two iterations (test knob) the best
    partition still has improvable vertices (SCAMD_LEIDEN_POLISH=0: the round-4 behaviour); the final polish -- monotone,
    lock-arbitrated single-vertex moves, then a verifying iteration -- leaves none, merges nothing that should not be,
    and only ever raises the quality"""
    from oracle import leiden as ol
    from oracle import leiden_guarantees as lg

    H, lib = emu
    n = 2500
    x = np.random.default_rng(5).standard_normal((n, 10)).astype(np.float32)
    idx, dist = oknn.knn_exact_f64(x, np.arange(n), 15)
    conn, _, _ = oconn.fuzzy_simplicial_set(idx, dist, n, 15)
    monkeypatch.setenv("SCAMD_LEIDEN_MAX_ITERS", "2")
    monkeypatch.setenv("SCAMD_LEIDEN_POLISH", "0")
    memb0, q0, _ = H.leiden(lib, conn, seed=0)
    assert lg.improving_moves(conn, memb0)["count"] > 0 and H.leiden_stats(lib)["polish_rounds"] == 0
    monkeypatch.setenv("SCAMD_LEIDEN_POLISH", "1")
    lib.emu_reset_stats()
    memb, q, nc = H.leiden(lib, conn, seed=0)
    st = H.leiden_stats(lib)
    assert st["polish_moves"] > 0 and st["polish_full_sweeps"] >= 2 and st["iterations"] > 2, st
    assert q > q0 and abs(q - ol.modularity(conn, memb)) < 1e-9 and nc == int(memb.max()) + 1
    assert lg.improving_moves(conn, memb)["count"] == 0 and lg.mergeable_pairs(conn, memb)["count"] == 0
    es = H.stats(lib)
    assert es["partial_collectives"] == es["mixed_collectives"] == es["reads_of_inactive_lanes"] == 0, es
    # a finite n_iterations promises nothing of the kind and is left alone
    H.leiden(lib, conn, seed=0, n_iterations=2)
    assert H.leiden_stats(lib)["polish_rounds"] == 0


def test_knn_approximate_mode(emu, monkeypatch):
    """scamd_knn_l2_ivf_f32: self first, exact float64 distances of whatever it returns, recall rising with nprobe, and the
    exact lists once every cell is probed"""
    from oracle import compare as cmp

    H, lib = emu
    n, d, k = 4096, 50, 15
    rng = np.random.default_rng(0)
    cent = rng.standard_normal((12, d))
    x = (cent[rng.integers(0, 12, n)] + rng.standard_normal((n, d))).astype(np.float32)  # overlapping clusters
    monkeypatch.setenv("SCAMD_KNN_CELL_ROWS", "256")  # 16 cells
    ei, ed = oknn.knn_exact_f64(x, np.arange(n), k)
    recall = {}
    for nprobe in (1, 3, 16):
        idx, dist, _ = H.knn(lib, x, k, nprobe=nprobe)
        assert np.array_equal(idx[:, 0], np.arange(n)) and not dist[:, 0].any() and (np.diff(dist, axis=1) >= 0).all()
        dd = np.sqrt(((x[idx[:, 1:]].astype(np.float64) - x[:, None, :].astype(np.float64)) ** 2).sum(-1))
        assert np.abs(dd - dist[:, 1:]).max() < 1e-12
        recall[nprobe] = float((idx[:, 1:, None] == ei[:, None, 1:]).any(1).mean())
        if nprobe == 16:
            assert cmp.knn_rows_differing_beyond_ties(idx, dist, ei, ed)[0] == 0
        else:
            assert lib.scamd_knn_last_select_pairs() < 0.5 * n * n
    assert 0.5 < recall[1] < recall[3] <= recall[16] == 1.0, recall


def test_leiden_component_split(emu):
    """the component split the polish applies after its moves (scamd_leiden_debug_split_f32): on a sparse random graph
    under a coarse random labelling -- thousands of disconnected pieces -- the result is scipy's components of the
    intra-label graph, ids = smallest member; a connected labelling is left alone"""
    from scipy import sparse
    from scipy.sparse.csgraph import connected_components

    H, lib = emu
    rng = np.random.default_rng(0)
    n = 3000
    r, c = rng.integers(0, n, 2600), rng.integers(0, n, 2600)
    a = sparse.coo_matrix((rng.random(2600).astype(np.float32) + 0.1, (r, c)), shape=(n, n)).tocsr()
    a = a.maximum(a.T).tocsr()
    a.setdiag(0)
    a.eliminate_zeros()
    lab = (rng.integers(0, 7, n) * 11).astype(np.int32)
    got, n_split = H.leiden_split(lib, a, lab)
    same = lab[np.repeat(np.arange(n), np.diff(a.indptr))] == lab[a.indices]
    inner = sparse.csr_matrix((same.astype(np.int8), a.indices.copy(), a.indptr.copy()), shape=a.shape)
    inner.eliminate_zeros()
    n_comp, comp = connected_components(inner, directed=False)
    assert n_split == n_comp - np.unique(lab).size > 1000
    first = np.full(n_comp, n)
    np.minimum.at(first, comp, np.arange(n))
    assert np.array_equal(got, first[comp])
    again, n_split2 = H.leiden_split(lib, a, got)
    assert n_split2 == 0 and np.array_equal(again, got)


@pytest.mark.parametrize(("n", "d", "k"), [(700, 150, 15), (600, 256, 40), (500, 40, 200), (520, 200, 256), (300, 129, 121)])
def test_knn_wide_rows_and_long_lists(emu, n, d, k):
    """d in (128, 256] (knn_select_kernel<128, 32, ...>) and k in (120, 256] (lists of 288 in LDS, one wave per block):
    the shapes the reference takes without a limit (src/scanpy/neighbors/__init__.py:88-103); duplicates included"""
    from oracle import compare as cmp

    H, lib = emu
    rng = np.random.default_rng(n + d + k)
    x = (rng.standard_normal((n, d)) + 3.0).astype(np.float32)
    x[: n // 10] = x[n // 10: 2 * (n // 10)]
    idx, dist, n_scan = H.knn(lib, x, k)
    ei, ed = oknn.knn_exact_f64(x, np.arange(n), k)
    assert cmp.knn_rows_differing_beyond_ties(idx, dist, ei, ed)[0] == 0
    assert n_scan <= n // 10  # the MFMA pass certifies (nearly) every query itself: the float64 scan is the exception


def test_leiden_coarse_row_builders_agree(emu, monkeypatch):
    """the coarse graph does not depend on which builder made a row: wave tier, workgroup tiers by table size or by work,
    one optimistic pass or the class passes after a failed trial, whole rows or rows cut into parts and merged"""
    H, lib = emu
    n = 3000
    x = np.random.default_rng(0).standard_normal((n, 10)).astype(np.float32)
    idx, dist = oknn.knn_exact_f64(x, np.arange(n), 15)
    conn, _, _ = oconn.fuzzy_simplicial_set(idx, dist, n, 15)
    monkeypatch.setenv("SCAMD_LEIDEN_SMALL", "0")
    base = H.leiden(lib, conn, seed=0)
    for env in ({"SCAMD_LEIDEN_AGG_WAVE_WORK": "64", "SCAMD_LEIDEN_AGG_MID_WORK": "400"},
                {"SCAMD_LEIDEN_AGG_WAVE_MAX": "0"},
                {"SCAMD_LEIDEN_AGG_WAVE_MAX": "0", "SCAMD_LEIDEN_AGG_MID_MAX": "0", "SCAMD_LEIDEN_AGG_PASS_KEYS": "16",
                 "SCAMD_LEIDEN_HUB_TRY_PROBES": "1"},
                # ~200 coarse rows per level cut into parts of 64 member entries, built as pseudo rows and merged
                {"SCAMD_LEIDEN_AGG_WAVE_WORK": "32", "SCAMD_LEIDEN_AGG_SPLIT_CHUNK": "64", "SCAMD_LEIDEN_AGG_SPLIT_WORK": "64"}):
        with monkeypatch.context() as mp:
            for k_, v_ in env.items():
                mp.setenv(k_, v_)
            got = H.leiden(lib, conn, seed=0)
        assert got[1] == base[1] and np.array_equal(got[0], base[0]), env


def test_leiden_cpm_objective(emu):
    """igraph's `objective_function='CPM'` (src/scanpy/tools/_leiden.py:188-196): vertex weights 1, resolution not normalised.
    At a resolution between the planted clusters' internal and external densities the clusters are recovered; at every
    resolution the partition's CPM quality is at least the planted one's, no vertex move and no merge improves it UNDER THE CPM
    OBJECTIVE (oracle/leiden_guarantees.py, objective='cpm'), and what is reported is the partition's modularity"""
    from sklearn.metrics import adjusted_rand_score

    from oracle import leiden as ol
    from oracle import leiden_guarantees as lg

    H, lib = emu
    rng = np.random.default_rng(0)
    n = 1500
    cent = rng.standard_normal((12, 10)) * 4
    truth = rng.integers(0, 12, n)
    x = (cent[truth] + rng.standard_normal((n, 10))).astype(np.float32)
    idx, dist = oknn.knn_exact_f64(x, np.arange(n), 15)
    conn, _, _ = oconn.fuzzy_simplicial_set(idx, dist, n, 15)
    for gamma, recovered in ((0.01, True), (0.05, False)):
        memb, q, nc = H.leiden(lib, conn, seed=0, resolution=gamma, objective=1)
        assert abs(q - ol.modularity(conn, memb)) < 1e-9 and nc == int(memb.max()) + 1
        q_cpm = lg.quality(conn, memb, resolution=gamma, objective="cpm")
        assert q_cpm >= lg.quality(conn, truth, resolution=gamma, objective="cpm") - 1e-12
        assert q_cpm > max(lg.quality(conn, np.arange(n), resolution=gamma, objective="cpm"),
                           lg.quality(conn, np.zeros(n, dtype=int), resolution=gamma, objective="cpm"))
        assert lg.improving_moves(conn, memb, resolution=gamma, objective="cpm")["count"] == 0
        assert lg.mergeable_pairs(conn, memb, resolution=gamma, objective="cpm")["count"] == 0
        assert (adjusted_rand_score(memb, truth) == 1.0) == recovered and (nc == 12) == recovered
    again = H.leiden(lib, conn, seed=0, resolution=0.05, objective=1)
    assert np.array_equal(again[0], memb)  # reproducible
    # the modularity objective is untouched by the switch
    m0 = H.leiden(lib, conn, seed=0)
    m1 = H.leiden(lib, conn, seed=0, objective=0)
    assert m0[1] == m1[1] and np.array_equal(m0[0], m1[0])


def test_leiden_cpm_node_weights(emu):
    """igraph's `node_weights` under the CPM objective (`sc.tl.leiden(flavor='igraph', objective_function='CPM',
    node_weights=...)`, `**clustering_args` at src/scanpy/tools/_leiden.py:66, 188-196): a community pays resolution x (sum of
    its members' weights)^2.  Weights that are all 1 give the partition of the unweighted call bit for bit (the fixed-point
    scale is a power of two); with weights in sixteenths (exact in the kernel's 16 fractional bits) no vertex move and no
    merge improves the WEIGHTED quality, which is at least the planted partition's; heavy vertices end in smaller
    communities; bad weights and the modularity objective are refused"""
    from oracle import leiden_guarantees as lg

    H, lib = emu
    rng = np.random.default_rng(0)
    n = 1500
    cent = rng.standard_normal((12, 10)) * 4
    truth = rng.integers(0, 12, n)
    x = (cent[truth] + rng.standard_normal((n, 10))).astype(np.float32)
    idx, dist = oknn.knn_exact_f64(x, np.arange(n), 15)
    conn, _, _ = oconn.fuzzy_simplicial_set(idx, dist, n, 15)
    gamma = 0.01
    plain = H.leiden(lib, conn, seed=0, resolution=gamma, objective=1)
    ones = H.leiden(lib, conn, seed=0, resolution=gamma, objective=1, node_weights=np.ones(n))
    assert np.array_equal(plain[0], ones[0]) and plain[1] == ones[1]
    nw = rng.integers(4, 65, n) / 16.0  # 0.25 .. 4 in sixteenths
    nw[truth == 3] *= 4.0               # one planted cluster of heavy vertices
    memb, q, nc = H.leiden(lib, conn, seed=0, resolution=gamma, objective=1, node_weights=nw)
    kw = dict(resolution=gamma, objective="cpm", node_weights=nw)
    assert lg.improving_moves(conn, memb, **kw)["count"] == 0 and lg.mergeable_pairs(conn, memb, **kw)["count"] == 0
    assert lg.quality(conn, memb, **kw) >= lg.quality(conn, truth, **kw) - 1e-12
    assert nc > plain[2]  # the heavy cluster cannot stay whole at this resolution
    heavy = np.unique(memb[truth == 3]).size
    assert heavy > 1 and all(np.unique(memb[truth == c]).size <= heavy for c in range(12))
    with pytest.raises(RuntimeError, match=r"node weights must lie in \[0, 1e6\]"):
        H.leiden(lib, conn, objective=1, node_weights=-np.ones(n))
    with pytest.raises(RuntimeError, match="node weights with the modularity objective"):
        H.leiden(lib, conn, objective=0, node_weights=np.ones(n))


def test_leiden_hub_rows(emu):
    """a vertex with 2500 neighbours (multi-pass hub tables) and vertices of 150 .. 1200 (overflow list, hub list tiers)"""
    from scipy import sparse

    from oracle import leiden as ol

    H, lib = emu
    rng = np.random.default_rng(2)
    n, deg = 3000, 8
    m = sparse.coo_matrix((rng.random(n * deg).astype(np.float32) * 0.9 + 0.1, (np.repeat(np.arange(n), deg), rng.integers(0, n, n * deg))),
                          shape=(n, n)).tocsr()
    for h, dh in ((0, 150), (1, 250), (2, 500), (3, 1200), (4, 2500)):
        t = rng.choice(n, dh, replace=False)
        m = m + sparse.coo_matrix((rng.random(dh).astype(np.float32) * 0.5 + 0.1, (np.full(dh, h), t)), shape=(n, n)).tocsr()
    m.setdiag(0)
    m.eliminate_zeros()
    m = m.maximum(m.T).tocsr().astype(np.float32)
    memb, q, _ = H.leiden(lib, m, seed=0)
    assert abs(q - ol.modularity(m, memb)) < 1e-9
    assert q > ol.leiden(m, seed=0)[1] - 0.01


def test_pca_chain(emu):
    """exact fixed-point Gram (bit for bit), then scamd_pca_csr_f32 (float64 MFMA GEMM, CholeskyQR2, Jacobi) against
    sklearn PCA(arpack)"""
    import bench
    from oracle import compare as cmp
    from oracle import pca as opca

    H, lib = emu
    n, g, k = 900, 200, 12
    x, _ = bench.make_matrix(n, g, 0, "planted")
    gram, colsum, sb, _ = H.csr_gram(lib, x)
    xd = np.asarray(x.todense(), dtype=np.float64)
    ref = np.zeros((g, g), dtype=np.int64)
    for r in range(n):
        ref += np.rint(np.outer(xd[r], xd[r]) * 2.0**sb).astype(np.int64)
    assert np.array_equal(gram[:g, :g], ref) and not gram[g:].any() and not gram[:, g:].any()
    assert np.array_equal(colsum[:g], np.rint(xd * 2.0**sb).astype(np.int64).sum(0))
    out = H.pca_csr(lib, x, k)
    r = opca.pca_reference(x, k)
    assert cmp.pca_loading_err(out["components"], r["components"]) < 1e-4
    assert np.abs(out["variance"] - r["variance"]).max() / r["variance"][0] < 1e-5
    assert np.abs(np.abs(out["scores"]) - np.abs(r["X_pca"])).max() < 1e-3


def test_knn_second_tier(emu, monkeypatch):
    """pruned sweep on the bf16 engine with a widened certificate (cert_scale 10): ~1000 of 5000 queries are rejected by
    the first tier and re-done by the float32 engine, a handful reach the float64 scan; the lists stay the brute-force ones"""
    from scanpy_amd.datasets import blobs_embedding

    H, lib = emu
    n, k = 5000, 15
    x, _ = blobs_embedding(n, 50, n_types=12, seed=11)
    x = (x + np.float32(40.0)).astype(np.float32)
    monkeypatch.setenv("SCAMD_KNN_IVF", "0")
    i0, d0, _ = H.knn(lib, x, k)
    for k_, v_ in (("SCAMD_KNN_IVF", "1"), ("SCAMD_KNN_CELL_ROWS", "512"), ("SCAMD_KNN_THR_MARGIN", "2"), ("SCAMD_KNN_TIER2_MIN", "0")):
        monkeypatch.setenv(k_, v_)
    i1, d1, n_scan = H.knn(lib, x, k, cert_scale=10.0)
    t2 = int(lib.scamd_knn_last_second_tier_queries())
    assert int(lib.scamd_knn_last_select_engine()) == 1 and t2 > 100 and n_scan < t2, (t2, n_scan)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)


def test_leiden_tiny_graphs(emu):
    """levels of <= 16 vertices move one vertex at a time: a single edge must end as ONE community (eight vertices deciding
    at once made its two ends swap communities for ever: two singletons, Q = -0.5), nothing ends below Q = 0"""
    from scipy import sparse

    from oracle import leiden as ol

    H, lib = emu
    rng = np.random.default_rng(1)
    graphs = [sparse.csr_matrix(np.array([[0, 1], [1, 0]], dtype=np.float32)),
              sparse.csr_matrix(np.array([[0, 1, 0], [1, 0, 0], [0, 0, 0]], dtype=np.float32))]
    while len(graphs) < 25:
        n = int(rng.integers(3, 24))
        a = (rng.random((n, n)) < rng.choice([0.1, 0.3, 0.6, 1.0])).astype(np.float32) * rng.random((n, n)).astype(np.float32)
        a = np.triu(a, 1)
        if a.sum() > 0:
            graphs.append(sparse.csr_matrix(a + a.T))
    memb, q, nc = H.leiden(lib, graphs[0], seed=0)
    assert nc == 1 and q == 0.0
    for g in graphs:
        for seed in (0, 1):
            memb, q, _ = H.leiden(lib, g, seed=seed)
            assert q > -1e-12 and abs(q - ol.modularity(g, memb)) < 1e-9
            assert q > min(ol.leiden(g, seed=s)[1] for s in range(3)) - 0.05


def test_front_ends_and_widening_kernels_on_the_emulator(emu):
    """A subset of the `-m gpu` tests themselves, run in a child process against the emulated library
    (SCAMD_TESTS_ON_EMULATOR=1, tests/emu/patch_torch.py): the drop-in front ends (sc.pp.pca goldens, sc.pp.neighbors
    options incl. the transformer plug-in route and gauss / jaccard, sc.tl.leiden parameters and errors), the
    normalize / log1p / HVG / scale kernels against the reference's goldens and the UMAP layout kernel against its
    synchronous restatement -- the cases a lane-by-lane executor finishes in a minute."""
    import os
    import re
    import subprocess

    keep = ("pca_transform_golden or pca_no_zero_center_golden or pca_shapes_and_errors or neighbors_key_added or leiden_errors or "
            "leiden_initial_membership or leiden_cpm_objective or "
            "neighbors_precomputed_distances or neighbors_fixture_vs_oracle or gauss_and_jaccard or neighbors_cosine_metric or "
            "transformer_plugin_route or leiden_restrict_to or leiden_basic_and_params or normalize_total or rep_mutation or "
            "test_scale or test_filters or chain_goldens or random_against_oracle or col_stats_clip or hvg_ or global-atomics or "
            "kernel_matches_synchronous_oracle or device_pruning or empty_matrix_and_bad_arguments")
    env = dict(os.environ, SCAMD_TESTS_ON_EMULATOR="1")
    out = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_pipeline.py"),
                          str(ROOT / "tests" / "test_gpu_preprocess.py"), str(ROOT / "tests" / "test_gpu_umap.py"), "-m", "gpu", "-q",
                          "-p", "no:cacheprovider", "-k", keep], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-1500:]
    assert out.returncode == 0, tail
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 42, tail
