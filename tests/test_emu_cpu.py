"""Kernels of scanpy_amd/csrc executed on the HOST, lane by lane (tests/emu/README.md): test infrastructure, a second
library built from the same sources -- the product library and `scanpy_amd` are not involved.  Asserted here: the
entry points that run correctly under the emulator agree with the oracle, and no cross-lane operation was executed by
a partial wave."""
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
