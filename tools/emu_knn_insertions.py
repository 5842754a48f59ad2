"""What does one query of the cell-pruned kNN sweep cost in list insertions?  (CPU only: the host emulation of the kernels.)

The hardware counters say the select kernel's matrix pipe is 30 % busy and its waves wait; they cannot say how many
survivors of the sign test a query meets.  The emulated kernel counts them (SCAMD_EMU_COUNT in csrc/knn.hip):

    python tools/emu_knn_insertions.py [n] [d] [n_clusters] [k]      # default 40000 50 40 15: isotropic blobs
    python tools/emu_knn_insertions.py 65536 planted                 # the bench's matrix (make_matrix), 50 PCs by sklearn

prints, per query: survivors of the sign test, insertions, (register, half) groups entered, and per wave the share of
32 x 32 sub-tiles that had a survivor at all -- with the default knobs and with the knobs given as KEY=VALUE arguments.
"""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "emu"))


def main():
    pos = [a for a in sys.argv[1:] if "=" not in a]
    env = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
    n = int(pos[0]) if len(pos) > 0 else 40_000
    planted = len(pos) > 1 and pos[1] in ("planted", "weak", "none")
    structure = pos[1] if planted else None
    if planted:
        pos[1] = "50"
    d = int(pos[1]) if len(pos) > 1 else 50
    n_c = int(pos[2]) if len(pos) > 2 else 40
    k = int(pos[3]) if len(pos) > 3 else 15
    os.environ.update(env)
    import harness

    lib = harness.load()
    if planted:
        import bench
        from sklearn.decomposition import PCA

        m, _ = bench.make_matrix(n, 2000, 0, structure)
        x = PCA(n_components=50, svd_solver="randomized", random_state=0).fit_transform(m.toarray()).astype(np.float32)
        n_c = structure
    else:
        rng = np.random.default_rng(0)
        centres = rng.normal(size=(n_c, d)) * 4.0
        x = (centres[rng.integers(0, n_c, n)] + rng.normal(size=(n, d))).astype(np.float32)
    lib.emu_reset_stats()
    t0 = time.perf_counter()
    out = harness.knn(lib, x, k)
    dt = time.perf_counter() - t0
    c = harness.user_counters(lib, 8)
    print(f"n={n} d={d} clusters={n_c} k={k} knobs={env} ({dt:.0f} s on the emulator; fallback/tier2 {out[2:] if len(out) > 2 else ''})")
    print(f"  per query: survivors {c[0] / n:.1f}, insertions {c[1] / n:.1f}, (register, half) groups entered {c[2] / n:.1f}")
    print(f"  sub-tiles per wave-of-32-queries: swept {c[3]} ({c[3] * 32 / n:.0f} per query), with a survivor {c[4]} "
          f"({100.0 * c[4] / max(c[3], 1):.1f} %), pre-pass {c[5]} ({c[5] * 32 / n:.0f} per query)")
    print(f"  candidates scored per query {c[3] * 32 * 32 / n:.0f} (+ pre-pass {c[5] * 32 * 32 / n:.0f})")


if __name__ == "__main__":
    main()
