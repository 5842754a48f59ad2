"""Leiden alone on the path's own fuzzy graph (for rocprofv3 --kernel-trace --stats and A/B timing of env knobs).
    python tools/leiden_only.py 1000000 planted 5"""
from __future__ import annotations

import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    structure = sys.argv[2] if len(sys.argv) > 2 else "planted"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    import torch

    import bench
    from scanpy_amd import _kernels as K
    from scanpy_amd._pipeline import run_path
    from scanpy_amd.preprocessing._pca_solver import GpuBackend

    x, truth = bench.make_matrix(n, 2000, 0, structure)
    backend = GpuBackend()
    res = run_path(backend.upload(x), n, backend=backend)
    ip, ix, w = res.conn_indptr, res.conn_indices, res.conn_data
    K.leiden(ip, ix, w, n)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        labels, q, nc = K.leiden(ip, ix, w, n)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"leiden n={n} {structure}: best {min(ts):.2f} ms, mean {sum(ts) / len(ts):.2f} ms over {reps}; nc {nc} Q {q!r}; "
          f"{K.leiden_last_stats()}", flush=True)


if __name__ == "__main__":
    main()
