"""Per-round trace of the GPU Leiden (SCAMD_LEIDEN_DEBUG=1) on the path's own graph of a synthetic matrix.
    python tools/leiden_trace.py 1000000 none > trace.log 2>&1
Prints the stage times first, then the library's round-by-round trace (stderr)."""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    structure = sys.argv[2] if len(sys.argv) > 2 else "none"
    import torch

    import bench
    from scanpy_amd import _kernels as K
    from scanpy_amd._pipeline import run_path
    from scanpy_amd.preprocessing._pca_solver import GpuBackend

    x, truth = bench.make_matrix(n, 2000, 0, structure)
    backend = GpuBackend()
    h = backend.upload(x)
    os.environ.pop("SCAMD_LEIDEN_DEBUG", None)
    res = run_path(h, n, backend=backend, timing=True)
    print("stage_ms", res.stage_ms, "nc", res.n_communities, "Q", res.modularity, flush=True)
    os.environ["SCAMD_LEIDEN_DEBUG"] = "1"  # (read by the library on every call)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    labels, q, nc = K.leiden(res.conn_indptr, res.conn_indices, res.conn_data, n)
    torch.cuda.synchronize()
    print(f"leiden again: {(time.perf_counter() - t0) * 1e3:.1f} ms, nc {nc}, Q {q}", flush=True)
    if structure != "none":
        from sklearn.metrics import adjusted_rand_score

        print("ARI vs truth", adjusted_rand_score(truth, labels.cpu().numpy()))


if __name__ == "__main__":
    main()
