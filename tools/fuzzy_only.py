"""Connectivity-stage runner for rocprofv3 passes (not a test): exact kNN (k = 15) of the Gaussian blobs embedding, then
`reps` launches of scamd_fuzzy_simplicial_set_f32 on it; prints the wall time of one call."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from scanpy_amd import _kernels as K  # noqa: E402
from scanpy_amd.datasets import blobs_embedding  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    x, _ = blobs_embedding(n, 50, seed=1)
    idx, dist, _ = K.knn(torch.from_numpy(x).cuda(), 15)
    d32 = dist.to(torch.float32)
    K.fuzzy_simplicial_set(idx, d32)
    torch.cuda.synchronize()
    for _ in range(reps):
        t0 = time.perf_counter()
        out = K.fuzzy_simplicial_set(idx, d32)
        torch.cuda.synchronize()
        print(f"fuzzy n={n}: {(time.perf_counter() - t0) * 1e3:.2f} ms, nnz={int(out[0][-1])}", flush=True)


if __name__ == "__main__":
    main()
