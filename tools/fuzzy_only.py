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
    src = sys.argv[3] if len(sys.argv) > 3 else "blobs"
    if src == "blobs":
        x, _ = blobs_embedding(n, 50, seed=1)
        xd = torch.from_numpy(x).cuda()
    else:  # the bench's own embedding (PCA 50 of its synthetic matrix)
        import bench
        from scanpy_amd.preprocessing._pca_solver import GpuBackend, pca_fit

        m, _ = bench.make_matrix(n, 2000, 0, src)
        be = GpuBackend()
        xd = pca_fit(be.upload(m), 50, backend=be).scores.contiguous()
        del m
    idx, dist, _ = K.knn(xd, 15)
    d32 = dist.to(torch.float32)
    K.fuzzy_simplicial_set(idx, d32)
    torch.cuda.synchronize()
    for _ in range(reps):
        t0 = time.perf_counter()
        out = K.fuzzy_simplicial_set(idx, d32)
        torch.cuda.synchronize()
        print(f"fuzzy n={n}: {(time.perf_counter() - t0) * 1e3:.2f} ms, nnz={int(out[0][-1])}", flush=True)
    lens = (out[0][1:] - out[0][:-1]).cpu()
    print(f"row lengths: max {int(lens.max())}, > 64: {int((lens > 64).sum())}, > 512: {int((lens > 512).sum())}, "
          f"> 2048: {int((lens > 2048).sum())}, sum of squares of the rows > 64: {float((lens[lens > 64].double() ** 2).sum()):.3e}", flush=True)


if __name__ == "__main__":
    main()
