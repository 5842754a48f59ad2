"""kNN-only runner for rocprofv3 PMC passes (not a test): one warm-up + `reps` launches of scamd_knn_l2_f32 on the
bench's OWN embedding (PCA 50 of the synthetic matrix bench.py times; `blobs` as 5th argument = the Gaussian blobs of
round 1), prints the select-kernel duration (HIP events inside the library) and the pairs it evaluated."""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from scanpy_amd import _kernels as K  # noqa: E402
from scanpy_amd import _lib  # noqa: E402
from scanpy_amd.datasets import blobs_embedding  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 15
    src = sys.argv[5] if len(sys.argv) > 5 else "planted"
    if src == "blobs":
        x, _ = blobs_embedding(n, d, seed=1)
        xd = torch.from_numpy(x).cuda()
    else:  # bench.py's workload: the embedding its PCA stage hands to the search
        import bench
        from scanpy_amd.preprocessing._pca_solver import GpuBackend, pca_fit

        m, _ = bench.make_matrix(n, 2000, 0, src)
        be = GpuBackend()
        xd = pca_fit(be.upload(m), d, backend=be).scores.contiguous()
        del m
    K.knn(xd[:8192].contiguous(), k)
    lib = _lib.load()
    for _ in range(reps):
        _, _, nfb = K.knn(xd, k)
        ms = float(lib.scamd_knn_last_select_ms())
        pairs, pre = float(lib.scamd_knn_last_select_pairs()), float(lib.scamd_knn_last_select_prepass_pairs())
        print(f"knn n={n} d={d} k={k} ({src}): select {ms:.2f} ms, swept pairs {pairs:.4e} (+ pre-pass {pre:.3e}) = "
              f"{2.0 * pairs * d / ms / 1e9:.1f} useful TFLOP/s, {2.0 * (pairs + pre) * d / ms / 1e9:.1f} executed; "
              f"brute-force equivalent {2.0 * n * n * d / ms / 1e9:.1f}; fallback={nfb} tier2={int(lib.scamd_knn_last_second_tier_queries())}", flush=True)


if __name__ == "__main__":
    main()
