"""kNN-only runner for rocprofv3 PMC passes (not a test): one warm-up + `reps` launches of scamd_knn_l2_f32 on a
blobs embedding, prints the select-kernel duration (HIP events inside the library)."""
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
    x, _ = blobs_embedding(n, d, seed=1)
    xd = torch.from_numpy(x).cuda()
    K.knn(xd[:8192].contiguous(), k)
    lib = _lib.load()
    for _ in range(reps):
        _, _, nfb = K.knn(xd, k)
        ms = float(lib.scamd_knn_last_select_ms())
        print(f"knn n={n} d={d} k={k}: select {ms:.2f} ms  {2.0 * n * n * d / ms / 1e9:.1f} TFLOP/s  fallback={nfb}", flush=True)


if __name__ == "__main__":
    main()
