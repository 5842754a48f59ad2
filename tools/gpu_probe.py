"""Bring-up timing probe (not a test): per-kernel timings with HIP events.  Writes gpurun_out/probe.log."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from scanpy_amd import _kernels as K  # noqa: E402
from scanpy_amd.datasets import blobs_embedding  # noqa: E402


def timed(fn, reps=1):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return out, s.elapsed_time(e) / reps


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [100_000]
    for n in sizes:
        x, _ = blobs_embedding(n, 50, seed=1)
        xd = torch.from_numpy(x).cuda()
        K.knn(xd[:4096].contiguous(), 15)  # warm up / module load
        (idx, dist, nfb), ms = timed(lambda: K.knn(xd, 15))
        flops = 2.0 * n * n * 50
        print(f"knn n={n} k=15: {ms:.1f} ms  {flops / ms / 1e9:.1f} TFLOP/s (of 157.3)  fallback={nfb}", flush=True)
        (res), ms2 = timed(lambda: K.fuzzy_simplicial_set(idx, dist.float()))
        print(f"fuzzy n={n}: {ms2:.2f} ms nnz={res[1].numel()}", flush=True)


if __name__ == "__main__":
    main()
