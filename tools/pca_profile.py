"""Section timing of pca_fit on the GPU (not a test): wraps the backend methods and the dense helpers with
synchronising timers."""
from __future__ import annotations

import sys
import time
from collections import defaultdict
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from scanpy_amd.datasets import synthetic_planted  # noqa: E402
from scanpy_amd.preprocessing import _pca_solver as S  # noqa: E402

acc = defaultdict(float)
cnt = defaultdict(int)


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t
        cnt[name] += 1
        return r
    return w


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    x, _ = synthetic_planted(n, 2000, seed=0)
    be = S.GpuBackend()
    h = be.upload(x)
    S.pca_fit(h, 50, backend=be)  # warm-up
    for m in ("transpose", "row_stats", "spmm", "spmm_f64acc", "colsum"):
        setattr(be, m, timed(m, getattr(be, m)))
    S._orth = timed("_orth", S._orth)
    S._rayleigh_ritz = timed("_rayleigh_ritz", S._rayleigh_ritz)
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = S.pca_fit(h, 50, backend=be)
    torch.cuda.synchronize()
    total = time.perf_counter() - t
    print(f"total {total * 1e3:.1f} ms  info={res.info}")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {k:16s} {cnt[k]:3d} calls {v * 1e3:8.2f} ms")
    print(f"  other            {(total - sum(acc.values())) * 1e3:8.2f} ms")


if __name__ == "__main__":
    main()
