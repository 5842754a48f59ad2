"""Host-side cost of `pp.neighbors` / `tl.leiden` with the device work removed: the kernel layer is replaced by
functions that hand back pre-built tensors at once, so what is timed is the scipy / pandas / numpy slot construction
around the kernels (runs on a machine without a GPU).  usage: host_overhead_probe.py [n_obs] [--profile]"""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

import scanpy_amd as sc
from scanpy_amd import _device, _kernels

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_000_000
k = 15
rng = np.random.default_rng(0)
idx = rng.integers(0, n, (n, k), dtype=np.int32)
idx[:, 0] = np.arange(n)
dist = np.sort(rng.random((n, k)), axis=1)
dist[:, 0] = 0
t_idx, t_dist = torch.from_numpy(idx), torch.from_numpy(dist)
nnz = n * 22
indptr = torch.from_numpy(np.arange(n + 1, dtype=np.int64) * 22)
indices = torch.from_numpy(np.sort(rng.integers(0, n, (n, 22), dtype=np.int32), axis=1).reshape(-1))
data = torch.from_numpy(rng.random(nnz, dtype=np.float32))
labels = torch.from_numpy(rng.integers(0, 64, n, dtype=np.int32))

_device.require_gpu = lambda: torch.device("cpu")
_kernels.knn = lambda x, kk, **kw: (t_idx, t_dist, 0)
_kernels.fuzzy_simplicial_set = lambda i, d: (indptr, indices, data, None, None)
_kernels.leiden = lambda ip, ix, w, nn, **kw: (labels, 0.9, 64)

adata = sc.AnnData(np.zeros((n, 1), dtype=np.float32))
adata.obsm["X_pca"] = rng.standard_normal((n, 50)).astype(np.float32)


def run():
    t0 = time.perf_counter()
    sc.pp.neighbors(adata, use_rep="X_pca")
    t1 = time.perf_counter()
    sc.tl.leiden(adata, flavor="igraph", n_iterations=-1)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1


for rep in range(3):
    a, b = run()
    print(f"rep {rep}: neighbors host side {a * 1e3:.0f} ms   leiden host side {b * 1e3:.0f} ms")
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    run()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
