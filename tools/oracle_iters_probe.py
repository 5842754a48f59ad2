"""How many iterations does the SEQUENTIAL oracle (oracle/leiden.c, n_iterations=-1) need on the SAME 1M-cell graph the GPU
optimiser is timed on?  ("Next round" 1a: is the cap of the GPU's outer loop the algorithm's behaviour or
an artefact of the synchronous sub-rounds?)

The graph is built by the GPU path (pca -> exact kNN -> fuzzy set), copied to the host, and handed to the oracle; the GPU
optimiser runs on the same CSR.  Prints both iteration traces (Q per iteration), times, and the ARI between the two.

    python tools/oracle_iters_probe.py 1000000 weak [seed]
"""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

import numpy as np
from scipy import sparse

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    structure = sys.argv[2] if len(sys.argv) > 2 else "weak"
    seeds = [int(v) for v in sys.argv[3].split(",") if v != "none"] if len(sys.argv) > 3 else [0]  # ("none": no oracle run)
    gpu_seeds = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else (seeds or [0])
    import torch

    import bench
    from oracle import compare as cmp
    from oracle import leiden as ol
    from scanpy_amd import _kernels as K
    from scanpy_amd._pipeline import run_path
    from scanpy_amd.preprocessing._pca_solver import GpuBackend

    ol.build()
    x, truth = bench.make_matrix(n, 2000, 0, structure)
    backend = GpuBackend()
    res = run_path(backend.upload(x), n, backend=backend)
    ip, ix, w = res.conn_indptr, res.conn_indices, res.conn_data
    torch.cuda.synchronize()
    K.leiden(ip, ix, w, n, seed=gpu_seeds[0])  # warm-up
    gpu = {}
    for sd in gpu_seeds:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        labels, q, nc = K.leiden(ip, ix, w, n, seed=sd)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
        stats = K.leiden_last_stats()
        gpu[sd] = labels.cpu().numpy()
        print(f"gpu seed {sd}: {t_gpu * 1e3:.1f} ms, Q {q!r}, {nc} communities, ARI vs truth {cmp.ari(gpu[sd], truth):.4f}, {stats}", flush=True)
    conn = sparse.csr_matrix((w.cpu().numpy().astype(np.float64), ix.cpu().numpy(), ip.cpu().numpy()), shape=(n, n))
    del res
    os.environ["ORACLE_LEIDEN_DEBUG"] = "1"
    for sd in seeds:
        t0 = time.perf_counter()
        om, oq = ol.leiden(conn, resolution=1.0, n_iterations=-1, seed=sd)
        t_cpu = time.perf_counter() - t0
        print(f"oracle seed {sd}: {t_cpu:.1f} s, Q {oq!r}, {int(om.max()) + 1} communities, ARI vs truth {cmp.ari(om, truth):.4f}; "
              f"ARI vs gpu seeds {[round(cmp.ari(g, om), 4) for g in gpu.values()]}", flush=True)


if __name__ == "__main__":
    main()
