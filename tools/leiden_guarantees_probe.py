"""Does the GPU Leiden's stable partition have the properties the Leiden paper guarantees?  (oracle/leiden_guarantees.py)

    python tools/leiden_guarantees_probe.py [n_cells] [structure ...]

The path's own fuzzy graph of the bench matrix -> `scamd_leiden_csr_f32` (n_iterations = -1) -> on the host: vertices a
single move would improve (node optimality), community pairs a merge would improve (g-separation), disconnected
communities."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    structures = sys.argv[2:] or ["planted", "none", "weak"]
    import torch
    from scipy import sparse
    from scipy.sparse.csgraph import connected_components

    import bench
    from oracle import leiden_guarantees as lg
    from scanpy_amd import _kernels as K
    from scanpy_amd._pipeline import run_path
    from scanpy_amd.preprocessing._pca_solver import GpuBackend

    backend = GpuBackend()
    for structure in structures:
        x, _ = bench.make_matrix(n, 2000, 0, structure)
        res = run_path(backend.upload(x), n, backend=backend)
        ip, ix, w = res.conn_indptr, res.conn_indices, res.conn_data
        labels, q, nc = K.leiden(ip, ix, w, n)
        torch.cuda.synchronize()
        conn = sparse.csr_matrix((w.cpu().numpy(), ix.cpu().numpy(), ip.cpu().numpy()), shape=(n, n))
        lab = labels.cpu().numpy()
        t0 = time.perf_counter()
        im = lg.improving_moves(conn, lab)
        mp = lg.mergeable_pairs(conn, lab)
        same = lab[np.repeat(np.arange(n), np.diff(conn.indptr))] == lab[conn.indices]
        inner = sparse.csr_matrix((same.astype(np.int8), conn.indices.copy(), conn.indptr.copy()), shape=conn.shape)  # (eliminate_zeros works in place)
        inner.eliminate_zeros()
        ncomp, _ = connected_components(inner, directed=False)
        print(f"{structure} n={n}: Q {q:.6f}, {nc} communities; improving moves {im['count']} ({im['fraction']:.2e} of the vertices, "
              f"max gain {im['max_gain']:.3e} Q); mergeable pairs {mp['count']} (max gain {mp['max_gain']:.3e}); "
              f"components of the within-community graph {ncomp} (= communities: {ncomp == nc}); checks {time.perf_counter() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
