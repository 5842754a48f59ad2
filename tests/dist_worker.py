"""Worker for the world_size-2 gloo tests (TEST INFRASTRUCTURE): runs the row-sharded host logic of the path on CPU.

The kernel layer is replaced by CPU stand-ins built from the oracle (tests may use the oracle as a checker; the
product never does): `CpuStubBackend` for the PCA data passes, sklearn brute kNN / oracle fuzzy set / oracle Leiden
for the `_kernels` entry points `run_path` calls.  What is exercised is the product's sharding and collective
logic: shard_bounds, the float64 all-reduces of the PCA panels, the all-gather of embedding rows and kNN lists,
Leiden on rank 0 and the label broadcast."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _patch_kernels():
    """CPU stand-ins with the signatures of scanpy_amd._kernels.{knn, fuzzy_simplicial_set, leiden}."""
    from oracle import connectivities as oc
    from oracle import knn as oknn
    from oracle import leiden as ol
    from scanpy_amd import _kernels

    def knn(x, k, *, q_begin=0, n_query=None, cert_scale=1.0, nprobe=None):  # (exact whatever nprobe says)
        xn = x.numpy()
        n = xn.shape[0]
        nq = n - q_begin if n_query is None else n_query
        idx, dist, _ = oknn.knn_sklearn(xn, k)
        sl = slice(q_begin, q_begin + nq)
        return (torch.from_numpy(idx[sl].astype(np.int32)), torch.from_numpy(dist[sl].astype(np.float64)), 0)

    def fuzzy_simplicial_set(knn_idx, knn_dist):
        n, k = knn_idx.shape
        conn, sig, rho = oc.fuzzy_simplicial_set(knn_idx.numpy(), knn_dist.numpy().astype(np.float32), n, k)
        return (torch.from_numpy(conn.indptr.astype(np.int64)), torch.from_numpy(conn.indices.astype(np.int32)),
                torch.from_numpy(conn.data.astype(np.float32)), torch.from_numpy(sig), torch.from_numpy(rho))

    def leiden(indptr, indices, weights, n, *, resolution=1.0, n_iterations=-1, beta=0.01, seed=0, initial_membership=None, objective="modularity", node_weights=None):
        from scipy import sparse

        assert objective == "modularity" and initial_membership is None, "the CPU stand-in starts from singletons"

        adj = sparse.csr_matrix((weights.numpy(), indices.numpy(), indptr.numpy()), shape=(n, n))
        memb, q = ol.leiden(adj, resolution=resolution, n_iterations=n_iterations, seed=seed)
        return torch.from_numpy(memb.astype(np.int32)), q, int(memb.max()) + 1

    def modularity(indptr, indices, weights, n, membership, *, resolution=1.0):
        """CPU stand-in of scamd_modularity_csr_f32 (what `tl.leiden` stores when the run's resolution is not 1)"""
        from scipy import sparse

        adj = sparse.csr_matrix((weights.numpy(), indices.numpy(), indptr.numpy()), shape=(n, n))
        return ol.modularity(adj, membership.numpy(), resolution=resolution)

    def fuzzy_weights(knn_idx, knn_dist, row_begin, n_total, sum_all):
        """membership strengths of a row shard (CPU stand-in of scamd_fuzzy_weights_f32)"""
        idx = knn_idx.numpy()
        d = knn_dist.numpy().astype(np.float32)
        n, k = idx.shape
        sig, rho = oc.smooth_knn_dist_vec(d, k)  # (the global mean only matters for rows without a positive distance)
        # compute_membership_strengths zeroes the entries equal to the LOCAL row number (and -1 = missing): map the own
        # global id to the local one and every other id to a value that is neither
        self_local = np.where(idx == (row_begin + np.arange(n))[:, None], np.arange(n)[:, None], n_total + 7)
        _, _, val = oc.compute_membership_strengths(self_local, d, sig, rho)
        return torch.from_numpy(val.reshape(n, k).astype(np.float32))

    def fuzzy_merge_rows(knn_idx, w, in_indptr, in_src, in_w):
        """W + W^T - W o W^T on the local rows from out-edges and received in-edges (stand-in of the merge kernel)"""
        idx, wv = knn_idx.numpy(), w.numpy()
        ip, src, win = in_indptr.numpy(), in_src.numpy(), in_w.numpy()
        n, k = idx.shape
        indptr, cols, vals = [0], [], []
        for i in range(n):
            out = {int(idx[i, j]): np.float32(wv[i, j]) for j in range(k) if wv[i, j] > 0}
            inn = {int(src[e]): np.float32(win[e]) for e in range(ip[i], ip[i + 1])}
            row = {}
            for c in set(out) | set(inn):
                a, b = out.get(c, np.float32(0)), inn.get(c, np.float32(0))
                row[c] = np.float32(np.float32(a + b) - np.float32(a * b))
            for c in sorted(row):
                cols.append(c)
                vals.append(row[c])
            indptr.append(len(cols))
        return (torch.from_numpy(np.asarray(indptr, dtype=np.int64)), torch.from_numpy(np.asarray(cols, dtype=np.int32)),
                torch.from_numpy(np.asarray(vals, dtype=np.float32)))

    _kernels.knn = knn
    _kernels.fuzzy_simplicial_set = fuzzy_simplicial_set
    _kernels.fuzzy_weights = fuzzy_weights
    _kernels.fuzzy_merge_rows = fuzzy_merge_rows
    _kernels.leiden = leiden
    _kernels.modularity = modularity
    _kernels.leiden_last_stats = lambda: {}  # (the library's thread-local stats say nothing about a stand-in run)


def run(rank: int, world: int, init_file: str, out_dir: str, n: int, g: int, n_comps: int, mode: str):
    import torch.distributed as dist

    from scanpy_amd._pipeline import run_path, shard_bounds
    from scanpy_amd.datasets import synthetic_planted
    from scanpy_amd.preprocessing._pca_solver import NoComm, TorchDistComm, pca_fit
    from stub_backend import CpuStubBackend

    if world > 1:
        dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
        comm = TorchDistComm()
    else:
        comm = NoComm()
    lo, hi = shard_bounds(n, world, rank)
    x, _ = synthetic_planted(n, g, n_types=12, seed=5, row_range=(lo, hi))
    be = CpuStubBackend()
    out = {}
    if mode == "pca":
        res = pca_fit(be.upload(x), n_comps, backend=be, comm=comm)
        out = dict(scores=res.scores.numpy(), components=res.components, variance=res.explained_variance,
                   ratio=res.explained_variance_ratio, mean=res.mean, lo=lo, hi=hi)
    else:
        _patch_kernels()
        if mode == "path_backed":  # every rank streams its own row block from the store the test wrote
            import scanpy_amd as sc
            from scanpy_amd.preprocessing._pca_solver import _ChunkedRows

            b = sc.read_zarr(Path(out_dir) / "store.zarr", backed="r").X
            assert (b.rows(lo, hi).to_scipy() != x).nnz == 0
            a = _ChunkedRows(b.row_chunks(173, lo, hi), g)
        else:
            a = be.upload(x)
        res = run_path(a, n, comm=comm, backend=be, n_comps=n_comps, n_neighbors=10)
        out = dict(scores=res.x_pca.numpy(), knn_idx=res.knn_indices.numpy(), knn_dist=res.knn_distances.numpy(),
                   labels=res.labels.numpy(), q=res.modularity, nc=res.n_communities, lo=lo, hi=hi,
                   has_graph=res.conn_indptr is not None)
        if res.conn_indptr is not None:
            out.update(conn_indptr=res.conn_indptr.numpy(), conn_indices=res.conn_indices.numpy(), conn_data=res.conn_data.numpy())
    np.savez(Path(out_dir) / f"rank{rank}_of{world}.npz", **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]), int(a[1]), a[2], a[3], int(a[4]), int(a[5]), int(a[6]), a[7])
