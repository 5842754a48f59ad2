"""Device-resident `pca -> neighbors -> leiden` on one rank's row shard (what bench.py times).

Everything stays in HBM between stages; the AnnData front-ends (`pp.pca`, `pp.neighbors`, `tl.leiden`)
are host-side wrappers around the same stage functions.

Row sharding: rank r owns the contiguous cell block [row_begin, row_end).
  pca        all-reduce of the int64 Gram matrix (32 MB at 2000 genes) -> scores stay sharded
  neighbors  all-gather of the n x 50 embedding (200 MB at 1M cells)   -> each rank answers its own queries
  graph      every rank: membership strengths of its own rows; all-to-all of the directed edges (j, i, w_ij) to the
             owner of row j (12 B per edge, E / P edges per rank); every rank merges its rows of
             C = W + W^T - W o W^T -- bit for bit the rows the single-device kernel produces
  leiden     does not shard (global community totals, order dependent): the CSR pieces are sent to rank 0, which
             runs it; labels broadcast
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import _kernels
from .preprocessing._pca_solver import GpuBackend, NoComm, _ChunkedRows, pca_fit


@dataclass
class PathResult:
    x_pca: torch.Tensor            # [n_local, n_comps] float32
    components: object             # np.ndarray [n_comps, g]
    variance: object
    variance_ratio: object
    knn_indices: torch.Tensor      # [n_local, k] int32 (self first)
    knn_distances: torch.Tensor    # [n_local, k] float64
    conn_indptr: torch.Tensor | None   # rank 0 only
    conn_indices: torch.Tensor | None
    conn_data: torch.Tensor | None
    labels: torch.Tensor           # [n_total] int32 on every rank
    modularity: float
    n_communities: int
    stage_ms: dict = field(default_factory=dict)
    info: dict = field(default_factory=dict)


class _Timer:
    def __init__(self, enabled: bool):
        self.enabled = enabled
        self.marks = []

    def mark(self, name: str):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))

    def result(self) -> dict:
        if not self.enabled or len(self.marks) < 2:
            return {}
        torch.cuda.synchronize()
        out = {}
        for (_, a), (name, b) in zip(self.marks[:-1], self.marks[1:]):
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        return out


def _all_gather_rows(t: torch.Tensor, comm, counts: list[int]) -> torch.Tensor:
    """Concatenate every rank's row block.  Blocks may differ in length by one row (shard_bounds); RCCL/gloo
    all-gather needs equal-sized contributions, so every block is padded to the longest one, gathered into one
    [world * max_rows, ...] buffer with a single collective, and the padding rows are dropped afterwards."""
    if comm.world_size == 1:
        return t
    import torch.distributed as dist

    mx = max(counts)
    t = t.contiguous()
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    group = getattr(comm, "group", None)
    if t.is_cuda and dist.get_backend(group) == "gloo":  # validation runs (several ranks on one GPU): host staging
        hbuf = torch.empty((comm.world_size * mx, *t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(hbuf, t.cpu(), group=group)
        buf = hbuf.to(t.device)
    else:
        buf = torch.empty((comm.world_size * mx, *t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf, t, group=group)
    if all(c == mx for c in counts):
        return buf
    return torch.cat([buf[r * mx: r * mx + c] for r, c in enumerate(counts)], dim=0)


def _staged(t: torch.Tensor, group) -> bool:
    """device tensors under gloo (validation runs with several ranks on one GPU) go through host memory"""
    import torch.distributed as dist

    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_to_all_rows(send: torch.Tensor, send_counts: list[int], group) -> torch.Tensor:
    """variable-sized all-to-all of the rows of `send` [m, c] (rows grouped by destination rank, `send_counts` rows
    each): one collective for the counts, one for the payload (RCCL all-to-all over xGMI; gloo in the CPU tests)"""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = send.device
    staged = _staged(send, group)
    cdev = torch.device("cpu") if staged else dev
    sc = torch.tensor(send_counts, dtype=torch.int64, device=cdev)
    rc = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(v) for v in rc.tolist()]
    src = send.cpu() if staged else send.contiguous()
    out = torch.empty((sum(recv_counts), send.shape[1]), dtype=send.dtype, device=cdev)
    dist.all_to_all_single(out, src, output_split_sizes=recv_counts, input_split_sizes=list(send_counts), group=group)
    return out.to(dev) if staged else out


def fixed_point_distance_sum(dist32: torch.Tensor, n_elements_total: int, comm) -> torch.Tensor:
    """Sum of ALL ranks' distances as a float64 [1] device tensor, bit-identical to the single-device kernels
    (csrc/fuzzy.hip: fss_max_kernel / fss_sum_kernel / fss_sum_final_kernel) for any sharding: the distances are added as
    int64 fixed point scaled by 2^S, S = 61 - e - ceil(log2(count)) with max < 2^e -- integer sums do not depend on the
    order, a float64 sum (torch.sum + all-reduce in rank order) does."""
    import math

    dev = dist32.device
    mx = dist32.max().reshape(1).to(torch.float64) if dist32.numel() else torch.zeros(1, dtype=torch.float64, device=dev)
    comm.allreduce_max_(mx)
    m = float(mx.item())
    e = math.frexp(m if m > 0.0 else 1.0)[1]
    lg = max(0, (int(n_elements_total) - 1).bit_length())
    s_bits = 61 - e - lg
    isum = torch.round(dist32.to(torch.float64) * (2.0 ** s_bits)).to(torch.int64).sum().reshape(1)
    comm.allreduce_(isum)
    return isum.to(torch.float64) * (2.0 ** -s_bits)


def sharded_fuzzy_rows(idx: torch.Tensor, dist32: torch.Tensor, comm, counts: list[int], row_begin: int, n_total: int):
    """This rank's rows of the symmetric fuzzy graph: local membership strengths, all-to-all of the
    directed edges to the owners of their targets, local merge.  -> (indptr [n_local + 1], indices (global), data)"""
    import numpy as np

    group = getattr(comm, "group", None)
    dev = idx.device
    n_local, k = idx.shape
    total = fixed_point_distance_sum(dist32, n_total * k, comm)  # the only global quantity of smooth_knn_dist
    w = _kernels.fuzzy_weights(idx, dist32, row_begin, n_total, total)
    mask = w > 0
    rows = torch.arange(row_begin, row_begin + n_local, device=dev, dtype=torch.int32)[:, None].expand(-1, k)
    j, i, ww = idx[mask].to(torch.int32), rows[mask], w[mask]
    ends = torch.from_numpy(np.cumsum(counts)).to(dev)
    dest = torch.bucketize(j.to(torch.int64), ends, right=True)  # owner of row j
    order = torch.argsort(dest, stable=True)
    send = torch.stack([j[order], i[order], ww[order].view(torch.int32)], dim=1).contiguous()  # (j, i, w_ij) as int32 x 3
    send_counts = torch.bincount(dest, minlength=comm.world_size).tolist()
    recv = _all_to_all_rows(send, send_counts, group)
    jl = recv[:, 0].to(torch.int64) - row_begin
    src = recv[:, 1].contiguous()
    order2 = torch.argsort(jl * n_total + src.to(torch.int64))  # by (row, source): the merge kernel bisects the sources
    in_indptr = torch.zeros(n_local + 1, dtype=torch.int64, device=dev)
    in_indptr[1:] = torch.cumsum(torch.bincount(jl, minlength=n_local), dim=0)
    return _kernels.fuzzy_merge_rows(idx, w, in_indptr, src[order2], recv[:, 2].contiguous().view(torch.float32)[order2])


def _gather_csr_rows_to_root(indptr, indices, data, comm, n_total: int):
    """rank 0 receives every rank's CSR rows and stitches the n_total-row matrix together; the others return None.
    Two collectives (row lengths; entries as (column, weight-bits) int32 pairs), each an all-to-all in which only
    rank 0 receives -- RCCL runs the P - 1 transfers concurrently over the point-to-point xGMI links, where round 2's
    loop of blocking send / recv pairs took them one after the other."""
    group = getattr(comm, "group", None)
    world = comm.world_size
    rowcnt = (indptr[1:] - indptr[:-1]).to(torch.int32).reshape(-1, 1).contiguous()
    ent = torch.stack([indices.to(torch.int32), data.to(torch.float32).view(torch.int32)], dim=1).contiguous()
    to_root = lambda m: [int(m)] + [0] * (world - 1)  # noqa: E731  (everything goes to rank 0)
    rc = _all_to_all_rows(rowcnt, to_root(rowcnt.shape[0]), group)
    en = _all_to_all_rows(ent, to_root(ent.shape[0]), group)
    if comm.rank != 0:
        return None, None, None
    # all_to_all output is ordered by source rank = by row block
    ci = torch.zeros(n_total + 1, dtype=torch.int64, device=indices.device)
    ci[1:] = torch.cumsum(rc[:, 0].to(torch.int64), dim=0)
    return ci, en[:, 0].contiguous(), en[:, 1].contiguous().view(torch.float32)


def shard_bounds(n_total: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced row blocks: the first n_total % world_size ranks get one extra row."""
    base, rem = divmod(n_total, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def run_path(a_handle, n_total: int, *, comm=None, backend=None, n_comps: int = 50, n_neighbors: int = 15,
             resolution: float = 1.0, n_iterations: int = -1, seed: int = 0, svd_solver: str = "arpack",
             timing: bool = False, nprobe: int | None = None) -> PathResult:
    """`a_handle` = `GpuBackend.upload(csr_rows_of_this_rank)`; rows of rank r are shard_bounds(n_total, W, r).
    Out of core: a `_ChunkedRows` over this rank's rows instead (e.g. `BackedCsr.row_chunks(step, row_begin, row_end)`
    of an on-disk matrix: every rank streams only its own block from the store).
    `nprobe`: None = exact kNN; p > 0 = the approximate IVF search (`pp.neighbors(transformer='ivf')`)."""
    comm = comm or NoComm()
    backend = backend or GpuBackend()
    tm = _Timer(timing)
    tm.mark("start")
    world, rank = comm.world_size, comm.rank
    counts = [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]
    row_begin, row_end = shard_bounds(n_total, world, rank)
    n_local = a_handle.n_rows if isinstance(a_handle, _ChunkedRows) else a_handle[3]
    assert n_local == row_end - row_begin, "shard does not match shard_bounds()"

    res = pca_fit(a_handle, n_comps, backend=backend, comm=comm, svd_solver=svd_solver, seed=seed)
    tm.mark("pca")
    emb = _all_gather_rows(res.scores, comm, counts)  # [n_total, n_comps] float32 on every rank
    k = min(n_neighbors, n_total)
    idx, dist, n_fallback = _kernels.knn(emb, k, q_begin=row_begin, n_query=row_end - row_begin, nprobe=nprobe)
    tm.mark("knn")
    dev = emb.device
    labels = torch.empty(n_total, dtype=torch.int32, device=dev)
    q, nc = 0.0, 0
    leiden_stats = None
    ci = cx = cd = None
    if world == 1:
        ci, cx, cd, _, _ = _kernels.fuzzy_simplicial_set(idx, dist.to(torch.float32))
    else:
        li, lx, ld = sharded_fuzzy_rows(idx, dist.to(torch.float32), comm, counts, row_begin, n_total)
        ci, cx, cd = _gather_csr_rows_to_root(li, lx, ld, comm, n_total)
    tm.mark("connectivities")
    if rank == 0:
        labels, q, nc = _kernels.leiden(ci, cx, cd, n_total, resolution=resolution, n_iterations=n_iterations, seed=seed)
        try:  # (diagnostics only: a backend that replaces `_kernels.leiden` need not have the library loaded)
            leiden_stats = _kernels.leiden_last_stats()
        except Exception:  # noqa: BLE001
            leiden_stats = {}
        tm.mark("leiden")
    if world > 1:
        import torch.distributed as tdist

        group = getattr(comm, "group", None)
        meta = torch.tensor([q, float(nc)], dtype=torch.float64, device=dev)
        if labels.is_cuda and tdist.get_backend(group) == "gloo":  # validation runs: host staging
            hl, hm = labels.cpu(), meta.cpu()
            tdist.broadcast(hl, src=0, group=group)
            tdist.broadcast(hm, src=0, group=group)
            labels, meta = hl.to(dev), hm
        else:
            tdist.broadcast(labels, src=0, group=group)
            tdist.broadcast(meta, src=0, group=group)
        q, nc = float(meta[0]), int(meta[1])
        tm.mark("broadcast")
    info = dict(res.info)
    info["knn_fallback_queries"] = n_fallback
    if leiden_stats is not None:
        info["leiden_stats"] = leiden_stats
    return PathResult(res.scores, res.components, res.explained_variance, res.explained_variance_ratio, idx, dist,
                      ci, cx, cd, labels, q, nc, tm.result(), info)
