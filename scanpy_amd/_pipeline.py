"""Device-resident `pca -> neighbors -> leiden` on one rank's row shard (what bench.py times).

Everything stays in HBM between stages; the AnnData front-ends (`pp.pca`, `pp.neighbors`, `tl.leiden`)
are host-side wrappers around the same stage functions.

Row sharding: rank r owns the contiguous cell block [row_begin, row_end).
  pca        all-reduce of g x b float64 panels (<= 1 MB)            -> scores stay sharded
  neighbors  all-gather of the n x 50 embedding (200 MB at 1M cells) -> each rank answers its own queries
  graph      all-gather of the kNN lists (n x k x 12 B)              -> fuzzy set + Leiden on rank 0
  leiden     does not shard (global community totals, order dependent): runs on rank 0, labels broadcast
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


def shard_bounds(n_total: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced row blocks: the first n_total % world_size ranks get one extra row."""
    base, rem = divmod(n_total, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def run_path(a_handle, n_total: int, *, comm=None, backend=None, n_comps: int = 50, n_neighbors: int = 15,
             resolution: float = 1.0, n_iterations: int = -1, seed: int = 0, svd_solver: str = "arpack",
             timing: bool = False) -> PathResult:
    """`a_handle` = `GpuBackend.upload(csr_rows_of_this_rank)`; rows of rank r are shard_bounds(n_total, W, r).
    Out of core: a `_ChunkedRows` over this rank's rows instead (e.g. `BackedCsr.row_chunks(step, row_begin, row_end)`
    of an on-disk matrix: every rank streams only its own block from the store)."""
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
    idx, dist, n_fallback = _kernels.knn(emb, k, q_begin=row_begin, n_query=row_end - row_begin)
    tm.mark("knn")
    idx_all = _all_gather_rows(idx, comm, counts)
    dist_all = _all_gather_rows(dist.to(torch.float32), comm, counts)
    dev = emb.device
    labels = torch.empty(n_total, dtype=torch.int32, device=dev)
    q, nc = 0.0, 0
    ci = cx = cd = None
    if rank == 0:
        ci, cx, cd, _, _ = _kernels.fuzzy_simplicial_set(idx_all, dist_all)
        tm.mark("connectivities")
        labels, q, nc = _kernels.leiden(ci, cx, cd, n_total, resolution=resolution, n_iterations=n_iterations, seed=seed)
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
    return PathResult(res.scores, res.components, res.explained_variance, res.explained_variance_ratio, idx, dist,
                      ci, cx, cd, labels, q, nc, tm.result(), info)
