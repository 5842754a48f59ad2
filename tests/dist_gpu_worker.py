"""Worker for the "N ranks on ONE GPU" validation of the row-sharded GPU path (TEST INFRASTRUCTURE).

Every rank drives cuda:0 with the real kernels (GpuBackend, scamd_* through the C ABI); the collectives go through gloo
with host staging (`TorchDistComm._host_staged`) because RCCL refuses two ranks on one device.  What this covers, on
real hardware, is everything of the multi-GPU path except RCCL itself: shard_bounds, the int64 Gram all-reduce, the
embedding all-gather, query-sharded cell-pruned kNN against all candidates, the kNN-list gather, graph + Leiden on
rank 0 and the label broadcast."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import os as _os  # noqa: E402

if _os.environ.get("SCAMD_TESTS_ON_EMULATOR") == "1":  # the host-emulated kernel library, tests/emu/README.md
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    import patch_torch  # noqa: E402

    patch_torch.activate()


def run(rank: int, world: int, init_file: str, out_dir: str, n: int, g: int, n_comps: int):
    import torch.distributed as dist

    from scanpy_amd._pipeline import run_path, shard_bounds
    from scanpy_amd.datasets import synthetic_planted
    from scanpy_amd.preprocessing._pca_solver import GpuBackend, NoComm, TorchDistComm

    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
        comm = TorchDistComm()
    else:
        comm = NoComm()
    lo, hi = shard_bounds(n, world, rank)
    x, _ = synthetic_planted(n, g, n_types=16, seed=9, row_range=(lo, hi))
    be = GpuBackend()
    res = run_path(be.upload(x), n, comm=comm, backend=be, n_comps=n_comps, n_neighbors=15)
    torch.cuda.synchronize()
    np.savez(Path(out_dir) / f"gpu_rank{rank}_of{world}.npz", scores=res.x_pca.cpu().numpy(), components=res.components,
             variance=res.variance, knn_idx=res.knn_indices.cpu().numpy(), knn_dist=res.knn_distances.cpu().numpy(),
             labels=res.labels.cpu().numpy(), q=res.modularity, nc=res.n_communities, lo=lo, hi=hi,
             has_graph=res.conn_indptr is not None,
             **({"conn_indptr": res.conn_indptr.cpu().numpy(), "conn_indices": res.conn_indices.cpu().numpy(),
                 "conn_data": res.conn_data.cpu().numpy()} if res.conn_indptr is not None else {}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]), int(a[1]), a[2], a[3], int(a[4]), int(a[5]), int(a[6]))
