"""Worker of the multi-resolution Leiden replica test (TEST INFRASTRUCTURE): `mode` cpu = CPU stand-in kernels (gloo),
gpu = real kernels, all ranks on cuda:0 (gloo)."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import os as _os  # noqa: E402

if _os.environ.get("SCAMD_TESTS_ON_EMULATOR") == "1":  # the host-emulated kernel library, tests/emu/README.md
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    import patch_torch  # noqa: E402

    patch_torch.activate()


def run(rank: int, world: int, init_file: str, out_dir: str, mode: str):
    import torch
    import torch.distributed as dist

    import scanpy_amd as sc

    if mode == "cpu":
        from dist_worker import _patch_kernels
        from scanpy_amd import _device

        _device.require_gpu = lambda: torch.device("cpu")
        _patch_kernels()
    else:
        torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    f = dict(np.load(ROOT / "tests" / "golden" / "pbmc68k_reduced.npz"))
    from scipy import sparse

    conn = sparse.csr_matrix((f["connectivities_data"].astype(np.float32), f["connectivities_indices"], f["connectivities_indptr"]),
                             shape=tuple(f["connectivities_shape"]))
    adata = sc.AnnData(f["X"])
    adata.obsp["connectivities"] = conn
    adata.uns["neighbors"] = dict(connectivities_key="connectivities", distances_key="distances", params=dict(method="umap"))
    keys = sc.tl.leiden_multires(adata, [0.3, 0.6, 1.0, 1.5, 2.5], flavor="igraph", n_iterations=-1)
    out = {k: adata.obs[k].cat.codes.to_numpy() for k in keys}
    out.update({f"q_{k}": adata.uns[k]["modularity"] for k in keys})
    np.savez(Path(out_dir) / f"multires_{mode}_rank{rank}_of{world}.npz", **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]), int(a[1]), a[2], a[3], a[4])
