"""Multi-resolution Leiden as REPLICAS (BASELINE configs[4] "multi-resolution leiden"): Leiden does not
shard, but a resolution sweep is embarrassingly parallel -- every rank holds the graph and runs `sc.tl.leiden` for its
share of the resolutions, one exchange at the end hands every rank all labelings.  No collective on the data path.

Not a reference function (a scanpy user loops over `sc.tl.leiden(adata, resolution=r, key_added=...)`); this is that
loop, distributed over the GPUs of `torch.distributed`'s world when one is initialised."""
from __future__ import annotations

import numpy as np
import pandas as pd

from .._anndata import is_anndata
from ._leiden import leiden


def resolution_owner(i: int, world_size: int) -> int:
    """round-robin assignment of the i-th resolution to a rank"""
    return i % world_size


def leiden_multires(adata, resolutions, *, key_prefix: str = "leiden_r", group=None, **leiden_kwargs) -> list[str]:
    """Run `tl.leiden` for every resolution; writes `obs[f'{key_prefix}{resolution}']` and
    `uns[...]['params'|'modularity']` for each one on EVERY rank and returns the keys.

    Single process: a plain loop on the one GPU.  Under `torch.distributed` (one process per GPU, the AnnData replicated
    on every rank): rank r computes the resolutions i with i % world_size == r."""
    if not is_anndata(adata):
        raise TypeError("leiden_multires() expects an AnnData-like object")
    resolutions = [float(r) for r in resolutions]
    if len(set(resolutions)) != len(resolutions):
        raise ValueError("resolutions must be distinct")
    for forbidden in ("key_added", "resolution", "copy"):
        if forbidden in leiden_kwargs:
            raise TypeError(f"{forbidden!r} is set per resolution by leiden_multires")
    world, rank, dist = 1, 0, None
    try:
        import torch.distributed as tdist

        if tdist.is_available() and tdist.is_initialized():
            dist, world, rank = tdist, tdist.get_world_size(group), tdist.get_rank(group)
    except ImportError:
        pass
    keys = [f"{key_prefix}{r:g}" for r in resolutions]
    mine = {}
    for i, (r, key) in enumerate(zip(resolutions, keys)):
        if resolution_owner(i, world) != rank:
            continue
        leiden(adata, resolution=r, key_added=key, **leiden_kwargs)
        mine[i] = (adata.obs[key].cat.codes.to_numpy().astype(np.int32), dict(adata.uns[key]))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine, group=group)  # labels: n int32 per resolution, once, off the data path
        for part in gathered:
            for i, (codes, uns) in part.items():
                if i in mine:
                    continue
                n_cat = int(codes.max()) + 1 if codes.size else 0
                adata.obs[keys[i]] = pd.Categorical.from_codes(codes, categories=[str(c) for c in range(n_cat)])
                adata.uns[keys[i]] = uns
    return keys
