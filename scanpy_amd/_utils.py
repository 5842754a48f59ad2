"""Host-side helpers mirroring the reference's L3 utilities for the path."""
from __future__ import annotations

import warnings

import numpy as np
from scipy import sparse

from ._settings import settings

_UNSET = object()


def resolve_seed(rng, random_state):
    """Mirror of `_accepts_legacy_random_state(0)` (src/scanpy/_utils/random.py:182-208).

    Returns (seed:int, meta:dict).  `meta` is `{'random_state': seed}` for the legacy/default call style
    and `{}` when the new-style `rng=` was given (src/scanpy/neighbors/__init__.py:205-207).
    """
    if rng is not None and random_state is not _UNSET:
        raise TypeError("Specify at most one of `rng` and `random_state`.")
    if rng is not None:
        if isinstance(rng, np.random.Generator):
            seed = int(rng.integers(0, 2**31 - 1))
        else:
            seed = int(np.random.default_rng(rng).integers(0, 2**31 - 1)) if not isinstance(rng, (int, np.integer)) else int(rng)
        return seed, {}
    rs = 0 if random_state is _UNSET else random_state
    if isinstance(rs, np.random.RandomState):
        seed = int(rs.randint(0, 2**31 - 1))
        return seed, {"random_state": rs}
    if rs is None:
        seed = int(np.random.default_rng().integers(0, 2**31 - 1))
        return seed, {"random_state": None}
    return int(rs), {"random_state": rs}


def choose_representation(adata, *, use_rep=None, n_pcs=None):
    """src/scanpy/tools/_utils.py:20-78."""
    if use_rep is None and n_pcs == 0:
        use_rep = "X"
    if use_rep is None:
        if adata.n_vars > settings.N_PCS:
            if "X_pca" in adata.obsm:
                if n_pcs is not None and n_pcs > adata.obsm["X_pca"].shape[1]:
                    msg = "`X_pca` does not have enough PCs. Rerun `sc.pp.pca` with adjusted `n_comps`."
                    raise ValueError(msg)
                x = adata.obsm["X_pca"][:, :n_pcs]
            else:
                warnings.warn(f"You’re trying to run this on {adata.n_vars} dimensions of `.X`, "
                              "if you really want this, set `use_rep='X'`.\n         "
                              "Falling back to preprocessing with `sc.pp.pca` and default params.", UserWarning, stacklevel=3)
                from .preprocessing._pca import pca

                n_pcs_pca = n_pcs if n_pcs is not None else settings.N_PCS
                pca(adata, n_comps=n_pcs_pca)
                x = adata.obsm["X_pca"]
        else:
            x = adata.X
    elif use_rep in adata.obsm and n_pcs is not None:
        if n_pcs > adata.obsm[use_rep].shape[1]:
            msg = f"{use_rep} does not have enough Dimensions. Provide a Representation with equal or more dimensions than`n_pcs` or lower `n_pcs` "
            raise ValueError(msg)
        x = adata.obsm[use_rep][:, :n_pcs]
    elif use_rep in adata.obsm and n_pcs is None:
        x = adata.obsm[use_rep]
    elif use_rep == "X":
        x = adata.X
    else:
        msg = f"Did not find {use_rep} in `.obsm.keys()`. You need to compute it first."
        raise ValueError(msg)
    if getattr(x, "is_backed", False):  # `.X` itself as the representation: a search needs it whole
        x = x.to_memory()
    return x


def choose_graph(adata, obsp=None, neighbors_key=None):
    """src/scanpy/_utils/__init__.py:969-986 (+ the slice of NeighborsView it needs)."""
    if obsp is not None and neighbors_key is not None:
        msg = "You can't specify both obsp, neighbors_key. Please select only one."
        raise ValueError(msg)
    if obsp is not None:
        return adata.obsp[obsp]
    key = "neighbors" if neighbors_key is None else neighbors_key
    if key not in adata.uns:
        if neighbors_key is None:
            msg = "You need to run `pp.neighbors` first to compute a neighborhood graph."
        else:
            msg = f"No {neighbors_key!r} in .uns"
        raise ValueError(msg) if neighbors_key is None else KeyError(msg)
    conn_key = adata.uns[key].get("connectivities_key", "connectivities")
    if conn_key not in adata.obsp:
        msg = f"No {conn_key!r} in .obsp"
        raise KeyError(msg)
    return adata.obsp[conn_key]


def natsorted_str(values):
    """Natural sort of string labels ('2' < '10'; restrict_to labels like 'a,3'); natsort is not installed."""
    import re

    def key(s):
        return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]

    return sorted(values, key=key)


def as_csr_f32(x):
    """Host CSR float32 with sorted int32 indices (what the kernels consume)."""
    if sparse.issparse(x):
        x = x.tocsr()
    else:
        x = sparse.csr_matrix(np.asarray(x))
    if not x.has_sorted_indices:
        x = x.sorted_indices()
    return x


def view_to_actual(adata) -> None:
    """src/scanpy/_utils/__init__.py:474-478: in-place functions turn a view into an actual AnnData first."""
    if getattr(adata, "is_view", False):
        warnings.warn("Received a view of an AnnData. Making a copy.", UserWarning, stacklevel=3)
        if hasattr(adata, "_init_as_actual"):  # a real anndata.AnnData
            adata._init_as_actual(adata.copy())
        else:  # the stand-in: its views already own their arrays
            adata.is_view = False
