"""A minimal AnnData-shaped container, used when the `anndata` package is not installed.

The path only touches X / layers / obs / var / obsm / varm / obsp / uns, `n_obs`, `n_vars`, `shape`,
`copy()` and column subsetting `adata[:, mask]` (src/scanpy/preprocessing/_pca/__init__.py:232).
A real `anndata.AnnData` is accepted everywhere this class is.
"""
from __future__ import annotations

import copy as _copy

import numpy as np
import pandas as pd
from scipy import sparse


class AnnData:
    def __init__(self, X=None, obs=None, var=None, *, obsm=None, varm=None, obsp=None, uns=None, layers=None):
        if X is not None and not sparse.issparse(X) and not getattr(X, "is_backed", False):  # `_backed.BackedCsr`
            X = np.asarray(X)
        self.X = X
        n_obs, n_vars = X.shape if X is not None else (len(obs), len(var))
        self.obs = obs if obs is not None else pd.DataFrame(index=pd.RangeIndex(n_obs).astype(str))
        self.var = var if var is not None else pd.DataFrame(index=pd.RangeIndex(n_vars).astype(str))
        self.obsm = dict(obsm or {})
        self.varm = dict(varm or {})
        self.obsp = dict(obsp or {})
        self.uns = dict(uns or {})
        self.layers = dict(layers or {})
        self.is_view = False
        self.raw = None  # optionally another AnnData over the same cells (`adata.raw`), set by the readers
        self.varp = {}

    @property
    def n_obs(self) -> int:
        return len(self.obs)

    @property
    def n_vars(self) -> int:
        return len(self.var)

    @property
    def shape(self):
        return (self.n_obs, self.n_vars)

    @property
    def var_names(self):
        return self.var.index

    @property
    def obs_names(self):
        return self.obs.index

    def _inplace_subset_var(self, index) -> None:
        """anndata's `AnnData._inplace_subset_var` (used by `highly_variable_genes(subset=True)`)."""
        sub = self[:, np.asarray(index)]
        self.X, self.var, self.varm, self.layers = sub.X, sub.var.copy(), sub.varm, sub.layers
        self.varp = sub.varp

    def _inplace_subset_obs(self, index) -> None:
        """anndata's `AnnData._inplace_subset_obs` (used by `filter_cells`)."""
        sub = self[np.asarray(index)]
        self.X, self.obs, self.obsm, self.layers, self.obsp = sub.X, sub.obs.copy(), sub.obsm, sub.layers, sub.obsp
        self.raw = sub.raw  # `raw` keeps all genes but follows the cells
        if self.raw is not None:
            self.raw.obs = self.obs

    def copy(self) -> "AnnData":
        out = AnnData(
            None if self.X is None else self.X.copy(),
            self.obs.copy(),
            self.var.copy(),
            obsm={k: v.copy() for k, v in self.obsm.items()},
            varm={k: v.copy() for k, v in self.varm.items()},
            obsp={k: v.copy() for k, v in self.obsp.items()},
            uns=_copy.deepcopy(self.uns),
            layers={k: v.copy() for k, v in self.layers.items()},
        )
        out.varp = {k: v.copy() for k, v in self.varp.items()}
        if self.raw is not None:
            r = self.raw
            out.raw = AnnData(None if r.X is None else r.X.copy(), out.obs, r.var.copy(),
                              varm={k: v.copy() for k, v in r.varm.items()})
        return out

    def __getitem__(self, index) -> "AnnData":
        """Only `adata[:, var_mask]` and `adata[obs_mask]` / `adata[obs_mask, :]` are supported."""
        if not isinstance(index, tuple):
            index = (index, slice(None))
        oi, vi = index
        oi = slice(None) if oi is None else oi
        X = self.X[oi][:, vi] if self.X is not None else None
        sub = AnnData(
            X,
            self.obs.iloc[oi] if not isinstance(oi, slice) or oi != slice(None) else self.obs,
            self.var.iloc[vi] if not isinstance(vi, slice) or vi != slice(None) else self.var,
            obsm={k: v[oi] for k, v in self.obsm.items()},
            varm={k: v[vi] for k, v in self.varm.items()},
            uns=self.uns,
            layers={k: v[oi][:, vi] for k, v in self.layers.items()},
        )
        all_obs = isinstance(oi, slice) and oi == slice(None)
        all_var = isinstance(vi, slice) and vi == slice(None)
        # pairwise slots follow their own axis only: obsp survives any var subset and vice versa
        sub.obsp = dict(self.obsp) if all_obs else {k: v[oi][:, oi] for k, v in self.obsp.items()}
        sub.varp = dict(self.varp) if all_var else {k: v[vi][:, vi] for k, v in self.varp.items()}
        if self.raw is not None:  # `raw` keeps every gene; it follows the cells only
            r = self.raw
            sub.raw = r if all_obs else AnnData(None if r.X is None else r.X[oi], sub.obs, r.var, varm=r.varm)
        sub.is_view = True
        return sub

    def __repr__(self) -> str:
        return (f"AnnData object with n_obs x n_vars = {self.n_obs} x {self.n_vars}\n    obs: {list(self.obs.columns)}\n"
                f"    var: {list(self.var.columns)}\n    uns: {list(self.uns)}\n    obsm: {list(self.obsm)}\n"
                f"    varm: {list(self.varm)}\n    obsp: {list(self.obsp)}")


def is_anndata(obj) -> bool:
    return all(hasattr(obj, a) for a in ("obs", "var", "obsm", "obsp", "uns", "n_obs", "n_vars"))
