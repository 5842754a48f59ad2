"""`sc.tl.leiden` on MI355X (src/scanpy/tools/_leiden.py:55-228): same signature, flavor validation,
`restrict_to` semantics and write-back; the optimiser is `scamd_leiden_csr_f32`."""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd
from scipy import sparse

from .._anndata import is_anndata
from .._settings import settings
from .._utils import _UNSET, choose_graph, natsorted_str, resolve_seed

_DEFAULT = object()


def _validate_flavor(flavor, *, partition_type, directed):
    """src/scanpy/tools/_leiden.py:231-268."""
    if was_default := (flavor is None or flavor is _DEFAULT):
        flavor = settings.leiden_flavor
    if flavor == "igraph":
        if directed:
            msg = "Cannot use igraph’s leiden implementation with a directed graph."
            raise ValueError(msg)
        if partition_type is not None:
            msg = "Do not pass in partition_type argument when using igraph."
            raise ValueError(msg)
    elif flavor == "leidenalg":
        if was_default:
            msg = ("In the future, the default backend for leiden will be igraph instead of leidenalg. "
                   "To achieve the future defaults please pass: `flavor='igraph'` and `n_iterations=2`. "
                   "`directed` must also be `False` to work with igraph’s implementation.")
            warnings.warn(msg, FutureWarning, stacklevel=3)
    else:
        msg = f"flavor must be either 'igraph' or 'leidenalg', but {flavor!r} was passed."
        raise ValueError(msg)
    return flavor


# leidenalg partition classes the device optimiser covers, by class name (leidenalg is absent from this image; on a user's
# machine `partition_type=leidenalg.CPMVertexPartition` arrives as the class): -> (objective, takes a resolution_parameter)
_PARTITION_TYPES = {
    "RBConfigurationVertexPartition": ("modularity", True),  # the reference's default, _leiden.py:174-175
    "ModularityVertexPartition": ("modularity", False),  # RBConfiguration at resolution 1, no resolution_parameter
    "CPMVertexPartition": ("cpm", True),  # node sizes 1: igraph's objective_function='CPM'
    # Reichardt-Bornholdt with the Erdos-Renyi null model: sum_ij (A_ij - gamma p n_i n_j) delta, p = the graph's density
    # (total weight / n (n - 1) on the directed graph the leidenalg flavor builds): CPM at the resolution gamma * p
    "RBERVertexPartition": ("cpm_density", True),
}


def _resolve_partition_type(partition_type, resolution):
    """`partition_type` of the leidenalg flavor (src/scanpy/tools/_leiden.py:107-110, 174-186) -> (objective, resolution).
    `leidenalg.find_partition` hands `resolution_parameter` to the class's constructor, so a class without one fails with a
    TypeError unless `resolution=None` (the reference's docstring: "Set to `None` if overriding `partition_type`")."""
    if partition_type is None:
        return "modularity", 1.0 if resolution is None else resolution
    name = partition_type if isinstance(partition_type, str) else getattr(partition_type, "__name__", type(partition_type).__name__)
    if name not in _PARTITION_TYPES:
        raise NotImplementedError(f"partition_type {name!r} is not built on the MI355X path (have: {sorted(_PARTITION_TYPES)})")
    objective, takes_resolution = _PARTITION_TYPES[name]
    if not takes_resolution:
        if resolution is not None:
            raise TypeError(f"{name}.__init__() got an unexpected keyword argument 'resolution_parameter' "
                            "(pass resolution=None with this partition_type)")
        return objective, 1.0
    return objective, 1.0 if resolution is None else resolution


def restrict_adjacency(adata, restrict_key, *, restrict_categories, adjacency):
    """src/scanpy/tools/_utils_clustering.py:33-50."""
    if not isinstance(restrict_categories[0], str):
        msg = "You need to use strings to label categories, e.g. '1' instead of 1."
        raise ValueError(msg)
    for c in restrict_categories:
        if c not in adata.obs[restrict_key].cat.categories:
            msg = f"{c!r} is not a valid category for {restrict_key!r}"
            raise ValueError(msg)
    restrict_indices = adata.obs[restrict_key].isin(restrict_categories).to_numpy()
    adjacency = adjacency[restrict_indices, :]
    adjacency = adjacency[:, restrict_indices]
    return adjacency, restrict_indices


def rename_groups(adata, restrict_key, *, key_added, restrict_categories, restrict_indices, groups):
    """src/scanpy/tools/_utils_clustering.py:16-30."""
    key_added = f"{restrict_key}_R" if key_added is None else key_added
    all_groups = adata.obs[restrict_key].astype("U")
    prefix = f"{'-'.join(restrict_categories)},"
    new_groups = [prefix + g for g in groups.astype("U")]
    all_groups.iloc[restrict_indices] = new_groups
    return all_groups


def _warn_unfinished(stats: dict, n_iterations: int) -> None:
    """`n_iterations < 0` promises a run "until it reaches an iteration that does not improve the clustering"
    (src/scanpy/tools/_leiden.py:65, 166).  The device optimiser bounds that loop (csrc/leiden.hip: MAX_OUTER_ITERS) and its
    final polish (MAX_POLISH_ROUNDS); when either bound ends a run the caller is told -- the partition returned is still
    the best one seen, connected, and (unless the polish itself was cut short) node optimal."""
    if n_iterations >= 0:
        return
    if stats.get("ended_by_iteration_cap"):
        warnings.warn(
            f"leiden(n_iterations={n_iterations}): every one of the {stats['iterations']} iterations still improved the "
            "quality; the run was stopped at the iteration cap of the MI355X optimiser (SCAMD_LEIDEN_ITER_CAP) and the best "
            "partition, polished to node optimality, is returned.  Graphs without clear community structure behave like "
            "this in the reference as well (dozens of iterations with gains of 1e-5 and less); pass n_iterations=2, the "
            "reference's recommendation for flavor='igraph', for a bounded run.",
            UserWarning, stacklevel=4)
    if stats.get("polish_ended_by_round_cap"):
        warnings.warn(
            "leiden: the final single-vertex polish stopped at its round cap; the partition may hold vertices that a "
            "move would still improve.", UserWarning, stacklevel=4)


def leiden_partition(adjacency, *, resolution=1.0, n_iterations=-1, seed=0, use_weights=True, beta=0.01,
                     initial_membership=None, objective="modularity", node_weights=None, report="objective"):
    """Symmetric adjacency (scipy sparse) -> (membership int32 [n], modularity).  `initial_membership`: a partition to
    start from (any non-negative integer labels, one per vertex), as leidenalg / igraph take it.
    `report`: which number comes back beside the labels -- 'objective': the optimiser's own (modularity at `resolution`;
    for CPM the weighted resolution-1 modularity); 'weighted' / 'unweighted': the plain resolution-1 modularity of the
    partition with / without the edge weights, what `part.modularity` is in the reference for the igraph / leidenalg
    flavor (see `leiden`)."""
    import torch

    from .. import _kernels
    from .._device import require_gpu

    dev = require_gpu()
    # (a CSR matrix is used as it is: re-wrapping it would drop its cached `has_sorted_indices`, an O(nnz) check)
    adj = adjacency if sparse.isspmatrix_csr(adjacency) else sparse.csr_matrix(adjacency)
    if not adj.has_sorted_indices:
        adj = adj.sorted_indices()
    n = adj.shape[0]
    indptr = torch.from_numpy(np.ascontiguousarray(adj.indptr, dtype=np.int64)).to(dev)
    from .._device import pinned_uploader

    indices = pinned_uploader.upload(np.ascontiguousarray(adj.indices, dtype=np.int32), dev)
    w = adj.data if use_weights else np.ones_like(adj.data)
    weights = pinned_uploader.upload(np.ascontiguousarray(w, dtype=np.float32), dev)
    init = None
    if initial_membership is not None:
        labels = np.asarray(initial_membership)
        if labels.shape != (n,) or not np.issubdtype(labels.dtype, np.integer) or (n and labels.min() < 0):
            raise ValueError(f"initial_membership must hold one non-negative integer per vertex ({n})")
        # (the kernel wants ids below n: any labelling is renamed to consecutive ids first)
        init = torch.from_numpy(np.unique(labels, return_inverse=True)[1].astype(np.int32)).to(dev)
    nw = None
    if node_weights is not None:
        # igraph's `node_weights` (a vertex attribute name or a list): one finite non-negative number per vertex
        arr = np.asarray(node_weights, dtype=np.float64)
        if arr.shape != (n,) or not np.isfinite(arr).all() or (n and (arr.min() < 0 or arr.max() > 1e6)):
            raise ValueError(f"node_weights must hold one number in [0, 1e6] per vertex ({n})")
        nw = torch.from_numpy(arr.astype(np.float32)).to(dev)
    memb, q, _ = _kernels.leiden(indptr, indices, weights, n, resolution=float(resolution),
                                 n_iterations=int(n_iterations), beta=beta, seed=int(seed), initial_membership=init,
                                 objective=objective, node_weights=nw)
    _warn_unfinished(_kernels.leiden_last_stats(), int(n_iterations))
    if report == "unweighted" and use_weights:
        q = _kernels.modularity(indptr, indices, torch.ones_like(weights), n, memb, resolution=1.0)
    elif report in ("weighted", "unweighted") and objective == "modularity" and float(resolution) != 1.0:
        q = _kernels.modularity(indptr, indices, weights, n, memb, resolution=1.0)
    return memb.cpu().numpy(), q


def leiden(  # noqa: PLR0913
    adata,
    resolution: float = 1,
    *,
    restrict_to=None,
    rng=None,
    random_state=_UNSET,
    key_added: str = "leiden",
    adjacency=None,
    directed: bool | None = None,
    use_weights: bool = True,
    n_iterations: int = -1,
    partition_type=None,
    neighbors_key: str | None = None,
    obsp: str | None = None,
    copy: bool = False,
    flavor=_DEFAULT,
    **clustering_args,
):
    """Cluster cells with the Leiden algorithm (drop-in for `scanpy.tl.leiden`, src/scanpy/tools/_leiden.py:55).

    Both reference flavors optimise the same objective on the symmetric connectivities
    (RBConfiguration modularity with `resolution`) and are served by the same GPU
    optimiser.  `objective_function='CPM'` (igraph) and `partition_type=` RBConfiguration / Modularity / CPM
    VertexPartition (leidenalg; matched by class name, `_PARTITION_TYPES`) select the objective of that optimiser.

    The two flavors do NOT see the same graph in the reference, and CPM -- unlike modularity -- is not scale invariant:
    under the default preset `get_igraph_from_adjacency` adds every STORED entry of the symmetric matrix as an undirected
    edge (src/scanpy/_utils/__init__.py:292-298), so igraph's optimiser sees every pair twice, `A_ij = 2 w_ij`, and
    `flavor='igraph', objective_function='CPM'` maximises `sum (2 w_ij - resolution n_i n_j)` -- `resolution / 2` on the
    matrix itself; the V2 preset builds the graph with `Weighted_Adjacency(mode=undirected)` (:285-290: every pair once).
    The leidenalg flavor gets a directed graph with both directions (`_leiden.py:172-173`): `CPMVertexPartition` maximises
    `sum (w_ij - resolution n_i n_j)` over the ordered pairs -- `resolution` as given.  `settings.preset` selects V1 / V2.

    Writes `.obs[key_added]` (categorical of str, naturally sorted categories, ids by decreasing community size) and
    `.uns[key_added] = {params, modularity}`; `modularity` is what the reference stores, `part.modularity` (:219): igraph's
    `VertexClustering.modularity` recomputed from the membership at resolution 1 -- with the edge weights for the igraph
    flavor (`community_leiden` hands `modularity_params=dict(weights=...)` to the clustering), WITHOUT them for the
    leidenalg flavor (`MutableVertexPartition` constructs the clustering with no modularity parameters)."""
    if not is_anndata(adata):
        raise TypeError("leiden() expects an AnnData-like object")
    flavor = _validate_flavor(flavor, partition_type=partition_type, directed=directed)
    seed, meta_random_state = resolve_seed(rng, random_state)
    unknown = set(clustering_args) - {"objective_function", "weights", "beta", "initial_membership", "node_weights", "node_sizes"}
    if unknown:
        raise TypeError(f"leiden() got unexpected clustering arguments {sorted(unknown)}")
    objective = str(clustering_args.get("objective_function", "modularity")).lower()
    if objective not in ("modularity", "cpm"):  # (igraph's own message, Graph.community_leiden)
        raise ValueError('objective_function must be "CPM" or "modularity".')
    if flavor == "leidenalg":
        if "objective_function" in clustering_args:
            # (the reference hands `objective_function` to igraph only, _leiden.py:188-196; leidenalg's find_partition would
            # pass it on to the partition class, which does not take it)
            raise TypeError("objective_function is igraph's argument: pass flavor='igraph', or "
                            "partition_type=leidenalg.CPMVertexPartition with flavor='leidenalg'")
        objective, gamma = _resolve_partition_type(partition_type, resolution)
    else:
        gamma = 1.0 if resolution is None else resolution  # (igraph's default when `resolution` is left out, _leiden.py:193-194)
        if objective == "cpm" and settings.preset == "ScanpyV1":
            gamma = gamma / 2.0  # every symmetric pair is an edge TWICE in the V1 graph (_utils/__init__.py:292-298)
    # vertex weights of the quality function: igraph calls them `node_weights`, leidenalg's partition classes `node_sizes`
    mine, other = ("node_sizes", "node_weights") if flavor == "leidenalg" else ("node_weights", "node_sizes")
    if clustering_args.get(other) is not None:
        raise TypeError(f"{other} is not an argument of the {flavor} flavor: it takes {mine}")
    node_weights = clustering_args.get(mine)
    if node_weights is not None:
        # (igraph: with the modularity objective the vertex weights default to the strengths and the resolution is divided by
        # 2m; what it does with OTHER weights there is not pinned by anything in the reference -- CPM's meaning is plain)
        if objective not in ("cpm", "cpm_density"):
            raise NotImplementedError(f"{mine} are taken with the CPM objective only on the MI355X path")
        if restrict_to is not None:
            raise NotImplementedError(f"{mine} together with restrict_to is not supported on the MI355X path")
        if isinstance(node_weights, str):
            raise NotImplementedError(f"{mine} as an igraph vertex attribute name: pass the numbers (one per cell)")
    initial_membership = clustering_args.get("initial_membership")
    if initial_membership is not None and restrict_to is not None:
        raise NotImplementedError("initial_membership together with restrict_to is not supported on the MI355X path")
    adata = adata.copy() if copy else adata
    if adjacency is None:
        adjacency = choose_graph(adata, obsp, neighbors_key)
    if restrict_to is not None:
        restrict_key, restrict_categories = restrict_to
        adjacency, restrict_indices = restrict_adjacency(
            adata, restrict_key, restrict_categories=restrict_categories, adjacency=adjacency)
    if objective == "cpm_density":  # RBERVertexPartition: the density of the graph that is clustered (after restrict_to)
        # (leidenalg's Graph::density: total edge weight / (N (N - 1)) on a directed graph, N = the total node size)
        n_v = float(adjacency.shape[0]) if node_weights is None else float(np.sum(np.asarray(node_weights, dtype=np.float64)))
        total = float(adjacency.sum()) if use_weights else float(adjacency.nnz)
        gamma = gamma * total / max(n_v * (n_v - 1.0), 1.0)
        objective = "cpm"
    groups, modularity = leiden_partition(adjacency, resolution=gamma, n_iterations=n_iterations, seed=seed,
                                          use_weights=use_weights, beta=clustering_args.get("beta", 0.01),
                                          initial_membership=initial_membership, objective=objective,
                                          node_weights=node_weights,
                                          report="weighted" if flavor == "igraph" else "unweighted")
    if restrict_to is not None:
        if key_added == "leiden":
            key_added += "_R"
        groups = rename_groups(adata, key_added=key_added, restrict_key=restrict_key,
                               restrict_categories=restrict_categories, restrict_indices=restrict_indices,
                               groups=groups)
    groups = np.asarray(groups)
    if np.issubdtype(groups.dtype, np.integer) and (groups.size == 0 or groups.min() >= 0):
        # same Categorical as the general branch below (`_leiden.py:210-213`): for non-negative integers the natural
        # order of the label strings is the numeric order, so the codes come straight from a searchsorted instead of
        # a million int -> str conversions (~100 ms at 1M cells)
        uniq = np.unique(groups)
        # (the optimiser numbers communities 0..C-1: then the labels are their own codes)
        codes = groups if uniq.size and uniq[0] == 0 and uniq[-1] == uniq.size - 1 else np.searchsorted(uniq, groups)
        adata.obs[key_added] = pd.Categorical.from_codes(codes, categories=[str(u) for u in uniq])
    else:
        adata.obs[key_added] = pd.Categorical(
            values=groups.astype("U"),
            categories=natsorted_str(list(map(str, np.unique(groups)))),
        )
    adata.uns[key_added] = {}
    adata.uns[key_added]["params"] = dict(resolution=resolution, n_iterations=n_iterations, **meta_random_state)
    adata.uns[key_added]["modularity"] = modularity
    return adata if copy else None
