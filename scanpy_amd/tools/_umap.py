"""`sc.tl.umap` on MI355X: signature, slots and parameter handling of the reference
(src/scanpy/tools/_umap.py:30-229); the layout optimisation is `scamd_umap_optimize_f32` (csrc/umap.hip).

What the reference delegates to umap-learn's `simplicial_set_embedding` is restated here around the device kernel:
`find_ab_params`, pruning of rarely sampled edges, `make_epochs_per_sample`, the initial embedding (spectral / random
/ given coordinates) and its rescaling to [0, 10].  Deviations, all confined to HOW the same objective is optimised:
  * the SGD is synchronous and race-free (see csrc/umap.hip) and draws negatives with a counter-based hash: the result
    is bitwise reproducible for a given seed, but it is not the sequence umap-learn's sequential sweep would produce;
  * `init_pos='spectral'` computes the leading eigenvectors of the symmetric normalised adjacency by Chebyshev-filtered
    subspace iteration to a residual of 2e-6 on the device (SpMM kernel of the PCA
    stage; orchestrated with torch.linalg on 8-column blocks); a disconnected graph is not laid out component by
    component -- the block iteration separates the components by itself;
  * `init_pos='paga'` places every cell around its group's node of a PAGA layout the caller brings along in
    `adata.uns['paga']` (`pos`, `groups`, `connectivities`: written by upstream `sc.tl.paga` + `sc.pl.paga`, which stay
    outside this path) -- `init_pos_from_paga`, after src/scanpy/tools/_utils.py:81-114."""
from __future__ import annotations

import logging

import numpy as np
from scipy import sparse

from .._anndata import is_anndata
from .._utils import _UNSET, choose_graph, resolve_seed

_log = logging.getLogger("scanpy_amd")


def find_ab_params(spread: float, min_dist: float):
    """umap.umap_.find_ab_params: fit 1 / (1 + a x^(2b)) to the offset exponential membership curve."""
    from scipy.optimize import curve_fit

    def curve(x, a, b):
        return 1.0 / (1.0 + a * x ** (2 * b))

    xv = np.linspace(0, spread * 3, 300)
    yv = np.zeros(xv.shape)
    yv[xv < min_dist] = 1.0
    yv[xv >= min_dist] = np.exp(-(xv[xv >= min_dist] - min_dist) / spread)
    params, _ = curve_fit(curve, xv, yv)
    return params[0], params[1]


def _prune_and_schedule(graph, n_epochs: int):
    """simplicial_set_embedding: drop the entries that would be sampled less than once, then
    make_epochs_per_sample.  -> CSR (sorted), epochs_per_sample float32 aligned with its data."""
    g = sparse.coo_matrix(graph).copy()
    g.sum_duplicates()
    default_epochs = 500 if g.shape[0] <= 10000 else 200
    cut = g.data.max() / float(n_epochs if n_epochs > 10 else default_epochs)
    g.data[g.data < cut] = 0.0
    g.eliminate_zeros()
    csr = g.tocsr()
    csr.sort_indices()
    w = csr.data.astype(np.float64)
    eps = -1.0 * np.ones(w.shape[0], dtype=np.float64)
    n_samples = n_epochs * (w / w.max())
    eps[n_samples > 0] = float(n_epochs) / n_samples[n_samples > 0]
    return csr, eps.astype(np.float32)


def prune_and_schedule_device(indptr, indices, data, n: int, n_epochs: int):
    """`_prune_and_schedule` on device tensors (the graph never leaves HBM in the resident pipeline).
    -> (indptr int64, indices int32, weights float32, epochs_per_sample float32)."""
    import torch

    default_epochs = 500 if n <= 10000 else 200
    wmax = data.max()
    keep = (data >= wmax / float(n_epochs if n_epochs > 10 else default_epochs)) & (data != 0)
    rows = torch.repeat_interleave(torch.arange(n, device=data.device), (indptr[1:] - indptr[:-1]))
    counts = torch.bincount(rows[keep], minlength=n)
    new_indptr = torch.zeros(n + 1, dtype=torch.int64, device=data.device)
    new_indptr[1:] = torch.cumsum(counts, 0)
    w = data[keep].contiguous()
    n_samples = n_epochs * (w.to(torch.float64) / wmax.to(torch.float64))
    eps = torch.where(n_samples > 0, float(n_epochs) / n_samples, torch.full_like(n_samples, -1.0)).to(torch.float32)
    return new_indptr, indices[keep].contiguous(), w, eps.contiguous()


def _top_eigenvectors_below_trivial(apply_s, trivial, dim: int, seed: int, *, tol: float = 2e-6, max_outer: int = 60,
                                    max_degree: int = 64, info: dict | None = None):
    """The `dim` eigenvectors of a symmetric operator S (spectrum in [-1, 1], largest eigenvalue 1 with the KNOWN
    eigenvector `trivial`) that follow the trivial one, by Chebyshev-filtered subspace iteration (Zhou & Saad) on
    M = (S + I) / 2 with the trivial direction projected out.

    `apply_s(V)` -> S V for a float64 [n, b] block.  Block power iteration -- what this replaced in round 5 -- converges
    like (lambda_{b+1} / lambda_j)^iterations: on a graph WITHOUT separated clusters (a sheet, a trajectory: eigenvalues
    1 - O(1e-3)) fifty iterations leave a random block, and `init_pos='spectral'` silently became a random start (found
    by the layout-quality test against the sequential oracle: graph neighbours kept 0.49 instead of 0.97).  The filter of
    degree m damps everything below the block's smallest Ritz value c by T_m((2 lambda - c) / c) ~ exp(m sqrt(8 (lambda -
    c) / c)) per outer iteration -- the square-root law of Lanczos, which is what the reference's solver (ARPACK through
    umap-learn's `spectral_layout`) has.  Stops on the residual of the wanted Ritz pairs: 2e-6, what a float32 operand of the
    SpMM allows and two orders below the eigenvalue gaps of a 100k-vertex sheet (at 1e-4 the plane it returned was rotated
    into the next eigenvectors: cosines 0.87 / 0.81 against ARPACK's)."""
    import torch

    n = trivial.shape[0]
    dev = trivial.device
    b = dim + 6
    t0 = (trivial / torch.linalg.norm(trivial)).to(torch.float64).reshape(n, 1)

    def tall_gram(p, q):
        """p^T q for very tall, very thin blocks: a batched product over 1024 row chunks + a sum (one GEMM with M = N = 8
        and K = 1e6 runs in a single workgroup: tens of ms)"""
        c = 1024
        m = n // c
        if m == 0:
            return p.T @ q
        head = torch.bmm(p[: m * c].view(c, m, -1).transpose(1, 2), q[: m * c].view(c, m, -1)).sum(dim=0)
        return head + p[m * c:].T @ q[m * c:]

    def deflate(y):
        return y - t0 @ tall_gram(t0, y)

    def cholqr2(y):
        y = y / torch.sqrt(torch.diagonal(tall_gram(y, y))).clamp_min(1e-300)
        for _ in range(2):
            l, bad = torch.linalg.cholesky_ex(tall_gram(y, y))
            if int(bad) != 0:
                return torch.linalg.qr(y, mode="reduced")[0]
            eye = torch.eye(l.shape[0], dtype=l.dtype, device=l.device)
            y = y @ torch.linalg.solve_triangular(l, eye, upper=False).T  # (tiny inverse formed explicitly: a trsm with a
        return y                                                           #  7-row factor and 1e6 right-hand sides took 55 ms)

    def apply_m(y):
        return 0.5 * (apply_s(y) + y)

    def rayleigh_ritz(z):
        mz = apply_m(z)
        t = tall_gram(z, mz)
        theta, w = torch.linalg.eigh(0.5 * (t + t.T))
        theta, w = theta.flip(0), w.flip(1)
        return theta, z @ w, mz @ w

    gen = torch.Generator(device="cpu").manual_seed(int(seed) & 0x7FFFFFFF)
    z = cholqr2(deflate(torch.randn((n, b), generator=gen, dtype=torch.float64).to(dev)))
    theta, v, mv = rayleigh_ritz(z)
    n_apply, resid, outer = 1, float("inf"), 0
    for outer in range(1, max_outer + 1):
        r = mv[:, :dim] - v[:, :dim] * theta[None, :dim]
        resid = float(torch.linalg.norm(r, dim=0).max())  # (|M| = 1: absolute = relative)
        if resid < tol:
            break
        c = float(theta[-1])
        if not 0.0 < c < 1.0:  # (a degenerate block: one plain step keeps it simple)
            theta, v, mv = rayleigh_ritz(cholqr2(deflate(mv)))
            n_apply += 1
            continue
        e = center = 0.5 * c
        # the degree: as high as the amplification SPREAD inside the wanted set allows (beyond ~1e9 every column is the
        # leading wanted vector plus rounding noise), the spectrum's upper end is 1
        x1, xk = (1.0 - center) / e, max((float(theta[dim - 1]) - center) / e, 1.0)
        spread = float(np.arccosh(x1) - np.arccosh(xk))
        m = max_degree if spread <= 0.0 else max(4, min(max_degree, int(np.floor(20.7 / spread))))
        sigma = e / (1.0 - center)
        sigma1 = sigma
        y_prev = v
        y = (mv - center * v) * (sigma1 / e)
        for _ in range(2, m + 1):
            sigma2 = 1.0 / (2.0 / sigma1 - sigma)
            y_new = (apply_m(y) - center * y) * (2.0 * sigma2 / e) - (sigma * sigma2) * y_prev
            y_prev, y, sigma = y, y_new, sigma2
        n_apply += m - 1
        theta, v, mv = rayleigh_ritz(cholqr2(deflate(y)))
        n_apply += 1
    if info is not None:
        info.update(outer_iterations=outer, operator_applications=n_apply, residual=resid, converged=bool(resid < tol),
                    ritz_values=[float(2.0 * t - 1.0) for t in theta[:dim]])
    return v[:, :dim]


def _spectral_init(indptr, indices, weights, n: int, dim: int, seed: int, *, info: dict | None = None):
    """Leading non-trivial eigenvectors of S = D^-1/2 A D^-1/2 (= the smallest of the normalised Laplacian, what
    umap.spectral.spectral_layout asks ARPACK for).  Device tensors in, float64 [n, dim] host array out."""
    import torch

    from .. import _kernels as K

    dev = weights.device
    if dev.type == "cuda" and dim <= 10:
        # the product path: ONE C call on the kernels of csrc/dense.hip (round 6; until then the block iteration below ran
        # on torch.linalg QR / Cholesky / eigh -- rocSOLVER -- and torch.bmm; it remains what the CPU tests run)
        from .._lib import ScamdError

        try:
            vec, dinfo = K.spectral_embedding(indptr, indices, weights, n, dim, seed=seed)
            if info is not None:
                info.update(dinfo)
            return vec.cpu().numpy()
        except ScamdError as exc:  # (the block could not be orthonormalised: the caller falls back to a random layout)
            if info is not None:
                info.update(converged=False, residual=float("nan"), outer_iterations=0, operator_applications=0, error=str(exc))
            return np.full((n, dim), np.nan)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]))
    deg = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows, weights.to(torch.float64))
    dis = torch.where(deg > 0, deg.rsqrt(), torch.zeros_like(deg))
    s_data = (weights.to(torch.float64) * dis[rows] * dis[indices.long()]).to(torch.float32).contiguous()

    def apply_s(y):
        return K.spmm(indptr, indices, s_data, n, n, y.to(torch.float32).contiguous()).to(torch.float64)

    vec = _top_eigenvectors_below_trivial(apply_s, torch.sqrt(deg), dim, seed, info=info)
    return vec.cpu().numpy()


def umap_embedding(connectivities, *, n_components=2, n_epochs=None, a, b, gamma=1.0, initial_alpha=1.0,
                   negative_sample_rate=5, init="spectral", seed=0) -> np.ndarray:
    """The device part of `simplicial_set_embedding`: symmetric fuzzy graph -> embedding float32 [n, n_components]."""
    import torch

    from .. import _kernels as K
    from .._device import require_gpu

    dev = require_gpu()
    n = connectivities.shape[0]
    if n_epochs is None:
        n_epochs = 500 if n <= 10000 else 200
    # the graph goes to the device as it is stored (CSR); pruning and the epoch schedule are computed there
    # (`prune_and_schedule_device` == `_prune_and_schedule`, which a COO round trip through scipy makes 50x slower than
    # the whole optimisation at 1M cells)
    # (a CSR matrix is used as it is: re-wrapping would drop the canonical-format flag `pp.neighbors` recorded)
    csr = connectivities if sparse.isspmatrix_csr(connectivities) else sparse.csr_matrix(connectivities)
    if not csr.has_canonical_format:
        csr = csr.copy()
        csr.sum_duplicates()
    indptr, indices, weights, eps_t = prune_and_schedule_device(
        torch.from_numpy(csr.indptr.astype(np.int64)).to(dev), torch.from_numpy(csr.indices.astype(np.int32)).to(dev),
        torch.from_numpy(np.ascontiguousarray(csr.data, dtype=np.float32)).to(dev), n, n_epochs)
    rs = np.random.RandomState(seed)
    if isinstance(init, str) and init == "random":
        emb = rs.uniform(low=-10.0, high=10.0, size=(n, n_components)).astype(np.float32)
    elif isinstance(init, str) and init == "spectral":
        sp_info: dict = {}
        ini = None
        if n > n_components + 6:
            ini = _spectral_init(indptr, indices, weights, n, n_components, seed, info=sp_info)
        if ini is None or not sp_info.get("converged", False) or not np.isfinite(ini).all():
            # umap-learn's spectral_layout falls back to a random layout with a warning when ARPACK does not converge
            # (umap/spectral.py); a graph too small for the block (n <= dim + 6) takes the same way out
            import warnings

            why = (f"the graph has only {n} vertices" if ini is None else
                   f"the Chebyshev-filtered subspace iteration stopped at residual {sp_info.get('residual', float('nan')):.2e} after "
                   f"{sp_info.get('outer_iterations')} outer iterations / {sp_info.get('operator_applications')} operator applications")
            warnings.warn(f"tl.umap: spectral initialisation failed ({why}); falling back to a random initialisation, as "
                          "umap-learn does when its eigensolver fails.", UserWarning, stacklevel=3)
            ini = rs.uniform(low=-10.0, high=10.0, size=(n, n_components))
        expansion = 10.0 / np.abs(ini).max()
        emb = (ini * expansion).astype(np.float32) + rs.normal(scale=0.0001, size=[n, n_components]).astype(np.float32)
    else:
        emb = np.array(init, dtype=np.float32)
        if emb.shape != (n, n_components):
            raise ValueError(f"init_pos has shape {emb.shape}, expected {(n, n_components)}")
    span = emb.max(0) - emb.min(0)
    emb = (10.0 * (emb - emb.min(0)) / np.where(span > 0, span, 1.0)).astype(np.float32, order="C")
    y = torch.from_numpy(emb).to(dev).contiguous()
    K.umap_optimize_(indptr, indices, eps_t, n, y, n_epochs=n_epochs, a=a, b=b, gamma=gamma,
                     initial_alpha=initial_alpha, negative_sample_rate=negative_sample_rate, seed=seed)
    return y.cpu().numpy()


def init_pos_from_paga(adata, *, seed: int = 0) -> np.ndarray:
    """Initial 2-D coordinates from a PAGA layout (src/scanpy/tools/_utils.py:81-114): a cell of group i starts at the
    group's node `pos[i]`, pulled half-way towards the node of the group i is most strongly connected to and jittered
    along that same direction -- `pos[i] - 0.5 d + u * d`, d = pos[i] - pos[nearest], u ~ U[0, 1)^2 from a generator of
    its own per group (`rng.spawn`) -- or exactly at the node when the group has no connection."""
    paga = adata.uns.get("paga", {}) if hasattr(adata.uns, "get") else {}
    if "pos" not in paga:
        raise ValueError("Plot PAGA first, so that `adata.uns['paga']['pos']` exists.")
    groups = adata.obs[paga["groups"]]
    cats = list(groups.cat.categories)
    codes = np.asarray(groups.cat.codes)
    node = np.asarray(paga["pos"], dtype=np.float64)
    coarse = paga["connectivities"]
    coarse = coarse.tocsr() if hasattr(coarse, "tocsr") else np.asarray(coarse)
    if len(cats) != node.shape[0]:
        raise ValueError(f"uns['paga']['pos'] holds {node.shape[0]} nodes for {len(cats)} groups")
    out = np.ones((adata.n_obs, 2), dtype=np.float64)
    streams = np.random.default_rng(seed).spawn(node.shape[0])
    for i, gen in enumerate(streams):
        members = np.flatnonzero(codes == i)
        row = coarse[i]
        row = np.asarray(row.todense()).ravel() if hasattr(row, "todense") else np.asarray(row).ravel()
        linked = np.flatnonzero(row)
        if linked.size == 0:
            out[members] = node[i]
            continue
        towards = node[i] - node[linked[np.argmax(row[linked])]]
        out[members] = node[i] - 0.5 * towards + gen.random((members.size, 2)) * towards
    return out


def umap(  # noqa: PLR0913
    adata,
    *,
    min_dist: float = 0.5,
    spread: float = 1.0,
    n_components: int = 2,
    maxiter: int | None = None,
    alpha: float = 1.0,
    gamma: float = 1.0,
    negative_sample_rate: int = 5,
    init_pos="spectral",
    rng=None,
    random_state=_UNSET,
    a: float | None = None,
    b: float | None = None,
    method: str = "umap",
    key_added: str | None = None,
    neighbors_key: str = "neighbors",
    copy: bool = False,
):
    """Embed the neighborhood graph using UMAP (drop-in for `scanpy.tl.umap`, `_umap.py:30`).

    Writes `obsm['X_umap' | key_added]` and `uns['umap' | key_added]['params'] = {a, b[, random_state]}`."""
    if not is_anndata(adata):
        raise TypeError("umap expects an AnnData object")
    seed, meta = resolve_seed(rng, random_state)
    adata = adata.copy() if copy else adata
    key_obsm, key_uns = ("X_umap", "umap") if key_added is None else (key_added, key_added)
    if neighbors_key is None:  # backwards compat (`:155-156`)
        neighbors_key = "neighbors"
    if neighbors_key not in adata.uns:
        raise ValueError(f"Did not find .uns[{neighbors_key!r}]. Run `sc.pp.neighbors` first.")
    if method != "umap":
        raise ValueError(f"Unknown method {method}")
    neighbors = adata.uns[neighbors_key]
    connectivities = choose_graph(adata, obsp=None, neighbors_key=neighbors_key)
    if "params" not in neighbors or neighbors["params"].get("method") != "umap":
        _log.warning('.obsp["%s"] have not been computed using umap', neighbors.get("connectivities_key", "connectivities"))
    if a is None or b is None:
        a, b = find_ab_params(spread, min_dist)
    adata.uns[key_uns] = dict(params=dict(a=a, b=b, **meta))
    if isinstance(init_pos, str) and init_pos in adata.obsm:
        init_coords = adata.obsm[init_pos]
    elif isinstance(init_pos, str) and init_pos == "paga":
        if n_components != 2:
            raise ValueError("init_pos='paga' gives 2-D coordinates: n_components must be 2")
        init_coords = init_pos_from_paga(adata, seed=seed)
    elif isinstance(init_pos, str) and init_pos not in {"spectral", "random"}:
        raise ValueError(f"init_pos={init_pos!r}: expected 'spectral', 'random', 'paga', a key of adata.obsm or an array")
    else:
        init_coords = init_pos
    if hasattr(init_coords, "dtype"):
        init_coords = np.asarray(init_coords, dtype=np.float32)  # `check_array(..., dtype=np.float32)`, `:187-188`
    default_epochs = 500 if connectivities.shape[0] <= 10000 else 200
    n_epochs = default_epochs if maxiter is None else maxiter
    adata.obsm[key_obsm] = umap_embedding(connectivities, n_components=n_components, n_epochs=n_epochs, a=a, b=b,
                                          gamma=gamma, initial_alpha=alpha, negative_sample_rate=negative_sample_rate,
                                          init=init_coords, seed=seed)
    return adata if copy else None
