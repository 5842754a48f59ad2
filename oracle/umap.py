"""TEST INFRASTRUCTURE -- CPU restatement of what `sc.tl.umap` computes (src/scanpy/tools/_umap.py:150-229): umap-learn
0.5.x `find_ab_params` + `simplicial_set_embedding` (spectral initialisation, epoch schedule, SGD of oracle/umap.c).

PARITY UNPINNED: umap-learn is not installed here (pyproject.toml:76 pins `umap-learn>=0.5.12`), the reference's tests
only smoke-test `sc.tl.umap` (tests/test_embedding.py:55-95) and the algorithm is stochastic; the restatement follows the
published algorithm and the package source from memory.  The GPU kernel is checked (a) element-wise against
`scheme='synchronous'` (the same Jacobi scheme on the CPU) and (b) for layout quality against `scheme='sequential'`
(the reference's sweep): fuzzy-set cross entropy, trustworthiness, separation of planted clusters.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
from scipy import sparse

HERE = Path(__file__).resolve().parent
LIB = HERE / "_lib" / "liboracle_umap.so"
_lib = None


def build() -> Path:
    if not LIB.exists() or LIB.stat().st_mtime < (HERE / "umap.c").stat().st_mtime:
        subprocess.run(["make", "-s", "-C", str(HERE), "_lib/liboracle_umap.so"], check=True)
    return LIB


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(str(build()))
        lib.oracle_umap_sequential.restype = C.c_int
        lib.oracle_umap_sequential.argtypes = [C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                               C.c_void_p]
        lib.oracle_umap_synchronous.restype = C.c_int
        lib.oracle_umap_synchronous.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                                C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_void_p]
        _lib = lib
    return _lib


def find_ab_params(spread: float = 1.0, min_dist: float = 0.5):
    """umap.umap_.find_ab_params: least-squares fit of 1 / (1 + a x^(2b)) to the offset exponential."""
    from scipy.optimize import curve_fit

    def curve(x, a, b):
        return 1.0 / (1.0 + a * x ** (2 * b))

    xv = np.linspace(0, spread * 3, 300)
    yv = np.zeros(xv.shape)
    yv[xv < min_dist] = 1.0
    yv[xv >= min_dist] = np.exp(-(xv[xv >= min_dist] - min_dist) / spread)
    params, _ = curve_fit(curve, xv, yv)
    return float(params[0]), float(params[1])


def prune_graph(graph, n_epochs: int):
    """simplicial_set_embedding: entries that would be sampled less than once are dropped; -> COO (row-major order)."""
    g = sparse.coo_matrix(graph).copy()
    g.sum_duplicates()
    default_epochs = 500 if g.shape[0] <= 10000 else 200
    cut = g.data.max() / float(n_epochs if n_epochs > 10 else default_epochs)
    g.data[g.data < cut] = 0.0
    g.eliminate_zeros()
    return g


def make_epochs_per_sample(weights: np.ndarray, n_epochs: int) -> np.ndarray:
    """umap.umap_.make_epochs_per_sample"""
    result = -1.0 * np.ones(weights.shape[0], dtype=np.float64)
    n_samples = n_epochs * (weights / weights.max())
    result[n_samples > 0] = float(n_epochs) / n_samples[n_samples > 0]
    return result


def spectral_layout(graph, dim: int) -> np.ndarray:
    """umap.spectral.spectral_layout for a connected graph: eigenvectors 1..dim of the symmetric normalised Laplacian."""
    from scipy.sparse.csgraph import connected_components
    from scipy.sparse.linalg import eigsh

    g = sparse.csr_matrix(graph).astype(np.float64)
    n = g.shape[0]
    if connected_components(g)[0] > 1:
        raise NotImplementedError("oracle: multi-component spectral layout is not restated")
    deg = np.asarray(g.sum(axis=0)).ravel()
    dm = sparse.diags(1.0 / np.sqrt(deg))
    lap = sparse.identity(n) - dm @ g @ dm
    k = dim + 1
    ncv = max(2 * k + 1, int(np.sqrt(n)))
    vals, vecs = eigsh(lap, k, which="SM", ncv=ncv, tol=1e-4, v0=np.ones(n), maxiter=n * 5)
    order = np.argsort(vals)[1:k]
    return vecs[:, order]


def initial_embedding(graph, dim: int, init, rs: np.random.RandomState) -> np.ndarray:
    """the `init` branch of simplicial_set_embedding + the rescaling to [0, 10] per dimension"""
    n = graph.shape[0]
    if isinstance(init, str) and init == "random":
        emb = rs.uniform(low=-10.0, high=10.0, size=(n, dim)).astype(np.float32)
    elif isinstance(init, str) and init == "spectral":
        ini = spectral_layout(graph, dim)
        expansion = 10.0 / np.abs(ini).max()
        emb = (ini * expansion).astype(np.float32) + rs.normal(scale=0.0001, size=[n, dim]).astype(np.float32)
    else:
        emb = np.array(init, dtype=np.float32)
    emb = (10.0 * (emb - emb.min(0)) / (emb.max(0) - emb.min(0))).astype(np.float32, order="C")
    return emb


def simplicial_set_embedding(graph, *, n_components=2, initial_alpha=1.0, a=None, b=None, gamma=1.0,
                             negative_sample_rate=5, n_epochs=None, init="spectral", seed=0, scheme="sequential"):
    """-> embedding float32 [n, n_components].  scheme 'sequential' = the reference; 'synchronous' = csrc/umap.hip's."""
    if a is None or b is None:
        a, b = find_ab_params(1.0, 0.5)
    n = graph.shape[0]
    if n_epochs is None:
        n_epochs = 500 if n <= 10000 else 200
    g = prune_graph(graph, n_epochs)
    rs = np.random.RandomState(seed)
    y = initial_embedding(g, n_components, init, rs)
    rng_state = rs.randint(np.iinfo(np.int32).min + 1, np.iinfo(np.int32).max - 1, 3).astype(np.int64)
    y = optimize_layout(g, y, n_epochs=n_epochs, a=a, b=b, gamma=gamma, initial_alpha=initial_alpha,
                        negative_sample_rate=negative_sample_rate, rng_state=rng_state, seed=seed, scheme=scheme)
    return y


def optimize_layout(g, y0, *, n_epochs, a, b, gamma=1.0, initial_alpha=1.0, negative_sample_rate=5, rng_state=None,
                    seed=0, scheme="sequential") -> np.ndarray:
    """g: pruned symmetric graph; y0 float32 [n, dim] scaled to [0, 10]."""
    lib = _load()
    y = np.ascontiguousarray(y0, dtype=np.float32).copy()
    n, dim = y.shape
    if scheme == "sequential":
        coo = sparse.coo_matrix(g)
        eps = make_epochs_per_sample(coo.data, n_epochs).astype(np.float32)
        head = np.ascontiguousarray(coo.row, dtype=np.int32)
        tail = np.ascontiguousarray(coo.col, dtype=np.int32)
        rstate = np.ascontiguousarray(rng_state if rng_state is not None else [1, 2, 3], dtype=np.int64)
        rc = lib.oracle_umap_sequential(n, dim, len(head), head.ctypes.data, tail.ctypes.data, eps.ctypes.data,
                                        int(n_epochs), a, b, gamma, initial_alpha, float(negative_sample_rate),
                                        rstate.ctypes.data, y.ctypes.data)
    elif scheme == "synchronous":
        csr = sparse.csr_matrix(g)
        csr.sort_indices()
        eps = make_epochs_per_sample(csr.data, n_epochs).astype(np.float32)
        indptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(csr.indices, dtype=np.int32)
        rc = lib.oracle_umap_synchronous(n, dim, indptr.ctypes.data, indices.ctypes.data, eps.ctypes.data, int(n_epochs),
                                         a, b, gamma, initial_alpha, float(negative_sample_rate), int(seed) & (2**64 - 1),
                                         y.ctypes.data)
    else:
        raise ValueError(scheme)
    if rc != 0:
        raise RuntimeError(f"oracle umap failed: {rc}")
    return y


# ---- layout quality (what the tests compare between the GPU result and the reference scheme) ---------------------
def fuzzy_cross_entropy(graph, y, a, b, *, n_neg=20, seed=0) -> float:
    """Monte-Carlo estimate of UMAP's objective: -sum_edges w log q - sum_sampled_non_edges log(1 - q),
    q = 1 / (1 + a d^(2b)), per vertex (float64)."""
    g = sparse.coo_matrix(graph)
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[0]
    d2 = ((y[g.row] - y[g.col]) ** 2).sum(axis=1)
    q = 1.0 / (1.0 + a * d2 ** b)
    attract = -(g.data * np.log(np.clip(q, 1e-12, 1.0))).sum()
    rng = np.random.default_rng(seed)
    i = np.repeat(np.arange(n), n_neg)
    j = rng.integers(0, n, size=i.size)
    keep = i != j
    d2 = ((y[i[keep]] - y[j[keep]]) ** 2).sum(axis=1)
    q = 1.0 / (1.0 + a * d2 ** b)
    repel = -np.log(np.clip(1.0 - q, 1e-12, 1.0)).sum() * (n / n_neg) / n
    return float((attract + repel) / n)
