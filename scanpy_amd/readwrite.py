"""`read_zarr` / `write_zarr` for AnnData `.zarr` stores (zarr format 3), with an out-of-core mode for `X`.

The reference re-exports anndata's functions (`src/scanpy/readwrite.py:23-25`; dispatch on the extension at `:837-841`,
writing at `:720-726`).  anndata and zarr-python are not in this image, so the on-disk element encodings are restated
from the anndata on-disk format specification ("encoding-type" / "encoding-version" attributes on every node):

    anndata      group {X, layers, obs, var, obsm, varm, obsp, varp, uns, raw}
    array        zarr array                 string-array   zarr array of data_type "string" (vlen-utf8)
    rec-array    zarr array of a `struct` data type
    csr_matrix / csc_matrix   group {data, indices, indptr}, attribute shape
    dataframe    group, attributes _index, column-order; one child per column + the index
    categorical  group {categories, codes}, attribute ordered
    dict         group                      numeric-scalar / string   0-d arrays
    nullable-integer / nullable-boolean     group {values, mask}

and pinned against the store that anndata itself wrote for the reference (`datasets/10x_pbmc68k_reduced.zarr.zip`,
tests/test_readwrite_zarr_cpu.py).  `backed='r'` leaves a CSR `X` on disk as a `_backed.BackedCsr`, which
`pp.pca` streams through the device by row chunks (the role of `anndata.experimental.read_elem_lazy` in
`docs/tutorials/experimental/dask.ipynb:843-879`).
"""
from __future__ import annotations

import os
import shutil
import warnings
from pathlib import Path

import numpy as np
import pandas as pd
from scipy import sparse

from . import _zarr3 as z3
from ._anndata import AnnData
from ._backed import BackedCsr, is_backed

# ---------------------------------------------------------------------------------------------------------------------
# reading


def _read_sparse(group):
    cls = sparse.csr_matrix if group.attrs["encoding-type"] == "csr_matrix" else sparse.csc_matrix
    shape = tuple(int(s) for s in group.attrs["shape"])
    return cls((group["data"].read(), group["indices"].read(), group["indptr"].read()), shape=shape)


def _read_dataframe(group) -> pd.DataFrame:
    index = read_elem(group[group.attrs["_index"]])
    cols = {}
    for name in group.attrs.get("column-order", []):
        col = read_elem(group[name])
        if col is not None:
            cols[name] = col
    df = pd.DataFrame(cols, index=pd.Index(np.asarray(index, dtype=object)))
    name = group.attrs["_index"]
    df.index.name = None if name in {"_index", "index"} else name
    return df


def _as_str(v):
    """HDF5 fixed-length strings come back as bytes: decode (zarr / variable-length strings already are str)"""
    if isinstance(v, np.ndarray) and v.dtype.kind == "S":
        return np.array([b.decode("utf-8", "replace") for b in v.reshape(-1).tolist()], dtype=object).reshape(v.shape)
    return v.decode("utf-8", "replace") if isinstance(v, bytes) else v


def read_elem(node):
    """One element of a zarr store or an HDF5 file -> its in-memory value (None + a warning for an encoding that is
    not read here).  Arrays are anything with `.read()` (`_zarr3.Array`, `_hdf5.Dataset`), groups anything else."""
    enc = _as_str(node.attrs.get("encoding-type"))
    if enc == "null":  # anndata >= 0.11 stores None this way
        return None
    if hasattr(node, "read"):
        if enc in {"numeric-scalar", "string"} or node.ndim == 0:
            v = _as_str(node.read()[()])
            return str(v) if node.is_string else v
        if enc in {None, "array", "string-array", "rec-array"}:
            v = node.read()
            v = _as_str(v) if node.is_string else v
            cats = node.attrs.get("categories")
            if isinstance(cats, str) and hasattr(node, "store"):
                # the same encoding in a zarr store: the attribute is the path `__categories/<column>` inside the frame
                cat_node = type(node)(node.store, f"{node.path.rsplit('/', 1)[0]}/{cats}")
                return pd.Categorical.from_codes(v, categories=pd.Index(_as_str(cat_node.read())),
                                                 ordered=bool(cat_node.attrs.get("ordered", False)))
            if cats is not None and hasattr(cats, "addr"):
                # anndata 0.7.x (dataframe encoding 0.1.0): integer codes whose `categories` attribute is an HDF5 object
                # reference to `<frame>/__categories/<column>`
                cat_node = node.file.deref(cats)
                return pd.Categorical.from_codes(v, categories=pd.Index(_as_str(cat_node.read())),
                                                 ordered=bool(cat_node.attrs.get("ordered", False)))
            return v
        warnings.warn(f"skipping {node.path!r}: array encoding {enc!r} is not read here", UserWarning, stacklevel=2)
        return None
    if enc in {"csr_matrix", "csc_matrix"}:
        return _read_sparse(node)
    if enc == "dataframe":
        return _read_dataframe(node)
    if enc == "categorical":
        cats = node["categories"].read()
        return pd.Categorical.from_codes(node["codes"].read(), categories=pd.Index(cats),
                                         ordered=bool(node.attrs.get("ordered", False)))
    if enc in {"nullable-integer", "nullable-boolean"}:
        values, mask = node["values"].read(), node["mask"].read().astype(bool)
        if enc == "nullable-boolean":
            return pd.arrays.BooleanArray(values.astype(bool), mask)
        return pd.arrays.IntegerArray(values, mask)
    if enc in {None, "dict", "anndata", "raw"}:
        out = {}
        for k in node.keys():
            v = read_elem(node[k])
            if v is not None:
                out[k] = v
        return out
    warnings.warn(f"skipping {node.path!r}: group encoding {enc!r} is not read here", UserWarning, stacklevel=2)
    return None


def _legacy_frame(v):
    """files written by anndata < 0.7 keep obs / var as one compound dataset (fixed-length strings, an `index` field)"""
    if isinstance(v, np.ndarray) and v.dtype.names:
        cols = {n: _as_str(v[n]) for n in v.dtype.names}
        key = next((k for k in ("index", "obs_names", "var_names", "_index") if k in cols), None)
        index = pd.Index(cols.pop(key)) if key else pd.RangeIndex(v.shape[0]).astype(str)
        return pd.DataFrame(cols, index=index)
    return v


def _read_anndata(root, where, backed) -> AnnData:
    if _as_str(root.attrs.get("encoding-type")) not in {"anndata", None}:
        raise ValueError(f"{where}: not an AnnData store (encoding-type {root.attrs.get('encoding-type')!r})")
    x = None
    if "X" in root:
        xn = root["X"]
        if backed and not hasattr(xn, "read"):
            x = BackedCsr(xn)
        elif backed:
            raise ValueError("backed='r' streams rows of a csr_matrix; this store's X is a dense array")
        else:
            x = read_elem(xn)
    obs = _legacy_frame(read_elem(root["obs"])) if "obs" in root else None
    var = _legacy_frame(read_elem(root["var"])) if "var" in root else None
    kw = {k: (read_elem(root[k]) or {}) if k in root else {} for k in ("obsm", "varm", "obsp", "uns", "layers")}
    adata = AnnData(x, obs, var, **kw)
    if "varp" in root:
        adata.varp = read_elem(root["varp"]) or {}
    if "raw" in root:  # `adata.raw`: the matrix (and var) before gene filtering, same cells (anndata's "raw" encoding)
        rg = root["raw"]
        if not hasattr(rg, "read") and "X" in rg:  # (a store without raw may hold a null / scalar placeholder there)
            rx = rg["X"]
            rx = BackedCsr(rx) if backed and not hasattr(rx, "read") else read_elem(rx)
            rvar = _legacy_frame(read_elem(rg["var"])) if "var" in rg else None
            adata.raw = AnnData(rx, adata.obs, rvar, varm=(read_elem(rg["varm"]) or {}) if "varm" in rg else {})
    return adata


def read_zarr(store, *, backed: str | None = None) -> AnnData:
    """Read an AnnData `.zarr` directory or `.zarr.zip` (zarr format 3, or format 2 as anndata < 0.11 wrote it).

    backed
        None: everything in memory (like `anndata.read_zarr`).  'r': a CSR `X` stays on disk as a `BackedCsr`
        (its `indptr` is loaded, 8 bytes per cell); `pp.pca` then streams it through the device by row chunks.
    """
    if backed not in {None, "r"}:
        raise ValueError("backed must be None or 'r' (stores are never modified in place)")
    st = z3.open_store(store)
    adata = _read_anndata(z3.open_root(st), store, backed)
    if not backed:
        st.close()  # (a backed matrix keeps reading from the store)
    return adata


def read_h5ad(filename, backed: str | None = None) -> AnnData:
    """Read an `.h5ad` file (`anndata.read_h5ad`, re-exported by the reference at `src/scanpy/readwrite.py:15-29` and
    reached from `sc.read(..., backed=...)`, `:832-835`).  The HDF5 container is read by `scanpy_amd/_hdf5.py` (no h5py
    in this image); the element encodings are those of `read_zarr`.

    backed
        None: everything in memory.  'r': a CSR `X` stays on disk as a `BackedCsr` and `pp.pca` streams it through the
        device by row chunks (chunks are inflated on a thread pool straight into recycled buffers).
    """
    if backed not in {None, "r"}:
        raise ValueError("backed must be None or 'r' (files are never modified in place)")
    from . import _hdf5

    f = _hdf5.File(filename)
    adata = _read_anndata(f.root, filename, backed)
    if not backed:
        f.close()  # (a backed matrix keeps reading from the file)
    return adata


def read_10x_h5(filename, *, genome: str | None = None, gex_only: bool = True, backup_url: str | None = None) -> AnnData:
    """Read a 10x-Genomics-formatted HDF5 file (drop-in for `scanpy.read_10x_h5`, src/scanpy/readwrite.py:159-351):
    Cell Ranger v3+ `matrix/` files and legacy per-genome files; cells x genes CSR float32 (10x stores the transpose
    as CSC, which is the same three arrays), `var['gene_ids', 'feature_types', 'genome', ...]`, barcodes as obs names.
    """
    from . import _hdf5

    path = Path(filename)
    if not path.is_file():
        raise FileNotFoundError(f"{path} does not exist" + (" (there is no network here to fetch backup_url)"
                                                            if backup_url else ""))

    def collect(dsets: dict, group) -> None:  # `_collect_datasets`, `:245-250`
        for k in group.keys():
            v = group[k]
            if hasattr(v, "read"):
                dsets[k] = v.read()
            else:
                collect(dsets, v)

    def matrix_of(dsets):  # `:259-268`: int32 counts are converted to float32 in place
        n_cols, n_rows = (int(v) for v in dsets["shape"])
        data = dsets["data"]
        if data.dtype == np.dtype("int32"):
            data = data.astype(np.float32)
        return sparse.csr_matrix((data, dsets["indices"], dsets["indptr"]), shape=(n_rows, n_cols))

    def text(a):
        return np.asarray(_as_str(a)).astype(str)

    with _hdf5.File(path) as f:
        try:
            if "matrix" in f:  # `_read_v3_10x_h5`, `:253-302`
                dsets: dict = {}
                collect(dsets, f["matrix"])
                obs = pd.DataFrame(index=pd.Index(text(dsets["barcodes"])))
                var_cols = {}
                if "gene_id" not in dsets:
                    var_cols["gene_ids"] = text(dsets["id"])
                else:  # a probe-barcode matrix
                    var_cols["gene_ids"] = text(dsets["gene_id"])
                    var_cols["probe_ids"] = text(dsets["id"])
                var_cols["feature_types"] = text(dsets["feature_type"])
                if "filtered_barcodes" in f["matrix"]:
                    obs["filtered_barcodes"] = dsets["filtered_barcodes"].astype(bool)
                if "features" not in f["matrix"]:
                    raise ValueError("10x h5 has no features group")
                feats = f["matrix"]["features"]
                for name in feats.keys():
                    item = feats[name]
                    if hasattr(item, "read") and name not in ["name", "feature_type", "id", "gene_id", "_all_tag_keys"]:
                        var_cols[name] = dsets[name].astype(bool) if item.dtype.kind == "b" else text(dsets[name])
                var = pd.DataFrame(var_cols, index=pd.Index(text(dsets["name"])))
                adata = AnnData(matrix_of(dsets), obs, var)
                if not var.index.is_unique and not (genome or gex_only):
                    warnings.warn("Variable names are not unique. To make them unique, call `.var_names_make_unique`.",
                                  UserWarning, stacklevel=2)
                if genome:
                    if genome not in set(adata.var["genome"]):
                        raise ValueError(f"Could not find data corresponding to genome {genome!r} in {path}. "
                                         f"Available genomes are: {list(adata.var['genome'].unique())}.")
                    adata = adata[:, (adata.var["genome"] == genome).to_numpy()]
                if gex_only:
                    adata = adata[:, (adata.var["feature_types"] == "Gene Expression").to_numpy()]
                if adata.is_view:
                    adata = adata.copy()
                return adata
            children = f.keys()  # `_read_legacy_10x_h5`, `:305-351`
            if not genome:
                if len(children) > 1:
                    raise ValueError(f"{path} contains more than one genome. For legacy 10x h5 files you must specify "
                                     f"the genome if more than one is present. Available genomes are: {children}")
                genome = children[0]
            elif genome not in children:
                raise ValueError(f"Could not find genome {genome!r} in {path}. Available genomes are: {children}")
            dsets = {}
            collect(dsets, f[genome])
            var = pd.DataFrame({"gene_ids": text(dsets["genes"])}, index=pd.Index(text(dsets["gene_names"])))
            return AnnData(matrix_of(dsets), pd.DataFrame(index=pd.Index(text(dsets["barcodes"]))), var)
        except KeyError as e:
            raise Exception("File is missing one or more required datasets.") from e  # noqa: TRY002 (`:241-242`)


def make_index_unique(index: pd.Index, join: str = "-") -> pd.Index:
    """`anndata.utils.make_index_unique`: later duplicates get '-1', '-2', ... appended (skipping names already taken)"""
    if index.is_unique:
        return index
    values = index.to_numpy().astype(object).copy()
    taken = set(values.tolist())
    counters: dict = {}
    dup = index.duplicated(keep="first")
    for i in np.flatnonzero(dup):
        v = values[i]
        k = counters.get(v, 0)
        while True:
            k += 1
            cand = f"{v}{join}{k}"
            if cand not in taken:
                break
        counters[v] = k
        taken.add(cand)
        values[i] = cand
    return pd.Index(values)


def read_10x_mtx(path, *, var_names: str = "gene_symbols", make_unique: bool = True, cache: bool = False,
                 cache_compression=None, gex_only: bool = True, prefix: str | None = None, compressed: bool = True,
                 sparse_format: str = "csr") -> AnnData:
    """Read a 10x-Genomics-formatted mtx directory (drop-in for `scanpy.read_10x_mtx`, src/scanpy/readwrite.py:512-654):
    Cell Ranger v2 (`genes.tsv`, plain files) and v3+ (`features.tsv.gz`, gzipped unless `compressed=False`) layouts,
    `prefix`, `var_names`, `make_unique`, `gex_only`; cells x genes float32 in `sparse_format`.  `cache` is accepted
    and ignored (there is no h5ad cache directory on this path)."""
    from scipy.io import mmread

    path = Path(path)
    prefix = "" if prefix is None else prefix
    if var_names not in {"gene_symbols", "gene_ids"}:
        raise ValueError("`var_names` needs to be 'gene_symbols' or 'gene_ids'")
    if sparse_format not in {"csr", "csc", "coo"}:
        raise ValueError("`sparse_format` needs to be 'csr', 'csc' or 'coo'")
    is_legacy = (path / f"{prefix}genes.tsv").is_file()
    suffix = "" if is_legacy else (".gz" if compressed else "")
    x = mmread(str(path / f"{prefix}matrix.mtx{suffix}"))  # genes x cells, COO
    if x.dtype != np.float32:
        x = x.astype(np.float32)
    x = x.T  # cells x genes
    x = x.tocsr() if sparse_format == "csr" else x.tocsc() if sparse_format == "csc" else x.tocoo()
    genes = pd.read_csv(path / f"{prefix}{'genes' if is_legacy else 'features'}.tsv{suffix}", header=None, sep="\t")
    if var_names == "gene_symbols":
        idx = pd.Index(genes[1].array)
        if make_unique:
            idx = make_index_unique(idx)
        var = pd.DataFrame({"gene_ids": genes[0].to_numpy()}, index=idx.astype("str"))
    else:
        var = pd.DataFrame({"gene_symbols": genes[1].to_numpy()}, index=pd.Index(genes[0].array.astype("str")))
    if not is_legacy:
        var["feature_types"] = genes[2].to_numpy()
    barcodes = pd.read_csv(path / f"{prefix}barcodes.tsv{suffix}", header=None)
    adata = AnnData(x, pd.DataFrame(index=pd.Index(barcodes[0].array.astype("str"))), var)
    if is_legacy or not gex_only:
        return adata
    return adata[:, (adata.var["feature_types"] == "Gene Expression").to_numpy()].copy()


# ---------------------------------------------------------------------------------------------------------------------
# writing

# elements of the CSR arrays per inner chunk / per shard object: 4 Mi values (16 MB of float32) decode in ~10 ms each
# on one core, and 64 of them per object keep a 10M x 4k matrix (2e9 values) at ~8 objects per array
CHUNK_ELEMS = 1 << 22
CHUNKS_PER_SHARD = 64


def _chunking(shape: tuple[int, ...], chunks) -> tuple[tuple[int, ...], tuple[int, ...]]:
    if len(shape) == 0:
        return (), ()
    if chunks is not None and not np.isscalar(chunks) and len(tuple(chunks)) == len(shape):
        c = tuple(int(v) for v in chunks)
    else:  # (also: a 2-d `chunks=` request does not apply to the 1-d components of a sparse matrix)
        row = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        c = (max(1, min(shape[0], CHUNK_ELEMS // max(1, row))),) + tuple(shape[1:])
    c = tuple(max(1, min(ci, max(1, s))) for ci, s in zip(c, shape))
    n0 = -(-shape[0] // c[0]) if shape[0] else 1
    shard = (c[0] * max(1, min(CHUNKS_PER_SHARD, n0)),) + c[1:]
    return c, shard


class _ZarrSink:
    """where `write_elem` puts groups and arrays: a zarr-v3 directory store"""

    def __init__(self, store, level: int):
        self.st, self.level = store, level

    def group(self, path: str, attrs: dict) -> None:
        z3.write_group(self.st, path, attrs)

    def array(self, path: str, value, enc: str | None, version: str, chunks=None) -> None:
        arr = value if isinstance(value, np.ndarray) else np.asarray(value)
        c, shard = _chunking(arr.shape, chunks)
        z3.write_array(self.st, path, arr, chunk_shape=c, shard_shape=shard, level=self.level,
                       attributes={"encoding-type": enc, "encoding-version": version} if enc else {})

    def close(self) -> None:
        self.st.close()


class _H5Sink:
    """... or an HDF5 file: the nodes are collected as a tree and written in post-order by `_hdf5_write.write_tree`
    (arrays are referenced, not copied, until then)"""

    SMALL = 1 << 16  # datasets below this many bytes stay contiguous, as anndata leaves scalars and short vectors
    streams = True   # `array` accepts columns that arrive in pieces (`_BackedColumn`)

    def __init__(self, path, compression: str | None, level: int):
        from . import _hdf5_write as hw

        self.hw, self.path, self.compression, self.level = hw, path, compression, level
        self.root = None

    @staticmethod
    def _attrs(attrs: dict) -> dict:
        out = {}
        for k, v in attrs.items():
            if k == "shape":
                v = np.asarray(v, dtype=np.int64)
            elif k == "column-order":
                v = np.asarray(list(v), dtype=object)
            out[k] = v
        return out

    def group(self, path: str, attrs: dict) -> None:
        node = self.hw.Node(self._attrs(attrs), is_group=True)
        if path == "":
            self.root = node
        else:
            self.root.add(path, node)

    def array(self, path: str, value, enc: str | None, version: str, chunks=None) -> None:
        arr = value if isinstance(value, (np.ndarray, str)) or hasattr(value, "pieces") else np.asarray(value)
        nbytes = arr.nbytes if not isinstance(arr, str) and np.dtype(arr.dtype).kind not in "OU" else 0
        comp = self.compression if nbytes >= self.SMALL else None
        attrs = {"encoding-type": enc, "encoding-version": version} if enc else {}
        self.root.add(path, self.hw.Node(attrs, arr, compression=comp))

    def close(self) -> None:
        self.hw.write_tree(self.path, self.root, level=self.level)


def _write_dataframe(sink, path: str, df: pd.DataFrame) -> None:
    index_key = df.index.name if df.index.name not in (None, "") else "_index"
    if index_key in df.columns:
        raise ValueError(f"the index name {index_key!r} is also a column")
    sink.group(path, {"_index": index_key, "column-order": [str(c) for c in df.columns],
                      "encoding-type": "dataframe", "encoding-version": "0.2.0"})
    write_elem(sink, f"{path}/{index_key}", np.asarray(df.index.astype(str), dtype=object))
    for name in df.columns:
        write_elem(sink, f"{path}/{name}", df[name].array if isinstance(df[name].dtype, pd.CategoricalDtype)
                   or pd.api.types.is_extension_array_dtype(df[name].dtype) else df[name].to_numpy())


class _BackedColumn:
    """`data` or `indices` of a `BackedCsr`, handed to a streaming writer piece by piece (row blocks of the source)"""

    def __init__(self, x, which: str, rows_per_piece: int = 1 << 20):
        self.arr = x._data if which == "data" else x._indices
        self.indptr, self.step = x.indptr, rows_per_piece
        self.shape, self.dtype = (int(x.indptr[-1]),), self.arr.dtype
        self.nbytes = self.shape[0] * self.arr.dtype.itemsize

    def pieces(self):
        n = self.indptr.shape[0] - 1
        for i0 in range(0, n, self.step):
            p0, p1 = int(self.indptr[i0]), int(self.indptr[min(i0 + self.step, n)])
            if p1 > p0:
                yield self.arr.read(p0, p1)


def write_elem(sink, path: str, value, *, chunks=None) -> None:
    """Write one in-memory value with the anndata encoding of its type (the same for both containers)."""
    if is_backed(value) and getattr(sink, "streams", False) and value._cols is None and not value._ops:
        # an on-disk matrix goes from store to file block by block: it is never whole in memory
        sink.group(path, {"shape": [int(s) for s in value.shape], "encoding-type": "csr_matrix",
                          "encoding-version": "0.1.0"})
        sink.array(f"{path}/data", _BackedColumn(value, "data"), None, "")
        sink.array(f"{path}/indices", _BackedColumn(value, "indices"), None, "")
        sink.array(f"{path}/indptr", value.indptr, None, "")
        return
    if is_backed(value):
        value = value.to_memory()
    if sparse.issparse(value):
        fmt = value.format
        if fmt not in {"csr", "csc"}:
            value, fmt = value.tocsr(), "csr"
        sink.group(path, {"shape": [int(s) for s in value.shape], "encoding-type": f"{fmt}_matrix",
                          "encoding-version": "0.1.0"})
        sink.array(f"{path}/data", value.data, None, "", chunks)
        sink.array(f"{path}/indices", value.indices, None, "", chunks)
        sink.array(f"{path}/indptr", value.indptr, None, "", chunks)
    elif isinstance(value, pd.DataFrame):
        _write_dataframe(sink, path, value)
    elif isinstance(value, (pd.Categorical, pd.Series)) and isinstance(value.dtype, pd.CategoricalDtype):
        cat = value if isinstance(value, pd.Categorical) else value.array
        sink.group(path, {"ordered": bool(cat.ordered), "encoding-type": "categorical", "encoding-version": "0.2.0"})
        write_elem(sink, f"{path}/categories", np.asarray(cat.categories))
        sink.array(f"{path}/codes", np.asarray(cat.codes), "array", "0.2.0")
    elif isinstance(value, (pd.arrays.IntegerArray, pd.arrays.BooleanArray)):
        kind = "nullable-boolean" if isinstance(value, pd.arrays.BooleanArray) else "nullable-integer"
        sink.group(path, {"encoding-type": kind, "encoding-version": "0.1.0"})
        fill = False if kind == "nullable-boolean" else 0
        sink.array(f"{path}/values", value.to_numpy(dtype=value.dtype.numpy_dtype, na_value=fill), "array", "0.2.0")
        sink.array(f"{path}/mask", np.asarray(value.isna()), "array", "0.2.0")
    elif isinstance(value, dict):
        sink.group(path, {"encoding-type": "dict", "encoding-version": "0.1.0"})
        for k, v in value.items():
            if v is None:
                continue
            write_elem(sink, f"{path}/{k}", v)
    elif isinstance(value, str):
        sink.array(path, np.asarray(value, dtype=object), "string", "0.2.0")
    elif isinstance(value, (bool, int, float, np.generic)) and not isinstance(value, np.str_):
        sink.array(path, np.asarray(value), "numeric-scalar", "0.2.0")
    else:
        if isinstance(value, pd.Series):
            value = value.to_numpy()
        arr = np.asarray(value)
        if arr.dtype.names:
            sink.array(path, arr, "rec-array", "0.2.0")
        elif arr.dtype.kind in "OUS":
            sink.array(path, arr.astype(object), "string" if arr.ndim == 0 else "string-array", "0.2.0",
                       chunks if arr.ndim else None)
        elif arr.dtype.kind in "biuf":
            sink.array(path, arr, "numeric-scalar" if arr.ndim == 0 else "array", "0.2.0",
                       chunks if arr.ndim == 2 else None)
        else:
            raise TypeError(f"cannot write {path!r}: values of dtype {arr.dtype} have no encoding here")


def _write_anndata(sink, adata, chunks=None) -> None:
    sink.group("", {"encoding-type": "anndata", "encoding-version": "0.1.0"})
    if adata.X is not None:
        write_elem(sink, "X", adata.X, chunks=chunks)
    write_elem(sink, "obs", adata.obs)
    write_elem(sink, "var", adata.var)
    for name in ("obsm", "varm", "obsp", "varp", "layers", "uns"):
        write_elem(sink, name, dict(getattr(adata, name, None) or {}))
    raw = getattr(adata, "raw", None)
    if raw is not None and getattr(raw, "X", None) is not None:
        sink.group("raw", {"encoding-type": "raw", "encoding-version": "0.1.0"})
        write_elem(sink, "raw/X", raw.X)
        write_elem(sink, "raw/var", raw.var)
        write_elem(sink, "raw/varm", dict(getattr(raw, "varm", None) or {}))
    sink.close()


def write_zarr(store, adata, *, chunks=None, level: int = 0) -> None:
    """Write `adata` as an AnnData `.zarr` directory (zarr format 3, zstd, `sharding_indexed`), readable by
    `anndata.read_zarr`.  `chunks` = chunk shape of a dense `X` (anndata's `write_zarr(chunks=...)`); CSR components
    are cut into 4 Mi-element chunks, 64 per shard object, so that row ranges decode in parallel."""
    path = Path(store)
    if path.suffix == ".zip":
        raise ValueError("write a directory store (zip it afterwards if needed)")
    # written next to the target and moved into place on success: `adata` may be backed by the very store it is
    # written to (`read(p, backed='r')` -> `write(p, ...)`), which must stay readable until the last block is copied
    tmp = path.with_name(f".{path.name}.tmp{os.getpid()}")
    if tmp.exists():
        shutil.rmtree(tmp)
    old = None
    try:
        _write_anndata(_ZarrSink(z3.open_store(tmp, "w"), level), adata, chunks)
        if path.exists():
            old = path.with_name(f".{path.name}.old{os.getpid()}")
            os.replace(path, old)
        os.replace(tmp, path)
        if old is not None:
            shutil.rmtree(old, ignore_errors=True) if old.is_dir() else old.unlink()
    except BaseException:
        # interrupted between the two renames: put the original store back before giving up
        if old is not None and old.exists() and not path.exists():
            os.replace(old, path)
        shutil.rmtree(tmp, ignore_errors=True)
        raise


def write_h5ad(filename, adata, *, compression: str | None = None, compression_opts: int | None = None) -> None:
    """Write `adata` as an `.h5ad` file (`AnnData.write_h5ad`, what `sc.write` calls: src/scanpy/readwrite.py:727-740).
    The classic HDF5 container h5py writes by default, produced by `scanpy_amd/_hdf5_write.py`; files were read back
    with the HDF5 library (h5py) in the tests.  compression: None or 'gzip' (+ shuffle), level `compression_opts`
    (default 4); 'lzf' is not written here."""
    if compression not in {None, "gzip"}:
        raise NotImplementedError(f"compression={compression!r}: None and 'gzip' are written here")
    # temporary file in the same directory + atomic rename: the target may be the file `adata.X` is backed by
    filename = Path(filename)
    tmp = filename.with_name(f".{filename.name}.tmp{os.getpid()}")
    try:
        _write_anndata(_H5Sink(tmp, compression, 4 if compression_opts is None else int(compression_opts)), adata)
        os.replace(tmp, filename)
    except BaseException:
        tmp.unlink(missing_ok=True)
        raise


def write(filename, adata, *, ext: str | None = None, compression: str | None = "gzip",
          compression_opts: int | None = None) -> None:
    """Write AnnData objects to file (drop-in for `scanpy.write`, src/scanpy/readwrite.py:657-740): the extension picks
    the container -- 'h5ad' (default when there is none) or 'zarr'; 'csv' is not offered on this path."""
    filename = Path(filename)
    if ext is None:
        ext = filename.suffix.lstrip(".") or "h5ad"
        if not filename.suffix:
            filename = filename.with_suffix(".h5ad")
    if ext == "zarr":
        write_zarr(filename, adata)
    elif ext in {"h5ad", "h5"}:
        write_h5ad(filename, adata, compression=compression, compression_opts=compression_opts)
    else:
        raise ValueError(f"This has to be a {'h5ad'!r} or {'zarr'!r} file ({ext!r} is not written on this path).")


def read(filename, backed: str | None = None, *, ext: str | None = None, **kwargs) -> AnnData:
    """Read a file by its extension (the `.h5ad` / `.h5` / `.zarr` branches of `scanpy.read`,
    src/scanpy/readwrite.py:71-157, 808-841; text, Excel, mtx and loom inputs are not offered on this path)."""
    filename = Path(filename)
    name = filename.name
    ext = ext or ("zarr" if name.endswith((".zarr", ".zarr.zip")) else filename.suffix.lstrip("."))
    if kwargs:
        raise TypeError(f"read() got unexpected arguments {sorted(kwargs)} (the text-format options do not apply here)")
    if ext in {"h5ad", "h5"}:
        return read_h5ad(filename, backed=backed)
    if ext == "zarr":
        return read_zarr(filename, backed=backed)
    raise ValueError(f"{filename}: only 'h5ad', 'h5' and 'zarr' files are read on this path (got {ext!r})")
