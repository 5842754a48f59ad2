"""Writes the HDF5 fixtures of tests/test_hdf5_cpu.py WITH THE HDF5 LIBRARY ITSELF (h5py), plus the arrays written.

Run with an interpreter that has h5py -- in the build container that is the image's conda Python, not the one the
package runs on:

    /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py

Outputs (committed, ~100 KB):
    tests/golden/h5/adata_layout.h5ad    an AnnData-layout file (anndata on-disk format 0.1.0 / 0.2.0 element encodings:
                                         csr X in gzip+shuffle chunks, dataframes, categoricals, string arrays, scalars)
    tests/golden/h5/variants.h5          container variants: libver='latest' (superblock 3, version-2 object headers,
                                         compact link messages, layout-4 chunk indexes), big-endian, compound, enum,
                                         fletcher32, chunked 2-d with edge chunks, missing chunks, a user block
    tests/golden/h5/tracked.h5           `track_order=True` groups under the default libver (dense links)
    tests/golden/h5/tenx_v3_like.h5      the 10x Genomics v3 `matrix/` layout (features x barcodes CSC)
    tests/golden/h5/expected.npz         the arrays that went into the files above
"""
from pathlib import Path

import h5py
import numpy as np

OUT = Path(__file__).resolve().parent / "h5"
OUT.mkdir(exist_ok=True)
rng = np.random.default_rng(0)
expected = {}


def vstr(values):
    return np.array(values, dtype=h5py.string_dtype("utf-8"))


def enc(node, kind, version):
    node.attrs["encoding-type"] = kind
    node.attrs["encoding-version"] = version


def write_csr(parent, name, data, indices, indptr, shape, fmt="csr_matrix", **kw):
    g = parent.create_group(name)
    enc(g, fmt, "0.1.0")
    g.attrs["shape"] = np.array(shape, dtype=np.int64)
    g.create_dataset("data", data=data, **kw)
    g.create_dataset("indices", data=indices, **kw)
    g.create_dataset("indptr", data=indptr, **kw)
    return g


# --------------------------------------------------------------------------------------------------------------------
# 1. AnnData layout
n, g = 500, 60
dense = (rng.random((n, g)) < 0.15) * rng.gamma(2.0, 1.0, (n, g))
dense[7] = 0
dense = dense.astype(np.float32)
indptr = np.zeros(n + 1, dtype=np.int32)
cols, vals = [], []
for i in range(n):
    nz = np.flatnonzero(dense[i])
    cols.append(nz)
    vals.append(dense[i, nz])
    indptr[i + 1] = indptr[i] + nz.size
indices = np.concatenate(cols).astype(np.int32)
data = np.concatenate(vals).astype(np.float32)
expected.update(ad_dense=dense, ad_data=data, ad_indices=indices, ad_indptr=indptr)

with h5py.File(OUT / "adata_layout.h5ad", "w") as f:
    enc(f, "anndata", "0.1.0")
    write_csr(f, "X", data, indices, indptr, (n, g), chunks=(397,), compression="gzip", shuffle=True)
    obs = f.create_group("obs")
    enc(obs, "dataframe", "0.2.0")
    obs.attrs["_index"] = "cell_id"
    obs.attrs["column-order"] = vstr(["n_counts", "louvain", "batch", "is_doublet", "score"])
    names = [f"cell-{i:04d}" for i in range(n)]
    d = obs.create_dataset("cell_id", data=vstr(names), chunks=(128,), compression="gzip")
    enc(d, "string-array", "0.2.0")
    counts = dense.sum(axis=1).astype(np.float32)
    enc(obs.create_dataset("n_counts", data=counts), "array", "0.2.0")
    cat = obs.create_group("louvain")
    enc(cat, "categorical", "0.2.0")
    cat.attrs["ordered"] = False
    codes = rng.integers(-1, 4, n).astype(np.int8)
    enc(cat.create_dataset("codes", data=codes), "array", "0.2.0")
    enc(cat.create_dataset("categories", data=vstr(["0", "1", "2", "10"])), "string-array", "0.2.0")
    batch = [("a", "bé", "ccc")[i % 3] for i in range(n)]
    enc(obs.create_dataset("batch", data=vstr(batch)), "string-array", "0.2.0")
    dbl = rng.random(n) < 0.1
    enc(obs.create_dataset("is_doublet", data=dbl), "array", "0.2.0")
    nul = obs.create_group("score")
    enc(nul, "nullable-integer", "0.1.0")
    score_v = rng.integers(0, 9, n).astype(np.int64)
    score_m = rng.random(n) < 0.2
    nul.create_dataset("values", data=score_v)
    nul.create_dataset("mask", data=score_m)
    var = f.create_group("var")
    enc(var, "dataframe", "0.2.0")
    var.attrs["_index"] = "_index"
    var.attrs["column-order"] = vstr(["highly_variable"])
    enc(var.create_dataset("_index", data=vstr([f"gene{i}" for i in range(g)])), "string-array", "0.2.0")
    hv = rng.random(g) < 0.6
    enc(var.create_dataset("highly_variable", data=hv), "array", "0.2.0")
    obsm = f.create_group("obsm")
    enc(obsm, "dict", "0.1.0")
    xpca = rng.standard_normal((n, 5)).astype(np.float32)
    enc(obsm.create_dataset("X_pca", data=xpca, chunks=(64, 5), compression="gzip"), "array", "0.2.0")
    for name in ("varm", "obsp", "varp"):
        enc(f.create_group(name), "dict", "0.1.0")
    layers = f.create_group("layers")
    enc(layers, "dict", "0.1.0")
    write_csr(layers, "as_csc", data, indices, indptr, (g, n), fmt="csc_matrix")  # the same arrays read as CSC of X^T
    uns = f.create_group("uns")
    enc(uns, "dict", "0.1.0")
    enc(uns.create_dataset("n_neighbors", data=np.int64(15)), "numeric-scalar", "0.2.0")
    enc(uns.create_dataset("resolution", data=np.float64(0.8)), "numeric-scalar", "0.2.0")
    enc(uns.create_dataset("flag", data=np.bool_(True)), "numeric-scalar", "0.2.0")
    enc(uns.create_dataset("method", data="umap", dtype=h5py.string_dtype("utf-8")), "string", "0.2.0")
    nb = uns.create_group("neighbors")
    enc(nb, "dict", "0.1.0")
    pr = nb.create_group("params")
    enc(pr, "dict", "0.1.0")
    enc(pr.create_dataset("metric", data="euclidean", dtype=h5py.string_dtype("utf-8")), "string", "0.2.0")
    enc(uns.create_dataset("colors", data=vstr(["#1f77b4", "#ff7f0e"])), "string-array", "0.2.0")
    rec = np.zeros(3, dtype=[("a", "<f4"), ("b", "<i8")])
    rec["a"], rec["b"] = [1.5, 2.5, 3.5], [1, 2, 3]
    enc(uns.create_dataset("rec", data=rec), "rec-array", "0.2.0")
expected.update(ad_names=np.array(names), ad_counts=counts, ad_codes=codes, ad_batch=np.array(batch), ad_dbl=dbl,
                ad_score_v=score_v, ad_score_m=score_m, ad_hv=hv, ad_xpca=xpca, ad_rec_a=rec["a"], ad_rec_b=rec["b"])

# the dataframe encoding of anndata 0.7.x (version 0.1.0): categorical columns are integer codes whose `categories`
# attribute is an object reference into `<frame>/__categories`
with h5py.File(OUT / "adata_07_layout.h5ad", "w") as f:
    enc(f, "anndata", "0.1.0")
    write_csr(f, "X", data, indices, indptr, (n, g))
    o = f.create_group("obs")
    enc(o, "dataframe", "0.1.0")
    o.attrs["_index"] = "_index"
    o.attrs["column-order"] = vstr(["louvain", "n_counts"])
    o.create_dataset("_index", data=vstr(names))
    cg = o.create_group("__categories")
    cd = cg.create_dataset("louvain", data=vstr(["0", "1", "2", "10"]))
    cd.attrs["ordered"] = False
    codes07 = np.where(codes < 0, 0, codes).astype(np.int8)
    lv = o.create_dataset("louvain", data=codes07)
    lv.attrs["categories"] = cd.ref
    o.create_dataset("n_counts", data=counts)
    v07 = f.create_group("var")
    enc(v07, "dataframe", "0.1.0")
    v07.attrs["_index"] = "_index"
    v07.attrs["column-order"] = np.array([], dtype=np.float64)  # what h5py stores for an empty list
    v07.create_dataset("_index", data=vstr([f"gene{i}" for i in range(g)]))
expected.update(ad07_codes=codes07)

# the layout anndata < 0.7 wrote: obs / var as compound datasets, X marked with h5sparse_* attributes
with h5py.File(OUT / "legacy_layout.h5ad", "w") as f:
    xg = f.create_group("X")
    xg.attrs["h5sparse_format"] = np.bytes_("csr")
    xg.attrs["h5sparse_shape"] = np.array([n, g], dtype=np.int64)
    xg.create_dataset("data", data=data, chunks=(512,), compression="gzip")
    xg.create_dataset("indices", data=indices, chunks=(512,), compression="gzip")
    xg.create_dataset("indptr", data=indptr)
    o = np.zeros(n, dtype=[("index", "S9"), ("n_counts", "<f4")])
    o["index"], o["n_counts"] = [s.encode() for s in names], counts
    f.create_dataset("obs", data=o)
    v = np.zeros(g, dtype=[("index", "S6"), ("highly_variable", "?")])
    v["index"], v["highly_variable"] = [f"gene{i}".encode() for i in range(g)], hv
    f.create_dataset("var", data=v)

# --------------------------------------------------------------------------------------------------------------------
# 2. container variants
big = rng.integers(0, 1000, 10_000).astype(np.int64)
two_d = rng.standard_normal((103, 7)).astype(np.float64)
with h5py.File(OUT / "variants.h5", "w", libver="latest", userblock_size=512) as f:
    f.attrs["title"] = "libver latest"
    f.attrs["numbers"] = np.arange(5, dtype=np.int16)
    g1 = f.create_group("grp")
    g1.attrs["note"] = vstr(["a", "bb"])
    g1.create_dataset("single_chunk", data=big[:100], chunks=(100,), compression="gzip", shuffle=True)
    g1.create_dataset("implicit", data=big[:256].astype(np.int32), chunks=(64,))
    g1.create_dataset("fixed_array", data=big, chunks=(1024,), compression="gzip", shuffle=True)
    g1.create_dataset("fixed_array_plain", data=big[:5000].astype(np.uint16), chunks=(512,), fletcher32=True)
    g1.create_dataset("paged", data=np.arange(40_000, dtype=np.int32), chunks=(16,), compression="gzip")
    g1.create_dataset("two_d", data=two_d, chunks=(10, 4), compression="gzip", shuffle=True)
    g2 = f.create_group("grp2")  # (at most 8 links per group: more would move the links into a fractal heap)
    g2.create_dataset("big_endian", data=big[:50].astype(">i4"))
    g2.create_dataset("compact", data=np.arange(6, dtype=np.uint8).reshape(2, 3))
    g2.create_dataset("scalar_f", data=np.float32(2.5))
    g2.create_dataset("empty", shape=(0,), dtype=np.float32)
    sparse_ds = g2.create_dataset("missing_chunks", shape=(300,), dtype=np.int32, chunks=(100,), compression="gzip")
    sparse_ds[100:200] = np.arange(100, dtype=np.int32)
    g2.create_dataset("resizable", data=big[:300], chunks=(100,), maxshape=(None,))  # extensible array: not read
    for i in range(12):  # > 8 links in a new-style group -> dense link storage (one fractal-heap direct block)
        f.require_group("dense").create_dataset(f"d{i}", data=np.int8(i))
    wide = f.create_group("dense_wide")  # enough links for an indirect block over several direct blocks
    target = wide.create_dataset("target", data=np.int16(7))
    for i in range(700):
        wide[f"hard_link_number_{i:04d}"] = target
    holes = f.create_group("dense_holes")  # links deleted from a dense group: refused, not guessed at
    for i in range(20):
        holes[f"h{i:02d}"] = target
    for i in (3, 4, 11):
        del holes[f"h{i:02d}"]
with h5py.File(OUT / "tracked.h5", "w", track_order=True) as f:  # default libver + creation-order tracking
    f.attrs["a"] = 1
    tg = f.create_group("g", track_order=True)
    for i in range(15):
        tg.create_dataset(f"k{14 - i:02d}", data=np.int32(i))
with h5py.File(OUT / "variants_v0.h5", "w") as f:  # default libver: superblock 0, symbol-table groups, B-tree v1
    many = f.create_group("many")
    for i in range(40):  # more than one SNOD leaf
        many.create_dataset(f"item{i:02d}", data=np.int32(i))
    f.create_dataset("btree", data=big, chunks=(37,), compression="gzip", shuffle=True)  # a multi-level chunk B-tree
    f.create_dataset("two_d", data=two_d, chunks=(10, 4), compression="gzip")
    f.create_dataset("lzf", data=big, chunks=(1000,), compression="lzf", shuffle=True)
    noise = rng.integers(0, 256, 4096).astype(np.uint8)  # incompressible: h5py stores such chunks raw (filter mask)
    f.create_dataset("lzf_noise", data=np.concatenate([noise, np.zeros(4096, np.uint8)]), chunks=(4096,),
                     compression="lzf")
    f.create_dataset("fixed_str", data=np.array([b"ab", b"cde", b""], dtype="S3"))
    u8 = f.create_dataset("utf8_fixed", shape=(2,), dtype=h5py.string_dtype("utf-8", 4))
    u8[0], u8[1] = "é", "zz"
    long_attr = f.create_group("attrs")
    for i in range(20):  # pushes the version-1 object header into continuation blocks
        long_attr.attrs[f"key{i}"] = f"value {i}"
    long_attr.attrs["bools"] = np.array([True, False])
    long_attr.attrs["empty"] = h5py.Empty("f")
expected.update(v_big=big, v_two_d=two_d, v_paged=np.arange(40_000, dtype=np.int32), v_noise=noise)

# --------------------------------------------------------------------------------------------------------------------
# 3. 10x v3 layout (features x barcodes, CSC = cells x genes CSR)
nb_, nf = 40, 25
m = (rng.random((nb_, nf)) < 0.2) * rng.integers(1, 9, (nb_, nf))
tp = np.zeros(nb_ + 1, dtype=np.int64)
ti, tv = [], []
for i in range(nb_):
    nz = np.flatnonzero(m[i])
    ti.append(nz)
    tv.append(m[i, nz])
    tp[i + 1] = tp[i] + nz.size
with h5py.File(OUT / "tenx_v3_like.h5", "w") as f:
    f.attrs["filetype"] = "matrix"
    mg = f.create_group("matrix")
    mg.create_dataset("barcodes", data=np.array([f"BC{i:03d}-1".encode() for i in range(nb_)], dtype="S18"))
    mg.create_dataset("data", data=np.concatenate(tv).astype(np.int32), chunks=(64,), compression="gzip", shuffle=True)
    mg.create_dataset("indices", data=np.concatenate(ti).astype(np.int64), chunks=(64,), compression="gzip")
    mg.create_dataset("indptr", data=tp, chunks=(16,), compression="gzip")
    mg.create_dataset("shape", data=np.array([nf, nb_], dtype=np.int32))
    ft = mg.create_group("features")
    ft.create_dataset("_all_tag_keys", data=np.array([b"genome"], dtype="S6"))
    ft.create_dataset("id", data=np.array([f"ENSG{i:05d}".encode() for i in range(nf)], dtype="S15"))
    ft.create_dataset("name", data=np.array([f"G{i % 20}".encode() for i in range(nf)], dtype="S15"))  # duplicates
    ftype = [b"Gene Expression"] * 20 + [b"Antibody Capture"] * 5
    ft.create_dataset("feature_type", data=np.array(ftype, dtype="S16"))
    ft.create_dataset("genome", data=np.array([b"GRCh38"] * 20 + [b""] * 5, dtype="S6"))
expected.update(tenx_dense=m.astype(np.float32))

np.savez_compressed(OUT / "expected.npz", **expected)
for p in sorted(OUT.iterdir()):
    print(p.name, p.stat().st_size)
