"""The HDF5 reader (scanpy_amd/_hdf5.py) and the readers on top of it (`read_h5ad`, `read_10x_h5`).

Pinned against files written by the HDF5 library: (1) committed fixtures that h5py wrote (tests/golden/make_h5_golden.py:
an AnnData-layout file, container variants old and new, a 10x v3 layout) with the arrays that went in; (2) when the
reference checkout is on this machine, its own Cell Ranger / PyTables files against the Matrix Market exports next to
them -- the comparison the reference makes in tests/test_read_10x.py:38-95."""
from __future__ import annotations

import gzip
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
from scipy import sparse
from scipy.io import mmread

import scanpy_amd as sc
from scanpy_amd import _hdf5 as h5
from scanpy_amd._backed import BackedCsr

H5 = Path(__file__).parent / "golden" / "h5"
REF_10X = Path("/root/reference/tests/_data/10x_data")
REF_VISIUM = Path("/root/reference/tests/_data/visium_data")
needs_reference = pytest.mark.skipif(not REF_10X.is_dir(), reason="the reference checkout is not on this machine")


@pytest.fixture(scope="module")
def expected():
    return np.load(H5 / "expected.npz")


def test_anndata_layout_file(expected):
    a = sc.read_h5ad(H5 / "adata_layout.h5ad")
    assert a.shape == (500, 60) and sparse.isspmatrix_csr(a.X) and a.X.dtype == np.float32
    np.testing.assert_array_equal(a.X.toarray(), expected["ad_dense"])
    np.testing.assert_array_equal(a.X.indptr, expected["ad_indptr"])
    assert list(a.obs.columns) == ["n_counts", "louvain", "batch", "is_doublet", "score"]  # column-order attribute
    assert list(a.obs_names) == list(expected["ad_names"]) and a.obs.index.name == "cell_id"
    np.testing.assert_array_equal(a.obs["n_counts"].to_numpy(), expected["ad_counts"])
    np.testing.assert_array_equal(a.obs["louvain"].cat.codes.to_numpy(), expected["ad_codes"])  # -1 = missing
    assert list(a.obs["louvain"].cat.categories) == ["0", "1", "2", "10"]
    assert list(a.obs["batch"]) == list(expected["ad_batch"])  # variable-length UTF-8 through the global heap
    np.testing.assert_array_equal(a.obs["is_doublet"].to_numpy(), expected["ad_dbl"])  # h5py bool = HDF5 enum
    assert a.obs["is_doublet"].dtype == bool
    score = a.obs["score"]
    np.testing.assert_array_equal(score.isna().to_numpy(), expected["ad_score_m"])
    np.testing.assert_array_equal(score.to_numpy(dtype="int64", na_value=0)[~expected["ad_score_m"]],
                                  expected["ad_score_v"][~expected["ad_score_m"]])
    np.testing.assert_array_equal(a.var["highly_variable"].to_numpy(), expected["ad_hv"])
    assert a.var_names[3] == "gene3" and a.var.index.name is None
    np.testing.assert_array_equal(a.obsm["X_pca"], expected["ad_xpca"])
    csc = a.layers["as_csc"]
    assert sparse.isspmatrix_csc(csc) and csc.shape == (60, 500)
    np.testing.assert_array_equal(csc.toarray(), expected["ad_dense"].T)
    u = a.uns
    assert u["n_neighbors"] == 15 and u["resolution"] == 0.8 and bool(u["flag"]) is True and u["method"] == "umap"
    assert u["neighbors"]["params"]["metric"] == "euclidean" and list(u["colors"]) == ["#1f77b4", "#ff7f0e"]
    np.testing.assert_array_equal(u["rec"]["a"], expected["ad_rec_a"])
    np.testing.assert_array_equal(u["rec"]["b"], expected["ad_rec_b"])


def test_backed_h5ad_rows(expected):
    a = sc.read_h5ad(H5 / "adata_layout.h5ad", backed="r")
    b = a.X
    assert isinstance(b, BackedCsr) and b.shape == (500, 60) and b.nnz == expected["ad_data"].size
    full = sparse.csr_matrix(expected["ad_dense"])
    for i0, i1 in ((0, 500), (0, 1), (7, 8), (100, 377), (499, 500), (250, 250)):  # chunks of 397 values: ranges cut them
        r = b.rows(i0, i1)
        assert (r.to_scipy() != full[i0:i1]).nnz == 0 and r.data.dtype == np.float32 and r.indptr.dtype == np.int64
    ring = (np.empty(b.nnz, np.int32), np.empty(b.nnz, np.float32))
    r = b.rows(3, 450, out=ring)
    assert np.shares_memory(r.data, ring[1]) and (r.to_scipy() != full[3:450]).nnz == 0
    assert b.absmax() == float(expected["ad_data"].max())
    mask = expected["ad_hv"]
    assert (b[:, mask].rows(10, 90).to_scipy() != full[10:90][:, mask]).nnz == 0
    with pytest.raises(ValueError, match="backed"):
        sc.read_h5ad(H5 / "adata_layout.h5ad", backed="r+")


def test_anndata_07_layout_categoricals_by_object_reference(expected):
    """dataframe encoding 0.1.0 (anndata 0.7.x): codes + a `categories` attribute that is an HDF5 object reference"""
    a = sc.read_h5ad(H5 / "adata_07_layout.h5ad")
    assert list(a.obs.columns) == ["louvain", "n_counts"] and a.var.shape == (60, 0)
    assert list(a.obs["louvain"].cat.categories) == ["0", "1", "2", "10"] and not a.obs["louvain"].cat.ordered
    np.testing.assert_array_equal(a.obs["louvain"].cat.codes.to_numpy(), expected["ad07_codes"])
    np.testing.assert_array_equal(a.obs["n_counts"].to_numpy(), expected["ad_counts"])
    assert (a.X != sparse.csr_matrix(expected["ad_dense"])).nnz == 0
    with h5.File(H5 / "adata_07_layout.h5ad") as f:
        ref = f["obs"]["louvain"].attrs["categories"]
        assert isinstance(ref, h5.Reference) and f.deref(ref).read().tolist() == ["0", "1", "2", "10"]


def test_legacy_layout(expected):
    """anndata < 0.7: compound obs / var datasets, `h5sparse_format` / `h5sparse_shape` on X"""
    a = sc.read_h5ad(H5 / "legacy_layout.h5ad", backed="r")
    assert isinstance(a.X, BackedCsr) and a.shape == (500, 60)
    assert (a.X.to_memory() != sparse.csr_matrix(expected["ad_dense"])).nnz == 0
    assert list(a.obs_names) == list(expected["ad_names"]) and list(a.obs.columns) == ["n_counts"]
    np.testing.assert_array_equal(a.obs["n_counts"].to_numpy(), expected["ad_counts"])
    np.testing.assert_array_equal(a.var["highly_variable"].to_numpy(), expected["ad_hv"])
    assert a.var["highly_variable"].dtype == bool and a.var_names[5] == "gene5"


def test_container_variants_new_style(expected):
    """libver='latest': superblock 3 behind a 512-byte user block, version-2 object headers, compact link messages,
    layout-4 chunk indexes (single chunk, fixed array, paged fixed array), fletcher32, big-endian, compact layout,
    dense link storage"""
    f = h5.File(H5 / "variants.h5")
    assert f.r.base == 512 and f.attrs["title"] == "libver latest"
    np.testing.assert_array_equal(f.attrs["numbers"], np.arange(5, dtype=np.int16))
    g = f["grp"]
    assert list(g.attrs["note"]) == ["a", "bb"]
    big, two_d = expected["v_big"], expected["v_two_d"]
    np.testing.assert_array_equal(g["single_chunk"].read(), big[:100])
    np.testing.assert_array_equal(g["implicit"].read(), big[:256])
    np.testing.assert_array_equal(g["fixed_array"].read(), big)
    np.testing.assert_array_equal(g["fixed_array"].read(1000, 5555), big[1000:5555])
    np.testing.assert_array_equal(g["fixed_array_plain"].read(), big[:5000].astype(np.uint16))
    np.testing.assert_array_equal(g["paged"].read(17, 33_333), expected["v_paged"][17:33_333])
    np.testing.assert_array_equal(g["two_d"].read(), two_d)
    np.testing.assert_array_equal(g["two_d"][9:31], two_d[9:31])
    g2 = f["grp2"]
    be = g2["big_endian"].read()
    assert be.dtype == np.dtype(">i4")
    np.testing.assert_array_equal(be, big[:50])
    np.testing.assert_array_equal(g2["compact"].read(), np.arange(6, dtype=np.uint8).reshape(2, 3))
    assert g2["scalar_f"][()] == np.float32(2.5) and g2["empty"].read().shape == (0,)
    want = np.zeros(300, dtype=np.int32)
    want[100:200] = np.arange(100)
    np.testing.assert_array_equal(g2["missing_chunks"].read(), want)  # unallocated chunks read as the fill value
    np.testing.assert_array_equal(g2["missing_chunks"].read(150, 250), want[150:250])
    with pytest.raises(NotImplementedError, match="extensible array"):
        g2["resizable"]
    # dense link storage (more than 8 links in a new-style group): fractal heap scanned in storage order
    assert f["dense"].keys() == sorted(f"d{i}" for i in range(12))
    assert [int(f["dense"][f"d{i}"][()]) for i in range(12)] == list(range(12))
    wide = f["dense_wide"]  # 701 links: an indirect block over several direct blocks
    assert wide.keys() == [f"hard_link_number_{i:04d}" for i in range(700)] + ["target"]
    assert wide["hard_link_number_0456"][()] == 7
    with pytest.raises(NotImplementedError, match="links were deleted"):
        f["dense_holes"].keys()  # stale link messages in the heap: refused rather than listed wrongly
    with pytest.raises(KeyError):
        g["nope"]
    with pytest.raises(IndexError):
        g["fixed_array"].read(5, 10_001)
    f.close()


def test_container_variants_old_style(expected):
    """default libver: superblock 0, symbol-table groups over several leaves, a multi-level chunk B-tree, version-1
    object headers spilling into continuation blocks"""
    with h5.File(H5 / "variants_v0.h5") as f:
        assert f["many"].keys() == [f"item{i:02d}" for i in range(40)]
        assert [int(f["many"][k][()]) for k in f["many"].keys()] == list(range(40))
        np.testing.assert_array_equal(f["btree"].read(), expected["v_big"])  # 271 chunks of 37
        np.testing.assert_array_equal(f["btree"].read(123, 4567), expected["v_big"][123:4567])
        np.testing.assert_array_equal(f["two_d"].read(5, 99), expected["v_two_d"][5:99])
        np.testing.assert_array_equal(f["lzf"].read(), expected["v_big"])  # h5py's lzf filter + shuffle
        np.testing.assert_array_equal(f["lzf"].read(1500, 7777), expected["v_big"][1500:7777])
        noise = f["lzf_noise"].read()  # an incompressible chunk is stored raw and flagged in the chunk's filter mask
        np.testing.assert_array_equal(noise[:4096], expected["v_noise"])
        assert not noise[4096:].any()
        assert f["fixed_str"].read().tolist() == [b"ab", b"cde", b""]
        assert f["utf8_fixed"].read().tolist() == ["é", "zz"]
        at = f["attrs"].attrs
        assert [at[f"key{i}"] for i in range(20)] == [f"value {i}" for i in range(20)]
        assert at["bools"].tolist() == [True, False] and at["bools"].dtype == bool and at["empty"] is None
        assert "btree" in f and "nope" not in f and f["many/item07"][()] == 7


def test_track_order_groups():
    """`h5py.File(..., track_order=True)` under the default libver: creation-order tracked groups with dense links"""
    with h5.File(H5 / "tracked.h5") as f:
        assert f.attrs["a"] == 1
        assert f["g"].keys() == [f"k{i:02d}" for i in range(15)]
        assert [int(f["g"][f"k{i:02d}"][()]) for i in range(15)] == list(range(14, -1, -1))


def test_not_hdf5(tmp_path):
    (tmp_path / "x.h5").write_bytes(b"definitely not HDF5" * 10)
    with pytest.raises(ValueError, match="not an HDF5 file"):
        h5.File(tmp_path / "x.h5")
    with pytest.raises(FileNotFoundError):
        sc.read_10x_h5(tmp_path / "missing.h5")


def test_read_10x_h5_v3_layout(expected):
    """`gex_only`, `genome`, duplicate names, int32 counts -> float32 (src/scanpy/readwrite.py:204-229, 259-262)"""
    dense = expected["tenx_dense"]
    a = sc.read_10x_h5(H5 / "tenx_v3_like.h5")
    assert a.shape == (40, 20) and sparse.isspmatrix_csr(a.X) and a.X.dtype == np.float32 and not a.is_view
    np.testing.assert_array_equal(a.X.toarray(), dense[:, :20])
    assert list(a.var.columns) == ["gene_ids", "feature_types", "genome"]
    assert a.obs_names[1] == "BC001-1" and a.var_names[2] == "G2" and a.var["gene_ids"].iloc[2] == "ENSG00002"
    with pytest.warns(UserWarning, match="not unique"):
        full = sc.read_10x_h5(H5 / "tenx_v3_like.h5", gex_only=False)
    assert full.shape == (40, 25) and set(full.var["feature_types"]) == {"Gene Expression", "Antibody Capture"}
    np.testing.assert_array_equal(full.X.toarray(), dense)
    g = sc.read_10x_h5(H5 / "tenx_v3_like.h5", genome="GRCh38", gex_only=False)
    assert g.shape == (40, 20)
    with pytest.raises(ValueError, match="Could not find data corresponding to genome 'mm10'"):
        sc.read_10x_h5(H5 / "tenx_v3_like.h5", genome="mm10")


def _read_mtx_dir(d: Path):
    """cells x genes matrix + names of a Cell Ranger Matrix Market export (plain v2 files or gzipped v3 files)"""
    opener = (lambda p: gzip.open(p, "rt")) if (d / "matrix.mtx.gz").is_file() else (lambda p: open(p))
    suffix = ".gz" if (d / "matrix.mtx.gz").is_file() else ""
    m = sparse.csr_matrix(mmread(str(d / f"matrix.mtx{suffix}")).T)
    with opener(d / f"barcodes.tsv{suffix}") as fh:
        barcodes = [line.strip() for line in fh]
    genes_file = d / (f"features.tsv{suffix}" if suffix else "genes.tsv")
    with opener(genes_file) as fh:
        genes = pd.read_csv(fh, sep="\t", header=None)
    return m, barcodes, genes


@needs_reference
@pytest.mark.parametrize(("mtx_rel", "h5_rel"), [
    ("1.2.0/filtered_gene_bc_matrices/hg19_chr21", "1.2.0/filtered_gene_bc_matrices_h5.h5"),
    ("3.0.0/filtered_feature_bc_matrix", "3.0.0/filtered_feature_bc_matrix.h5"),
])
def test_read_10x_h5_equals_the_matrix_market_export(mtx_rel, h5_rel):
    """tests/test_read_10x.py:38-95: the h5 file and the mtx directory of the same Cell Ranger run hold the same data"""
    m, barcodes, genes = _read_mtx_dir(REF_10X / mtx_rel)
    a = sc.read_10x_h5(REF_10X / h5_rel)
    assert a.shape == m.shape and sparse.isspmatrix_csr(a.X) and a.X.dtype == np.float32
    assert np.allclose(a.X.toarray(), m.toarray())
    assert list(a.obs_names) == barcodes
    assert list(a.var["gene_ids"]) == list(genes[0]) and list(a.var_names) == list(genes[1])
    if "3.0.0" in h5_rel:
        assert list(a.var["feature_types"]) == list(genes[2]) and "genome" in a.var.columns


@needs_reference
def test_read_10x_h5_legacy_genomes_and_probe_matrices():
    """tests/test_read_10x.py:98-131 (multiple genomes), :194-225 (probe-barcode matrices)"""
    multi = REF_10X / "1.2.0" / "multiple_genomes.h5"
    with pytest.raises(ValueError, match="contains more than one genome"):
        sc.read_10x_h5(multi)
    with pytest.raises(ValueError, match="Could not find genome 'nope'"):
        sc.read_10x_h5(multi, genome="nope")
    one = sc.read_10x_h5(multi, genome="hg19_chr21")
    same = sc.read_10x_h5(REF_10X / "1.2.0" / "filtered_gene_bc_matrices_h5.h5")
    assert one.shape == same.shape and (one.X != same.X).nnz == 0 and list(one.var_names) == list(same.var_names)
    probe = sc.read_10x_h5(REF_VISIUM / "2.1.0" / "raw_probe_bc_matrix.h5")
    assert probe.shape == (4987, 1000) and probe.X.nnz == 858
    assert {"gene_ids", "probe_ids", "feature_types", "filtered_probes", "gene_name", "genome",
            "probe_region"} <= set(probe.var.columns)
    assert probe.var["filtered_probes"].dtype == bool and probe.obs["filtered_barcodes"].dtype == bool
    assert probe.var_names[0] == "Itgb2l|2ef1e7b" and probe.var["probe_ids"].iloc[0].endswith("|Itgb2l|2ef1e7b")


def test_host_codecs_of_the_c_abi():
    """`scamd_unshuffle` / `scamd_lzf_decompress` (include/scanpy_amd.h, csrc/hostio.cpp): no GPU involved"""
    import ctypes

    from scanpy_amd._lib import load

    lib = load()
    rng = np.random.default_rng(0)
    for es in (1, 2, 3, 4, 8, 18):
        for n in (0, 1, 7, 4096, 10_001):
            elems = rng.integers(0, 256, (n, es), dtype=np.uint8)
            planes = np.ascontiguousarray(elems.T).tobytes()
            for shift in (0, 1):  # aligned and unaligned destinations
                buf = np.zeros(n * es + shift, dtype=np.uint8)
                dst = buf[shift:]
                assert lib.scamd_unshuffle(planes, dst.ctypes.data, n, es) == 0
                np.testing.assert_array_equal(dst.reshape(n, es), elems)
    assert lib.scamd_unshuffle(b"", None, 0, 0) < 0

    def lzf(stream: bytes, cap: int):
        out = ctypes.create_string_buffer(max(cap, 1))
        got = lib.scamd_lzf_decompress(stream, len(stream), out, cap)
        return got, out.raw[:max(got, 0)]

    # literal run "abc"; back reference of length 2 + 2 at distance 3 -> "abca"; run-length: distance 1, length 9
    assert lzf(bytes([2]) + b"abc" + bytes([(2 << 5) | 0, 2]), 64) == (7, b"abcabca")
    assert lzf(bytes([0]) + b"x" + bytes([(7 << 5) | 0, 0, 0]), 64) == (10, b"x" * 10)
    assert lzf(b"", 8) == (0, b"")
    for bad in (bytes([5]) + b"ab",                      # literal run longer than the input
                bytes([(1 << 5) | 0, 0]),                 # reference before the start of the output
                bytes([0]) + b"a" + bytes([(7 << 5)]),    # truncated extended length
                bytes([0]) + b"a" + bytes([(1 << 5)])):   # truncated distance
        assert lzf(bad, 64)[0] < 0
    assert lzf(bytes([3]) + b"abcd", 2)[0] < 0  # destination too small
    assert b"buffers" in lib.scamd_last_error() or b"truncated" in lib.scamd_last_error()


def _write_mtx_dir(d: Path, m, barcodes, ids, names, types=None, *, gz: bool, prefix: str = ""):
    from scipy.io import mmwrite

    d.mkdir(parents=True, exist_ok=True)
    mmwrite(str(d / f"{prefix}matrix.mtx"), sparse.coo_matrix(m.T), field="integer")
    feats = "\n".join("\t".join(filter(None, (i, n, t))) for i, n, t in
                      zip(ids, names, types or [None] * len(ids))) + "\n"
    files = {f"{prefix}barcodes.tsv": "\n".join(barcodes) + "\n",
             f"{prefix}{'features' if types else 'genes'}.tsv": feats}
    for name, text in files.items():
        (d / name).write_text(text)
    if gz:
        for name in [f"{prefix}matrix.mtx", *files]:
            raw = (d / name).read_bytes()
            with gzip.open(d / (name + ".gz"), "wb") as fh:
                fh.write(raw)
            (d / name).unlink()


def test_read_10x_mtx_layouts(tmp_path):
    """src/scanpy/readwrite.py:512-654: v2 (`genes.tsv`) and v3 (`features.tsv.gz`) directories, `prefix`, `var_names`,
    `make_unique`, `gex_only`, `compressed=False`, `sparse_format`"""
    rng = np.random.default_rng(0)
    m = sparse.csr_matrix((rng.random((30, 8)) < 0.3) * rng.integers(1, 9, (30, 8)))
    barcodes = [f"BC{i:02d}-1" for i in range(30)]
    ids = [f"ENSG{i:04d}" for i in range(8)]
    names = ["A", "B", "A", "C", "A", "A-1", "D", "E"]  # duplicates, and a name that collides with a made-unique one
    types = ["Gene Expression"] * 6 + ["Antibody Capture"] * 2
    _write_mtx_dir(tmp_path / "v2", m, barcodes, ids, names, gz=False)
    a = sc.read_10x_mtx(tmp_path / "v2")
    assert a.shape == (30, 8) and sparse.isspmatrix_csr(a.X) and a.X.dtype == np.float32 and (a.X != m).nnz == 0
    assert list(a.var_names) == ["A", "B", "A-2", "C", "A-3", "A-1", "D", "E"]  # `anndata.utils.make_index_unique`
    assert list(a.var["gene_ids"]) == ids and list(a.obs_names) == barcodes and list(a.var.columns) == ["gene_ids"]
    assert list(sc.read_10x_mtx(tmp_path / "v2", make_unique=False).var_names) == names
    b = sc.read_10x_mtx(tmp_path / "v2", var_names="gene_ids", sparse_format="csc")
    assert list(b.var_names) == ids and list(b.var["gene_symbols"]) == names and sparse.isspmatrix_csc(b.X)
    _write_mtx_dir(tmp_path / "v3", m, barcodes, ids, names, types, gz=True, prefix="s1_")
    c = sc.read_10x_mtx(tmp_path / "v3", prefix="s1_")
    assert c.shape == (30, 6) and set(c.var["feature_types"]) == {"Gene Expression"} and (c.X != m[:, :6]).nnz == 0
    full = sc.read_10x_mtx(tmp_path / "v3", prefix="s1_", gex_only=False)
    assert full.shape == (30, 8) and list(full.var.columns) == ["gene_ids", "feature_types"]
    _write_mtx_dir(tmp_path / "star", m, barcodes, ids, names, types, gz=False)  # STARsolo: v3 files, not gzipped
    assert sc.read_10x_mtx(tmp_path / "star", compressed=False, gex_only=False).shape == (30, 8)
    with pytest.raises(ValueError, match="`var_names` needs to be"):
        sc.read_10x_mtx(tmp_path / "v2", var_names="nope")


@needs_reference
@pytest.mark.parametrize(("mtx_rel", "h5_rel"), [
    ("1.2.0/filtered_gene_bc_matrices/hg19_chr21", "1.2.0/filtered_gene_bc_matrices_h5.h5"),
    ("3.0.0/filtered_feature_bc_matrix", "3.0.0/filtered_feature_bc_matrix.h5"),
])
def test_read_10x_mtx_equals_read_10x_h5(mtx_rel, h5_rel):
    """the reference's own assertion (tests/test_read_10x.py:58-95): both readers return the same AnnData"""
    mtx = sc.read_10x_mtx(REF_10X / mtx_rel, var_names="gene_symbols")
    h5f = sc.read_10x_h5(REF_10X / h5_rel)
    if "3.0.0" in h5_rel:
        del h5f.var["genome"]
    assert sparse.isspmatrix_csr(mtx.X) and mtx.shape == h5f.shape
    assert (mtx.obs == h5f.obs).all(axis=None) and (mtx.var == h5f.var).all(axis=None)
    assert list(mtx.var_names) == list(h5f.var_names) and list(mtx.obs_names) == list(h5f.obs_names)
    assert np.allclose(mtx.X.toarray(), h5f.X.toarray())
