"""`read_zarr` / `write_zarr` and the zarr-v3 layer under them (scanpy_amd/_zarr3.py, _backed.py, readwrite.py).

Pinned two ways: (1) against a store written by anndata + zarr-python themselves -- the reference's bundled
`src/scanpy/datasets/10x_pbmc68k_reduced.zarr.zip`, read in place when /root/reference exists (this container) and
compared with the values `tests/golden/make_golden.py` extracted from it; (2) write -> read round trips, metadata
compared key by key with what zarr-python wrote, and range reads compared with in-memory slicing for ragged chunk /
shard geometries."""
from __future__ import annotations

import json
import warnings
import zipfile
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
from scipy import sparse

import scanpy_amd as sc
from scanpy_amd import _zarr3 as z3
from scanpy_amd import readwrite as rw
from scanpy_amd._backed import BackedCsr

REF_ZIP = Path("/root/reference/src/scanpy/datasets/10x_pbmc68k_reduced.zarr.zip")
needs_reference = pytest.mark.skipif(not REF_ZIP.is_file(), reason="the reference checkout is not on this machine")
GOLDEN = Path(__file__).parent / "golden" / "pbmc68k_reduced.npz"


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert z3.crc32c(b"123456789") == 0xE3069283
    assert z3.crc32c(bytes(32)) == 0x8A9136AA
    assert z3.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert z3.crc32c(bytes(range(32))) == 0x46DD794E


@needs_reference
def test_reads_the_store_anndata_wrote():
    """every element of the reference fixture decodes, and the numeric ones equal the golden extraction"""
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # nothing in this store may be skipped
        a = sc.read_zarr(REF_ZIP)
    g = np.load(GOLDEN)
    assert a.shape == (700, 765)
    np.testing.assert_array_equal(a.X, g["X"])
    np.testing.assert_array_equal(a.obsm["X_pca"], g["X_pca"])
    np.testing.assert_array_equal(a.obs["louvain"].cat.codes.to_numpy(), g["louvain_codes"])
    np.testing.assert_array_equal(a.obs["bulk_labels"].cat.codes.to_numpy(), g["bulk_labels_codes"])
    np.testing.assert_array_equal(a.var["highly_variable"].to_numpy(), g["highly_variable"])
    np.testing.assert_array_equal(a.obs["n_counts"].to_numpy(), g["obs_n_counts"])
    for name, m in (("counts", a.layers["counts"]), ("distances", a.obsp["distances"]),
                    ("connectivities", a.obsp["connectivities"])):
        assert sparse.isspmatrix_csr(m) and m.shape == tuple(g[f"{name}_shape"])
        np.testing.assert_array_equal(m.data, g[f"{name}_data"])
        np.testing.assert_array_equal(m.indices, g[f"{name}_indices"])
        np.testing.assert_array_equal(m.indptr, g[f"{name}_indptr"])
    assert a.uns["neighbors"]["params"]["method"] == "umap"
    assert int(a.uns["neighbors"]["params"]["n_neighbors"][0]) == int(g["n_neighbors"][0]) == 10
    assert a.obs_names[0] == "AAAGCCTGGCTAAC-1" and list(a.obs["louvain"].cat.categories) == [str(i) for i in range(11)]
    names = a.uns["rank_genes_groups"]["names"]  # a record array (zarr `struct` of fixed_length_utf32)
    assert names.dtype.names[0] == "CD4+/CD25 T Reg" and names[0][0] == "RGS19"


@needs_reference
def test_backed_csr_of_the_reference_store_and_range_reads():
    st = z3.open_store(REF_ZIP)
    root = z3.Group(st)
    g = np.load(GOLDEN)
    b = BackedCsr(root["layers"]["counts"])
    assert b.shape == (700, 765) and b.nnz == g["counts_data"].size
    full = sparse.csr_matrix((g["counts_data"], g["counts_indices"], g["counts_indptr"]), shape=b.shape)
    for i0, i1 in ((0, 700), (0, 1), (13, 13), (100, 333), (699, 700)):
        got = b.rows(i0, i1).to_scipy()
        assert (got != full[i0:i1]).nnz == 0 and got.shape == (i1 - i0, 765)
    # dense 2-d array, chunks (175, 383): row ranges cutting through the chunk grid
    x = root["X"]
    for i0, i1 in ((0, 700), (170, 180), (349, 351), (5, 5)):
        np.testing.assert_array_equal(x.read(i0, i1), g["X"][i0:i1])
    with pytest.raises(ValueError, match="not a csr_matrix"):
        BackedCsr(root["obsm"])


@needs_reference
def test_written_metadata_has_the_keys_zarr_python_writes(tmp_path):
    """our writer's array / group documents vs zarr-python's, key by key (values that depend on the data aside)"""
    a = sc.read_zarr(REF_ZIP)
    sc.write_zarr(tmp_path / "w.zarr", a)
    z = zipfile.ZipFile(REF_ZIP)

    def strip(meta):
        meta = json.loads(json.dumps(meta))
        meta.pop("consolidated_metadata", None)  # optional; zarr-python falls back to the per-node documents
        if meta["node_type"] == "array":
            meta["chunk_grid"]["configuration"].pop("chunk_shape")
            if meta["codecs"][0]["name"] == "sharding_indexed":
                meta["codecs"][0]["configuration"].pop("chunk_shape")
        return meta

    for node in ("", "X", "obs", "obs/louvain", "obs/louvain/codes", "obs/louvain/categories", "obsp/distances",
                 "obsp/distances/data", "obsp/distances/indptr", "uns/neighbors/params/method", "uns/pca/variance",
                 "uns/rank_genes_groups/names", "var/highly_variable", "layers/counts"):
        key = f"{node}/zarr.json" if node else "zarr.json"
        theirs = strip(json.loads(z.read(key)))
        ours = strip(json.loads((tmp_path / "w.zarr" / key).read_text()))
        if node == "obs":  # we name an unnamed index `_index` (anndata's default); the fixture's is called `index`
            theirs["attributes"]["_index"] = ours["attributes"]["_index"] = "_index"
        assert ours == theirs, node


def _toy_adata(n=300, g=40, seed=0):
    rng = np.random.default_rng(seed)
    x = sparse.random(n, g, density=0.2, format="csr", dtype=np.float32, random_state=seed)
    obs = pd.DataFrame({"n": rng.integers(0, 9, n), "f": rng.random(n).astype(np.float32),
                        "c": pd.Categorical(rng.choice(["a", "b", "ccc"], n)), "s": rng.choice(["x", "yy"], n),
                        "nb": pd.array(rng.choice([True, False, None], n), dtype="boolean"),
                        "ni": pd.array(rng.choice([1, 2, None], n), dtype="Int64")},
                       index=[f"cell{i}" for i in range(n)])
    var = pd.DataFrame({"hv": rng.random(g) < 0.5}, index=[f"g{i}" for i in range(g)])
    return sc.AnnData(x, obs, var, obsm={"X_pca": rng.random((n, 5)).astype(np.float32)},
                      obsp={"connectivities": sparse.random(n, n, density=0.02, format="csr", random_state=1)},
                      layers={"dense": rng.random((n, g))},
                      uns={"k": 3, "pi": 3.25, "flag": True, "name": "héllo", "empty": np.zeros(0, dtype=np.int32),
                           "nested": {"a": np.arange(4), "colors": np.array(["#fff", "#000000"]), "csc": x.tocsc()}})


def test_round_trip_every_encoding(tmp_path):
    a = _toy_adata()
    sc.write_zarr(tmp_path / "a.zarr", a)
    b = sc.read_zarr(tmp_path / "a.zarr")
    assert (b.X != a.X).nnz == 0 and b.X.dtype == np.float32 and b.X.indices.dtype == a.X.indices.dtype
    pd.testing.assert_frame_equal(b.obs, a.obs.assign(s=a.obs["s"].astype(object)), check_index_type=False)
    pd.testing.assert_frame_equal(b.var, a.var, check_index_type=False)
    np.testing.assert_array_equal(b.obsm["X_pca"], a.obsm["X_pca"])
    np.testing.assert_array_equal(b.layers["dense"], a.layers["dense"])
    assert (b.obsp["connectivities"] != a.obsp["connectivities"]).nnz == 0
    u = b.uns
    assert u["k"] == 3 and u["pi"] == 3.25 and bool(u["flag"]) is True and u["name"] == "héllo"
    assert u["empty"].shape == (0,) and u["empty"].dtype == np.int32
    np.testing.assert_array_equal(u["nested"]["a"], np.arange(4))
    assert list(u["nested"]["colors"]) == ["#fff", "#000000"]
    assert sparse.isspmatrix_csc(u["nested"]["csc"]) and (u["nested"]["csc"] != a.X.tocsc()).nnz == 0


@pytest.mark.parametrize(("chunk", "per_shard"), [(1 << 22, 64), (997, 1), (1009, 5), (64, 3)])
def test_backed_rows_equal_memory_slices_for_ragged_geometries(tmp_path, monkeypatch, chunk, per_shard):
    monkeypatch.setattr(rw, "CHUNK_ELEMS", chunk)
    monkeypatch.setattr(rw, "CHUNKS_PER_SHARD", per_shard)
    rng = np.random.default_rng(3)
    x = sparse.random(2000, 70, density=0.1, format="csr", dtype=np.float32, random_state=3)
    x.data[:] = rng.random(x.nnz, dtype=np.float32) + 0.5
    x[17] = 0  # empty rows
    x[1999] = 0
    x.eliminate_zeros()
    sc.write_zarr(tmp_path / "x.zarr", sc.AnnData(x))
    a = sc.read_zarr(tmp_path / "x.zarr", backed="r")
    b = a.X
    assert isinstance(b, BackedCsr) and b.shape == x.shape and b.nnz == x.nnz and a.shape == x.shape
    for i0, i1 in ((0, 2000), (0, 0), (16, 19), (500, 1500), (1998, 2000), (2000, 2000)):
        r = b.rows(i0, i1)
        ref = x[i0:i1]
        np.testing.assert_array_equal(r.indptr, ref.indptr)
        np.testing.assert_array_equal(r.indices, ref.indices)
        np.testing.assert_array_equal(r.data, ref.data)
        assert r.indptr.dtype == np.int64 and r.data.dtype == np.float32
    mask = rng.random(70) < 0.4
    sub = b[:, mask]
    assert sub.shape == (2000, int(mask.sum()))
    assert (sub.rows(3, 1234).to_scipy() != x[3:1234][:, mask]).nnz == 0
    assert (sub[:, np.arange(0, sub.shape[1], 2)].rows(0, 50).to_scipy() != x[:50][:, mask][:, ::2]).nnz == 0
    assert (b[10:20] != x[10:20]).nnz == 0 and b[:] is b and (b.to_memory() != x).nnz == 0
    chunks = b.row_chunks(600)
    assert [c.shape[0] for c in chunks] == [600, 600, 600, 200]
    assert sum(c.nbytes for c in chunks) == 8 * x.nnz + 8 * (2000 + 4)
    with pytest.raises(IndexError):
        b.rows(5, 2001)
    with pytest.raises(IndexError):
        b[[1, 2, 3]]


def test_unsorted_rows_on_disk_are_sorted_on_load(tmp_path):
    x = sparse.random(50, 30, density=0.3, format="csr", dtype=np.float32, random_state=0)
    shuffled = x.copy()
    for r in range(50):  # reverse every row's entries: same matrix, unsorted indices
        s, e = shuffled.indptr[r], shuffled.indptr[r + 1]
        shuffled.indices[s:e] = shuffled.indices[s:e][::-1].copy()
        shuffled.data[s:e] = shuffled.data[s:e][::-1].copy()
    shuffled.has_sorted_indices = False
    sc.write_zarr(tmp_path / "u.zarr", sc.AnnData(shuffled))
    r = sc.read_zarr(tmp_path / "u.zarr", backed="r").X.rows(5, 45)
    np.testing.assert_array_equal(r.indices, x[5:45].indices)
    np.testing.assert_array_equal(r.data, x[5:45].data)


def test_plain_and_gzip_codecs_absent_chunks_and_errors(tmp_path):
    """arrays as other writers lay them out: unsharded chunks, gzip, index at the start, missing (fill) chunks"""
    st = z3.open_store(tmp_path / "s", "w")
    z3.write_group(st, "")
    data = np.arange(100, dtype=np.int32)
    meta = {"shape": [100], "data_type": "int32", "zarr_format": 3, "node_type": "array", "fill_value": 7,
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [30]}},
            "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}}, "attributes": {},
            "codecs": [{"name": "bytes", "configuration": {"endian": "little"}},
                       {"name": "gzip", "configuration": {"level": 5}}, {"name": "crc32c"}]}
    st.set("a/zarr.json", json.dumps(meta).encode())
    pipe = z3._Pipeline(meta["codecs"], np.dtype("<i4"), "a")
    for c in (0, 1, 3):  # chunk 2 is absent -> fill value
        piece = np.full(30, 7, dtype=np.int32)
        part = data[30 * c:30 * c + 30]
        piece[:part.size] = part
        st.set(f"a/c/{c}", pipe.encode(piece))
    want = data.copy()
    want[60:90] = 7
    arr = z3.Group(st)["a"]
    np.testing.assert_array_equal(arr.read(), want)
    np.testing.assert_array_equal(arr[25:95], want[25:95])
    # "." separated keys + shard index at the start
    meta2 = json.loads(json.dumps(meta))
    meta2["chunk_key_encoding"] = {"name": "default", "configuration": {"separator": "."}}
    meta2["chunk_grid"]["configuration"]["chunk_shape"] = [60]
    inner = [{"name": "bytes", "configuration": {"endian": "little"}}, {"name": "zstd", "configuration": {"level": 1}}]
    meta2["codecs"] = [{"name": "sharding_indexed", "configuration": {
        "chunk_shape": [20], "codecs": inner, "index_codecs": [{"name": "bytes"}], "index_location": "start"}}]
    st.set("b/zarr.json", json.dumps(meta2).encode())
    p2 = z3._Pipeline(inner, np.dtype("<i4"), "b")
    for s in range(2):
        parts, index, pos = [], np.full((3, 2), 2 ** 64 - 1, dtype="<u8"), 3 * 16
        for j in range(3):
            lo = 60 * s + 20 * j
            if lo >= 100 or (s, j) == (0, 1):  # one absent inner chunk
                continue
            piece = np.full(20, 7, dtype=np.int32)
            piece[:min(20, 100 - lo)] = data[lo:lo + 20]
            enc = p2.encode(piece)
            index[j] = (pos, len(enc))
            parts.append(enc)
            pos += len(enc)
        st.set(f"b/c.{s}", index.tobytes() + b"".join(parts))
    want2 = data.copy()
    want2[20:40] = 7
    np.testing.assert_array_equal(z3.Group(st)["b"].read(), want2)
    np.testing.assert_array_equal(z3.Group(st)["b"].read(35, 81), want2[35:81])
    # corrupt index checksum / unsupported codec / wrong format
    sc.write_zarr(tmp_path / "c.zarr", sc.AnnData(np.eye(4, dtype=np.float32)))
    f = tmp_path / "c.zarr" / "X" / "c" / "0" / "0"
    raw = bytearray(f.read_bytes())
    raw[-1] ^= 0xFF
    f.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        sc.read_zarr(tmp_path / "c.zarr")
    meta3 = json.loads(json.dumps(meta))
    meta3["codecs"][1] = {"name": "blosc", "configuration": {}}
    st.set("d/zarr.json", json.dumps(meta3).encode())
    with pytest.raises(NotImplementedError, match="blosc"):
        z3.Group(st)["d"]
    st.set("e/zarr.json", json.dumps({"zarr_format": 2}).encode())
    with pytest.raises(ValueError, match="format 3"):
        z3.Group(st)["e"]
    with pytest.raises(ValueError, match="backed"):
        sc.read_zarr(tmp_path / "c.zarr", backed="r+")
    with pytest.raises(FileNotFoundError):
        sc.read_zarr(tmp_path / "nope.zarr")


def test_backed_needs_a_csr_x(tmp_path):
    sc.write_zarr(tmp_path / "csc.zarr", sc.AnnData(sparse.random(20, 10, density=0.3, format="csc", dtype=np.float32)))
    with pytest.raises(ValueError, match="csr_matrix"):
        sc.read_zarr(tmp_path / "csc.zarr", backed="r")
    a = sc.read_zarr(tmp_path / "csc.zarr")
    assert sparse.isspmatrix_csc(a.X)


@pytest.mark.parametrize("copies", [True, False])
def test_chunked_rows_reader_thread_and_recycled_buffers(tmp_path, monkeypatch, copies):
    """`_ChunkedRows` over lazy chunks: a reader thread loads one chunk ahead; when the backend's upload copies, two
    buffer pairs are recycled (never more), and every pass still hands over exactly the rows of the matrix"""
    from scanpy_amd.preprocessing._pca_solver import _ChunkedRows

    monkeypatch.setattr(rw, "CHUNK_ELEMS", 777)
    monkeypatch.setattr(rw, "CHUNKS_PER_SHARD", 3)
    x = sparse.random(1000, 50, density=0.15, format="csr", dtype=np.float32, random_state=5)
    sc.write_zarr(tmp_path / "x.zarr", sc.AnnData(x))
    b = sc.read_zarr(tmp_path / "x.zarr", backed="r").X

    class Backend:
        upload_copies = copies
        seen_buffers: set = set()

        def upload(self, c):
            self.seen_buffers.add(c.data.__array_interface__["data"][0] if c.data.base is None
                                  else c.data.base.__array_interface__["data"][0])
            m = c.to_scipy()
            return m.copy() if copies else m

    rows = _ChunkedRows(b.row_chunks(130), 50)
    assert rows.n_chunks == 8 and rows.n_rows == 1000 and not rows.resident
    be = Backend()
    for _ in range(2):  # two passes, like the Gram pass and the projection pass
        got = list(rows.handles(be))
        assert len(got) == 8
        assert (sparse.vstack(got) != x).nnz == 0
    if copies:
        assert len(rows._ring) == 2 and len(be.seen_buffers) == 2
    else:
        assert rows._ring is None and len(be.seen_buffers) > 2
    # resident mode keeps the uploaded handles: the second pass does not touch the disk
    res = _ChunkedRows(b.row_chunks(400), 50, resident_budget_bytes=1 << 30)
    first = list(res.handles(be))
    monkeypatch.setattr(type(b), "rows", lambda *a, **k: (_ for _ in ()).throw(AssertionError("read again")))
    assert [id(h) for h in res.handles(be)] == [id(h) for h in first]


def test_backed_absmax_from_the_value_array(tmp_path, monkeypatch):
    from scanpy_amd.preprocessing._pca_solver import _ChunkedRows

    monkeypatch.setattr(rw, "CHUNK_ELEMS", 501)
    monkeypatch.setattr(rw, "CHUNKS_PER_SHARD", 2)
    x = sparse.random(800, 30, density=0.2, format="csr", dtype=np.float32, random_state=2)
    x.data -= 0.3  # negative values too
    x.data[1234] = -7.5
    sc.write_zarr(tmp_path / "x.zarr", sc.AnnData(x))
    b = sc.read_zarr(tmp_path / "x.zarr", backed="r").X
    assert b.absmax() == 7.5 == float(np.abs(x.data).max())
    assert b[:, np.arange(30) % 2 == 0].absmax() is None  # excluded columns would count
    assert _ChunkedRows(b.row_chunks(100), 30).host_absmax() == 7.5
    assert _ChunkedRows(b.row_chunks(100)[:-1], 30).host_absmax() is None  # not the whole matrix
    assert _ChunkedRows([x[:400], x[400:]], 30).host_absmax() is None  # in-memory chunks: the device pass answers
    sc.write_zarr(tmp_path / "e.zarr", sc.AnnData(sparse.csr_matrix((5, 4), dtype=np.float32)))
    assert sc.read_zarr(tmp_path / "e.zarr", backed="r").X.absmax() == 0.0


def test_staging_buffers_come_from_the_backend_when_it_offers_them(tmp_path, monkeypatch):
    """`GpuBackend.host_buffers` (page-locked when SCAMD_PIN_STAGING=1) is where `_ChunkedRows` gets its two decode
    buffer pairs from; without the switch they are plain numpy arrays of the on-disk dtypes"""
    from scanpy_amd.preprocessing._pca_solver import GpuBackend, _ChunkedRows

    x = sparse.random(500, 20, density=0.2, format="csr", dtype=np.float32, random_state=0)
    sc.write_zarr(tmp_path / "x.zarr", sc.AnnData(x))
    b = sc.read_zarr(tmp_path / "x.zarr", backed="r").X
    handed = []

    class Backend:
        upload_copies = True

        def host_buffers(self, n, idt, vdt):
            pair = GpuBackend.host_buffers(self, n, idt, vdt)
            handed.append(pair)
            return pair

        def upload(self, c):
            return c.to_scipy().copy()

    monkeypatch.delenv("SCAMD_PIN_STAGING", raising=False)
    rows = _ChunkedRows(b.row_chunks(120), 20)
    got = list(rows.handles(Backend()))
    assert (sparse.vstack(got) != x).nnz == 0
    assert len(handed) == 2 and all(p[0].dtype == np.int32 and p[1].dtype == np.float32 for p in handed)
    assert rows._ring[0][0] is handed[0][0]
