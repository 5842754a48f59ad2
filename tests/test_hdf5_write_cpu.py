"""`write_h5ad` / `write` / `read` (scanpy_amd/_hdf5_write.py, readwrite.py).

Two readers check every file: our own (`_hdf5.py`, pinned to library-written files in test_hdf5_cpu.py) and -- where an
interpreter with h5py exists, which in this image is /opt/conda/bin/python3.9 -- THE HDF5 LIBRARY ITSELF, run as a
subprocess (`tests/h5py_check.py`, under a timeout: a malformed B-tree or heap makes the library loop, not fail)."""
from __future__ import annotations

import json
import subprocess
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
from scipy import sparse

import scanpy_amd as sc
from scanpy_amd import _hdf5 as h5
from scanpy_amd import _hdf5_write as hw

H5PY_PYTHON = Path("/opt/conda/bin/python3.9")


def _has_h5py() -> bool:
    if not H5PY_PYTHON.is_file():
        return False
    try:
        return subprocess.run([str(H5PY_PYTHON), "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except (OSError, subprocess.TimeoutExpired):
        return False


needs_h5py = pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py on this machine")


def _library_view(path) -> dict:
    res = subprocess.run([str(H5PY_PYTHON), str(Path(__file__).parent / "h5py_check.py"), str(path)],
                         capture_output=True, text=True, timeout=180)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout)


def _toy_adata(n=400, g=50, seed=0):
    rng = np.random.default_rng(seed)
    x = sparse.random(n, g, density=0.2, format="csr", dtype=np.float32, random_state=seed)
    obs = pd.DataFrame({"n": rng.integers(0, 9, n), "f": rng.random(n).astype(np.float32),
                        "c": pd.Categorical(rng.choice(["a", "b", "ccc"], n)), "s": rng.choice(["x", "yé"], n),
                        "nb": pd.array(rng.choice([True, False, None], n), dtype="boolean"),
                        "ni": pd.array(rng.choice([1, 2, None], n), dtype="Int64")},
                       index=[f"cell{i}" for i in range(n)])
    var = pd.DataFrame({"hv": rng.random(g) < 0.5}, index=[f"g{i}" for i in range(g)])
    rec = np.zeros(3, dtype=[("a", "<f4"), ("b", "<i8")])
    rec["a"], rec["b"] = [1.5, 2.5, 3.5], [1, 2, 3]
    return sc.AnnData(x, obs, var, obsm={"X_pca": rng.random((n, 5)).astype(np.float32)},
                      obsp={"connectivities": sparse.random(n, n, density=0.02, format="csr", random_state=1)},
                      layers={"dense": rng.random((n, g))},
                      uns={"k": 3, "pi": 3.25, "flag": True, "name": "héllo", "empty": np.zeros(0, dtype=np.int32),
                           "rec": rec, "nested": {"a": np.arange(4), "colors": np.array(["#fff", "#000000"]),
                                                  "csc": x.tocsc(), "deeper": {"z": "q"}}})


def _assert_same(b, a):
    assert (b.X != a.X).nnz == 0 and b.X.dtype == a.X.dtype and b.X.indices.dtype == a.X.indices.dtype
    pd.testing.assert_frame_equal(b.obs, a.obs.assign(s=a.obs["s"].astype(object)), check_index_type=False)
    pd.testing.assert_frame_equal(b.var, a.var, check_index_type=False)
    np.testing.assert_array_equal(b.obsm["X_pca"], a.obsm["X_pca"])
    np.testing.assert_array_equal(b.layers["dense"], a.layers["dense"])
    assert (b.obsp["connectivities"] != a.obsp["connectivities"]).nnz == 0
    u = b.uns
    assert u["k"] == 3 and u["pi"] == 3.25 and bool(u["flag"]) is True and u["name"] == "héllo"
    assert u["empty"].shape == (0,) and u["empty"].dtype == np.int32
    np.testing.assert_array_equal(u["rec"], a.uns["rec"])
    np.testing.assert_array_equal(u["nested"]["a"], np.arange(4))
    assert list(u["nested"]["colors"]) == ["#fff", "#000000"] and u["nested"]["deeper"]["z"] == "q"
    assert sparse.isspmatrix_csc(u["nested"]["csc"]) and (u["nested"]["csc"] != a.X.tocsc()).nnz == 0


@pytest.mark.parametrize("compression", [None, "gzip"])
def test_write_h5ad_round_trip(tmp_path, compression, monkeypatch):
    monkeypatch.setattr(hw, "CHUNK_BYTES", 1 << 12)  # several chunks per dataset at this size
    monkeypatch.setattr("scanpy_amd.readwrite._H5Sink.SMALL", 256)
    a = _toy_adata()
    sc.write_h5ad(tmp_path / "a.h5ad", a, compression=compression)
    _assert_same(sc.read_h5ad(tmp_path / "a.h5ad"), a)
    back = sc.read_h5ad(tmp_path / "a.h5ad", backed="r")
    assert back.X.is_backed and (back.X.rows(37, 311).to_scipy() != a.X[37:311]).nnz == 0
    with h5.File(tmp_path / "a.h5ad") as f:
        d = f["X"]["data"]
        assert (d.layout[0] == "chunked") == (compression == "gzip")
        assert [fid for fid, _ in d.filters] == ([2, 1] if compression else [])
    with pytest.raises(NotImplementedError, match="lzf"):
        sc.write_h5ad(tmp_path / "b.h5ad", a, compression="lzf")


@needs_h5py
@pytest.mark.parametrize("compression", [None, "gzip"])
def test_the_hdf5_library_reads_what_we_write(tmp_path, compression, monkeypatch):
    monkeypatch.setattr(hw, "CHUNK_BYTES", 1 << 12)
    monkeypatch.setattr("scanpy_amd.readwrite._H5Sink.SMALL", 256)
    a = _toy_adata()
    sc.write_h5ad(tmp_path / "a.h5ad", a, compression=compression)
    v = _library_view(tmp_path / "a.h5ad")
    assert v["attrs"] == {"encoding-type": "anndata", "encoding-version": "0.1.0"}
    kids = v["children"]
    assert sorted(kids) == ["X", "layers", "obs", "obsm", "obsp", "uns", "var", "varm", "varp"]
    x = kids["X"]
    assert x["attrs"] == {"encoding-type": "csr_matrix", "encoding-version": "0.1.0", "shape": [400, 50]}
    xd = x["children"]["data"]
    assert xd["shape"] == [a.X.nnz] and xd["dtype"] == "float32" and abs(xd["sum"] - float(a.X.data.sum(dtype=np.float64))) < 1e-3
    assert xd["compression"] == compression and xd["shuffle"] == (compression == "gzip")
    assert (xd["chunks"] is not None) == (compression == "gzip")
    assert x["children"]["indices"]["head"] == a.X.indices[:8].tolist()
    assert x["children"]["indptr"]["sum"] == float(a.X.indptr.sum())
    obs = kids["obs"]
    assert obs["attrs"]["_index"] == "_index" and obs["attrs"]["column-order"] == ["n", "f", "c", "s", "nb", "ni"]
    assert obs["attrs"]["encoding-type"] == "dataframe"
    assert obs["children"]["_index"]["strings"][:3] == ["cell0", "cell1", "cell2"]
    assert obs["children"]["_index"]["attrs"]["encoding-type"] == "string-array"
    assert obs["children"]["s"]["strings"] == list(a.obs["s"])
    cat = obs["children"]["c"]
    assert cat["attrs"] == {"encoding-type": "categorical", "encoding-version": "0.2.0", "ordered": False}
    assert cat["children"]["categories"]["strings"] == ["a", "b", "ccc"] and cat["children"]["codes"]["dtype"] == "int8"
    assert obs["children"]["nb"]["children"]["mask"]["dtype"] == "bool"  # h5py maps our enum back to numpy bool
    assert kids["var"]["children"]["hv"]["dtype"] == "bool"
    assert kids["var"]["children"]["hv"]["sum"] == float(a.var["hv"].sum())
    assert kids["obsm"]["children"]["X_pca"]["shape"] == [400, 5]
    assert abs(kids["layers"]["children"]["dense"]["sum"] - a.layers["dense"].sum()) < 1e-6
    uns = kids["uns"]["children"]
    assert uns["name"]["strings"] == "héllo" and uns["name"]["attrs"]["encoding-type"] == "string"
    assert uns["k"]["head"] == [3] and uns["pi"]["head"] == [3.25] and uns["flag"]["dtype"] == "bool"
    assert uns["empty"]["shape"] == [0] and uns["rec"]["fields"] == {"a": [1.5, 2.5, 3.5], "b": [1, 2, 3]}
    assert uns["nested"]["children"]["colors"]["strings"] == ["#fff", "#000000"]
    assert uns["nested"]["children"]["deeper"]["children"]["z"]["strings"] == "q"
    assert kids["varm"]["children"] == {} and kids["varp"]["attrs"]["encoding-type"] == "dict"


@needs_h5py
def test_wide_groups_many_chunks_and_long_string_columns(tmp_path, monkeypatch):
    """structures that need more than one node: > 2K symbol-table leaves, a multi-level chunk B-tree, strings over
    several global heap collections"""
    monkeypatch.setattr(hw, "CHUNK_BYTES", 256)
    root = hw.Node({}, is_group=True)
    wide = hw.Node({"encoding-type": "dict"}, is_group=True)
    root.children["wide"] = wide
    for i in range(2500):  # 79 leaves of 32 -> two levels of group B-tree nodes (64 children each)
        wide.children[f"k{i:05d}"] = hw.Node({}, np.int32(i))
    big = np.arange(300_000, dtype=np.int64)  # 32 values per chunk -> 9375 chunks: three B-tree levels
    root.children["big"] = hw.Node({}, big, compression="gzip")
    names = np.array([f"barcode-{i:07d}-{'x' * (i % 50)}" for i in range(70_000)], dtype=object)  # > 65535 per heap
    root.children["names"] = hw.Node({}, names)
    hw.write_tree(tmp_path / "w.h5", root)
    with h5.File(tmp_path / "w.h5") as f:
        assert f["wide"].keys() == [f"k{i:05d}" for i in range(2500)] and f["wide"]["k01234"][()] == 1234
        np.testing.assert_array_equal(f["big"].read(), big)
        np.testing.assert_array_equal(f["big"].read(12_345, 250_001), big[12_345:250_001])
        got = f["names"].read()
        assert got[0] == names[0] and got[-1] == names[-1] and (got == names).all()
    v = _library_view(tmp_path / "w.h5")
    assert len(v["children"]["wide"]["children"]) == 2500
    assert v["children"]["wide"]["children"]["k02499"]["head"] == [2499]
    assert v["children"]["big"]["sum"] == float(big.sum()) and v["children"]["big"]["chunks"] == [32]
    assert v["children"]["names"]["strings"] == list(names[:2000]) and v["children"]["names"]["shape"] == [70_000]


def test_write_and_read_dispatch_on_the_extension(tmp_path):
    """src/scanpy/readwrite.py:657-740 (`write`), :808-841 (`read`)"""
    a = _toy_adata(2000, 12)  # (big enough for `layers['dense']` to be chunked when compression is on)
    sc.write(tmp_path / "x.h5ad", a)
    sc.write(tmp_path / "noext", a, compression=None)
    sc.write(tmp_path / "y.zarr", a)
    for p in ("x.h5ad", "noext.h5ad", "y.zarr"):
        b = sc.read(tmp_path / p)
        assert (b.X != a.X).nnz == 0 and list(b.obs_names) == list(a.obs_names)
    assert sc.read(tmp_path / "x.h5ad", backed="r").X.is_backed
    with h5.File(tmp_path / "x.h5ad") as f:
        assert f["layers"]["dense"].filters and not h5.File(tmp_path / "noext.h5ad")["layers"]["dense"].filters
    with pytest.raises(ValueError, match="h5ad"):
        sc.write(tmp_path / "z.csv", a)
    with pytest.raises(ValueError, match="only 'h5ad'"):
        sc.read(tmp_path / "z.csv")
    with pytest.raises(TypeError, match="unexpected"):
        sc.read(tmp_path / "x.h5ad", delimiter=",")


@pytest.mark.parametrize("compression", [None, "gzip"])
@pytest.mark.parametrize("source", ["zarr", "h5ad"])
def test_backed_matrix_is_copied_block_by_block(tmp_path, monkeypatch, compression, source):
    """`write_h5ad` of an AnnData whose X is still on disk streams `data` / `indices` from the source store into the
    new file (never whole in memory); the copy equals the original"""
    from scanpy_amd import readwrite as rw
    from scanpy_amd._backed import BackedCsr

    monkeypatch.setattr(hw, "CHUNK_BYTES", 1 << 10)
    monkeypatch.setattr("scanpy_amd.readwrite._H5Sink.SMALL", 256)
    real_init = rw._BackedColumn.__init__
    monkeypatch.setattr(rw._BackedColumn, "__init__", lambda self, x, which, rows=97: real_init(self, x, which, rows))
    monkeypatch.setattr(BackedCsr, "to_memory", lambda self: (_ for _ in ()).throw(AssertionError("materialised")))
    a = _toy_adata(1000, 30)
    src = tmp_path / ("src.zarr" if source == "zarr" else "src.h5ad")
    sc.write(src, a, **({} if source == "zarr" else {"compression": None}))
    b = sc.read(src, backed="r")
    sc.write_h5ad(tmp_path / "copy.h5ad", b, compression=compression)
    monkeypatch.undo()
    c = sc.read_h5ad(tmp_path / "copy.h5ad")
    assert (c.X != a.X).nnz == 0 and c.X.dtype == a.X.dtype and list(c.obs_names) == list(a.obs_names)
    np.testing.assert_array_equal(c.X.indptr, a.X.indptr)
    if _has_h5py():
        v = _library_view(tmp_path / "copy.h5ad")["children"]["X"]["children"]
        assert v["data"]["shape"] == [a.X.nnz] and abs(v["data"]["sum"] - float(a.X.data.sum(dtype=np.float64))) < 1e-3
        assert v["indices"]["sum"] == float(a.X.indices.sum()) and v["data"]["compression"] == compression


@pytest.mark.parametrize("ext", ["h5ad", "zarr"])
def test_raw_round_trips(tmp_path, ext):
    """`adata.raw` (matrix + var before gene filtering; anndata's `raw` group) survives write -> read in both containers"""
    a = _toy_adata(200, 30)
    full = sparse.random(200, 45, density=0.2, format="csr", dtype=np.float32, random_state=9)
    a.raw = sc.AnnData(full, a.obs, pd.DataFrame({"gene_ids": [f"E{i}" for i in range(45)]},
                                                 index=[f"r{i}" for i in range(45)]))
    sc.write(tmp_path / f"a.{ext}", a, **({"compression": None} if ext == "h5ad" else {}))
    b = sc.read(tmp_path / f"a.{ext}")
    assert b.raw is not None and b.raw.shape == (200, 45) and (b.raw.X != full).nnz == 0
    assert list(b.raw.var_names) == [f"r{i}" for i in range(45)] and list(b.raw.var["gene_ids"])[:2] == ["E0", "E1"]
    assert list(b.raw.obs_names) == list(a.obs_names) and (b.X != a.X).nnz == 0
    c = sc.read(tmp_path / f"a.{ext}", backed="r")
    assert c.raw.X.is_backed and c.X.is_backed
    if ext == "h5ad" and _has_h5py():
        v = _library_view(tmp_path / "a.h5ad")["children"]["raw"]
        assert v["attrs"]["encoding-type"] == "raw" and sorted(v["children"]) == ["X", "var", "varm"]
        assert v["children"]["X"]["attrs"]["shape"] == [200, 45]
