"""Zarr format 2 stores and Blosc frames (scanpy_amd/_zarr2.py).

Pins: (1) the reference's own v2 test store `tests/_data/10x-10k-subset.zarr` (Blosc-LZ4 + byte shuffle, split and
unsplit blocks, record arrays), read in place when the checkout is on this machine: the decoded matrix must be what a
count matrix is -- non-negative integers, ~7 % non-zero -- and the names must be barcodes and mouse genes; one 14 KB
chunk of it (`var/0`, real c-blosc output) is committed as tests/golden/zarr2/var_chunk.blosc; (2) stores and Blosc
frames built by this file's own encoder (LZ4 through liblz4), covering every branch of the frame decoder."""
from __future__ import annotations

import ctypes
import json
import struct
import zlib
from pathlib import Path

import numpy as np
import pytest

import scanpy_amd as sc
from scanpy_amd import _zarr2 as z2
from scanpy_amd import _zarr3 as z3

REF = Path("/root/reference/tests/_data/10x-10k-subset.zarr")
GOLD = Path(__file__).parent / "golden" / "zarr2"


def test_real_cblosc_chunk_of_record_strings():
    """`var/0` of the reference store: typesize 36, blocks not split (flag 0x10), LZ4, byte shuffle"""
    meta = json.loads((GOLD / "var.zarray.json").read_text())
    dt = z2._dtype(meta["dtype"])
    frame = (GOLD / "var_chunk.blosc").read_bytes()
    out = np.empty(meta["chunks"][0], dtype=dt)
    z2.blosc_decompress_into(frame, out.view(np.uint8).reshape(-1))
    assert out.dtype.names == ("index", "gene_ids")
    assert [s.decode() for s in out["index"][:3]] == ["Xkr4", "Gm1992", "Gm37381"]
    assert [s.decode() for s in out["gene_ids"][:3]] == ["ENSMUSG00000051951", "ENSMUSG00000089699",
                                                         "ENSMUSG00000102343"]
    assert all(s.startswith(b"ENSMUSG") and len(s) == 18 for s in out["gene_ids"])
    with pytest.raises(ValueError, match="expected"):
        z2.blosc_decompress_into(frame, np.empty(10, dtype=np.uint8))


@pytest.mark.skipif(not REF.is_dir(), reason="the reference checkout is not on this machine")
def test_reads_the_reference_v2_store():
    a = sc.read_zarr(REF)
    assert a.shape == (10000, 1000) and a.X.dtype == np.float32
    x = a.X
    assert x.min() == 0 and np.array_equal(x, np.round(x)) and 0.05 < (x != 0).mean() < 0.1  # a count matrix
    import re

    assert a.obs_names[0] == "AAACCTGAGATAGGAG-1" and all(re.fullmatch(r"[ACGT]{16}-[12]", s) for s in a.obs_names)
    assert list(a.var_names[:2]) == ["Xkr4", "Gm1992"] and a.var["gene_ids"].iloc[0] == "ENSMUSG00000051951"
    arr = z3.open_root(z3.open_store(REF))["X"]  # chunks (2000, 1000): a range that cuts two of them
    np.testing.assert_array_equal(arr.read(1990, 4020), x[1990:4020])


# -- an encoder for the tests -------------------------------------------------------------------------------------------

def _lz4_compress(raw: bytes) -> bytes:
    lib = ctypes.CDLL("liblz4.so.1")
    lib.LZ4_compress_default.restype = ctypes.c_int
    lib.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.LZ4_compressBound.restype = ctypes.c_int
    cap = lib.LZ4_compressBound(len(raw))
    out = ctypes.create_string_buffer(cap)
    n = lib.LZ4_compress_default(raw, out, len(raw), cap)
    assert n > 0
    return out.raw[:n]


def blosc_frame(raw: bytes, typesize: int, blocksize: int, *, codec: str = "lz4", shuffle: bool = True,
                dont_split: bool = False) -> bytes:
    codec_id = {"lz4": 1, "zlib": 3, "zstd": 4}[codec]
    compress = {"lz4": _lz4_compress, "zlib": lambda b: zlib.compress(b, 5),
                "zstd": lambda b: z3._zstd.compress(b, 3)}[codec]
    nbytes = len(raw)
    nblocks = -(-nbytes // blocksize)
    body, starts = b"", []
    for b in range(nblocks):
        blk = raw[b * blocksize:(b + 1) * blocksize]
        if shuffle and typesize > 1:
            n = len(blk) // typesize
            arr = np.frombuffer(blk, np.uint8)
            blk = np.ascontiguousarray(arr[:n * typesize].reshape(n, typesize).T).tobytes() + blk[n * typesize:]
        split = not dont_split and len(blk) == blocksize and typesize <= 16 and blocksize // typesize >= 128
        nsplit = typesize if split else 1
        ne = len(blk) // nsplit
        starts.append(16 + 4 * nblocks + len(body))
        for s in range(nsplit):
            piece = blk[s * ne:(s + 1) * ne]
            c = compress(piece)
            if len(c) >= ne:  # incompressible: stored raw, recognisable by its size
                c = piece
            body += struct.pack("<i", len(c)) + c
    flags = (1 if shuffle and typesize > 1 else 0) | (0x10 if dont_split else 0) | (codec_id << 5)
    head = struct.pack("<BBBBIII", 2, 1, flags, typesize, nbytes, blocksize, 16 + 4 * nblocks + len(body))
    return head + struct.pack(f"<{nblocks}i", *starts) + body


@pytest.mark.parametrize("codec", ["lz4", "zlib", "zstd"])
@pytest.mark.parametrize(("typesize", "blocksize", "dont_split"), [(4, 4096, False), (4, 4096, True), (8, 2048, False),
                                                                    (1, 1024, False), (18, 1800, False), (4, 256, False)])
def test_blosc_frames_of_every_shape(codec, typesize, blocksize, dont_split):
    rng = np.random.default_rng(typesize * blocksize)
    n = 10_007  # not a multiple of anything: a leftover block, and leftover bytes inside the shuffle
    raw = (np.cumsum(rng.integers(0, 3, n)).astype(np.uint8).tobytes() + rng.bytes(300))  # compressible + raw-stored parts
    raw = raw[:len(raw) // typesize * typesize]
    frame = blosc_frame(raw, typesize, blocksize, codec=codec, dont_split=dont_split)
    out = np.empty(len(raw), dtype=np.uint8)
    z2.blosc_decompress_into(frame, out)
    assert out.tobytes() == raw
    plain = blosc_frame(raw, typesize, blocksize, codec=codec, shuffle=False)
    z2.blosc_decompress_into(plain, out)
    assert out.tobytes() == raw


def test_blosc_memcpy_frames_and_refusals():
    raw = bytes(range(200))
    out = np.empty(200, dtype=np.uint8)
    z2.blosc_decompress_into(struct.pack("<BBBBIII", 2, 1, 0x02 | 0x01, 4, 200, 200, 216) + raw, out)
    assert out.tobytes() == raw
    with pytest.raises(NotImplementedError, match="bit shuffle"):
        z2.blosc_decompress_into(struct.pack("<BBBBIII", 2, 1, 0x04 | (1 << 5), 4, 200, 200, 216) + raw, out)
    with pytest.raises(NotImplementedError, match="blosclz"):
        z2.blosc_decompress_into(struct.pack("<BBBBIII", 2, 1, 0x01, 4, 200, 200, 216) + raw, out)


def _write_v2(root: Path, path: str, arr, chunks, compressor, encode, *, filters=None, sep=".", attrs=None):
    d = root / path
    d.mkdir(parents=True, exist_ok=True)
    is_obj = arr.dtype.kind == "O"
    meta = {"zarr_format": 2, "shape": list(arr.shape), "chunks": list(chunks), "order": "C", "filters": filters,
            "dtype": "|O" if is_obj else arr.dtype.str if not arr.dtype.names else
            [[n, arr.dtype.fields[n][0].str] for n in arr.dtype.names],
            "compressor": compressor, "fill_value": "" if is_obj else 0}
    if sep != ".":
        meta["dimension_separator"] = sep
    (d / ".zarray").write_text(json.dumps(meta))
    if attrs:
        (d / ".zattrs").write_text(json.dumps(attrs))
    grid = [range(-(-s // c)) for s, c in zip(arr.shape, chunks)]
    import itertools

    for idx in itertools.product(*grid):
        sel = tuple(slice(i * c, (i + 1) * c) for i, c in zip(idx, chunks))
        piece = arr[sel]
        if is_obj:
            full = np.full(chunks, "", dtype=object)
            full[tuple(slice(0, n) for n in piece.shape)] = piece
            raw = struct.pack("<I", full.size) + b"".join(struct.pack("<I", len(s.encode())) + s.encode()
                                                         for s in full.reshape(-1))
        else:
            full = np.zeros(chunks, dtype=arr.dtype)
            full[tuple(slice(0, n) for n in piece.shape)] = piece
            raw = full.tobytes()
        key = sep.join(str(i) for i in idx)
        target = d / key
        target.parent.mkdir(parents=True, exist_ok=True)
        target.write_bytes(encode(raw, arr.dtype.itemsize if not is_obj else 1))


def test_v2_anndata_store_with_the_encodings_of_anndata_0_8(tmp_path):
    """groups / `.zattrs` encodings as anndata 0.7-0.10 wrote them to zarr v2: csr X, dataframe obs with a vlen-utf8
    index and a categorical, Blosc-LZ4 chunks ('.' and '/' chunk keys), zlib and uncompressed arrays"""
    from scipy import sparse

    rng = np.random.default_rng(0)
    n, g = 700, 40
    x = sparse.random(n, g, density=0.2, format="csr", dtype=np.float32, random_state=0)
    root = tmp_path / "v2.zarr"
    root.mkdir()
    (root / ".zgroup").write_text('{"zarr_format": 2}')
    (root / ".zattrs").write_text(json.dumps({"encoding-type": "anndata", "encoding-version": "0.1.0"}))

    def group(path, attrs):
        (root / path).mkdir(parents=True, exist_ok=True)
        (root / path / ".zgroup").write_text('{"zarr_format": 2}')
        (root / path / ".zattrs").write_text(json.dumps(attrs))

    blosc = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}

    def enc_blosc(raw, ts):
        return blosc_frame(raw, ts, 512)

    group("X", {"encoding-type": "csr_matrix", "encoding-version": "0.1.0", "shape": [n, g]})
    _write_v2(root, "X/data", x.data, (1000,), blosc, enc_blosc)
    _write_v2(root, "X/indices", x.indices, (777,), blosc, enc_blosc, sep="/")
    _write_v2(root, "X/indptr", x.indptr, (n + 1,), {"id": "zlib", "level": 1}, lambda raw, ts: zlib.compress(raw, 1))
    group("obs", {"encoding-type": "dataframe", "encoding-version": "0.2.0", "_index": "_index",
                  "column-order": ["louvain", "n"]})
    names = np.array([f"cell-{i}" for i in range(n)], dtype=object)
    _write_v2(root, "obs/_index", names, (256,), blosc, lambda raw, ts: blosc_frame(raw, 1, 4096),
              filters=[{"id": "vlen-utf8"}], attrs={"encoding-type": "string-array", "encoding-version": "0.2.0"})
    group("obs/louvain", {"encoding-type": "categorical", "encoding-version": "0.2.0", "ordered": False})
    codes = rng.integers(0, 3, n).astype(np.int8)
    _write_v2(root, "obs/louvain/codes", codes, (n,), None, lambda raw, ts: raw)
    _write_v2(root, "obs/louvain/categories", np.array(["a", "b", "c"], dtype=object), (3,), None, lambda raw, ts: raw,
              filters=[{"id": "vlen-utf8"}], attrs={"encoding-type": "string-array", "encoding-version": "0.2.0"})
    nn = rng.integers(0, 100, n)
    _write_v2(root, "obs/n", nn, (300,), blosc, enc_blosc, attrs={"encoding-type": "array", "encoding-version": "0.2.0"})
    # ... and a categorical the way anndata 0.7.x wrote it: codes + `categories` = a path into `__categories`
    (root / "obs" / ".zattrs").write_text(json.dumps({"encoding-type": "dataframe", "encoding-version": "0.1.0",
                                                       "_index": "_index", "column-order": ["louvain", "n", "phase"]}))
    (root / "obs" / "__categories").mkdir()
    (root / "obs" / "__categories" / ".zgroup").write_text('{"zarr_format": 2}')
    _write_v2(root, "obs/__categories/phase", np.array(["G1", "S"], dtype=object), (2,), None, lambda raw, ts: raw,
              filters=[{"id": "vlen-utf8"}], attrs={"ordered": True})
    phase = rng.integers(0, 2, n).astype(np.int8)
    _write_v2(root, "obs/phase", phase, (n,), None, lambda raw, ts: raw, attrs={"categories": "__categories/phase"})
    group("var", {"encoding-type": "dataframe", "encoding-version": "0.2.0", "_index": "_index", "column-order": []})
    _write_v2(root, "var/_index", np.array([f"g{i}" for i in range(g)], dtype=object), (g,), None, lambda raw, ts: raw,
              filters=[{"id": "vlen-utf8"}])
    group("obsm", {"encoding-type": "dict", "encoding-version": "0.1.0"})
    pca = rng.standard_normal((n, 5)).astype(np.float32)
    _write_v2(root, "obsm/X_pca", pca, (256, 5), blosc, enc_blosc)

    a = sc.read_zarr(root)
    assert (a.X != x).nnz == 0 and list(a.obs_names) == list(names) and list(a.obs.columns) == ["louvain", "n", "phase"]
    assert list(a.obs["phase"].cat.categories) == ["G1", "S"] and a.obs["phase"].cat.ordered
    np.testing.assert_array_equal(a.obs["phase"].cat.codes.to_numpy(), phase)
    np.testing.assert_array_equal(a.obs["louvain"].cat.codes.to_numpy(), codes)
    np.testing.assert_array_equal(a.obs["n"].to_numpy(), nn)
    np.testing.assert_array_equal(a.obsm["X_pca"], pca)
    assert a.var_names[7] == "g7"
    b = sc.read_zarr(root, backed="r")
    assert b.X.is_backed and (b.X.rows(123, 599).to_scipy() != x[123:599]).nnz == 0
    with pytest.raises(ValueError, match="not a zarr store"):
        sc.read_zarr(tmp_path)
