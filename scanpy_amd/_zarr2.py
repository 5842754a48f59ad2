"""Zarr format 2 stores (what anndata < 0.11 and zarr-python 2 wrote: `.zgroup` / `.zarray` / `.zattrs`, one object per
chunk, numcodecs compressors -- Blosc by default), read-only, with the array / group protocol of `_zarr3.py`.

The reference's own zarr test data is such a store (`tests/_data/10x-10k-subset.zarr`, dense float32 `X` in Blosc-LZ4
chunks with byte shuffle, record-array `obs` / `var`), and so is most AnnData zarr on disk today.  Restated from the
zarr v2 specification and the c-blosc 1.x frame format:

    blosc frame = 16-byte header {version, versionlz, flags, typesize, nbytes u32, blocksize u32, cbytes u32}
                  flags: 0x1 byte shuffle, 0x2 stored uncompressed, 0x4 bit shuffle, 0x10 blocks not split,
                         bits 5-7 codec (0 blosclz, 1 lz4 / lz4hc, 2 snappy, 3 zlib, 4 zstd)
                  + i32 start offset per block; a block = 1 or `typesize` streams (split by byte significance), each
                  {i32 compressed size, data}; a stream whose size equals its decoded size is stored raw; after
                  decoding, the byte shuffle is undone per block

numcodecs, blosc and lz4 are not importable here: LZ4 goes through liblz4.so.1 (ctypes), the un-shuffle through
`scamd_unshuffle`; blosclz / snappy streams and the bit shuffle raise NotImplementedError.
"""
from __future__ import annotations

import base64
import ctypes
import itertools
import json
import struct
import threading
import zlib

import numpy as np

from ._zarr3 import _zstd, decode_pool

_lz4 = None


def _lz4_decompress_into(src: bytes, dst: np.ndarray) -> None:
    global _lz4
    if _lz4 is None:
        lib = ctypes.CDLL("liblz4.so.1")
        lib.LZ4_decompress_safe.restype = ctypes.c_int
        lib.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _lz4 = lib
    got = _lz4.LZ4_decompress_safe(src, dst.ctypes.data, len(src), dst.nbytes)
    if got != dst.nbytes:
        raise ValueError(f"corrupt lz4 stream (decoded {got} bytes, expected {dst.nbytes})")


def _unshuffle(src: np.ndarray, dst: np.ndarray, typesize: int) -> None:
    """blosc's byte shuffle of one block: the first (n // typesize) elements are stored as typesize planes, the
    remaining n % typesize bytes as they are"""
    from ._hdf5 import _unshuffle_into

    n = src.nbytes // typesize
    _unshuffle_into(src[:n * typesize], dst[:n * typesize].reshape(n, typesize))
    dst[n * typesize:] = src[n * typesize:]


def blosc_decompress_into(frame: bytes, dst: np.ndarray) -> None:
    """one Blosc 1.x frame -> `dst` (C-contiguous uint8 of the frame's `nbytes`)"""
    version, _, flags, typesize, nbytes, blocksize, cbytes = struct.unpack_from("<BBBBIII", frame, 0)
    if version != 2:
        raise NotImplementedError(f"blosc frame format version {version} is not read here")
    if nbytes != dst.nbytes:
        raise ValueError(f"blosc frame holds {nbytes} bytes, expected {dst.nbytes}")
    if cbytes > len(frame):
        raise ValueError("truncated blosc frame")
    if nbytes == 0:
        return
    if flags & 0x02:  # stored as is
        dst[:] = np.frombuffer(frame, np.uint8, count=nbytes, offset=16)
        return
    if flags & 0x04:
        raise NotImplementedError("blosc bit shuffle is not read here")
    codec = flags >> 5
    if codec not in (1, 3, 4):
        raise NotImplementedError(f"blosc codec {['blosclz', 'lz4', 'snappy', 'zlib', 'zstd'][codec] if codec < 5 else codec}"
                                  " is not read here (lz4, zlib and zstd are)")
    shuffled = bool(flags & 0x01) and typesize > 1
    dont_split = bool(flags & 0x10)
    nblocks = -(-nbytes // blocksize)
    starts = struct.unpack_from(f"<{nblocks}i", frame, 16)
    tmp = np.empty(blocksize, dtype=np.uint8) if shuffled else None
    for b in range(nblocks):
        lo = b * blocksize
        bsize = min(blocksize, nbytes - lo)
        leftover = bsize < blocksize
        split = not dont_split and not leftover and typesize <= 16 and blocksize // typesize >= 128
        nsplit = typesize if split else 1
        neblock = bsize // nsplit
        target = tmp[:bsize] if shuffled else dst[lo:lo + bsize]
        p = starts[b]
        for s in range(nsplit):
            (csize,) = struct.unpack_from("<i", frame, p)
            p += 4
            piece = target[s * neblock:(s + 1) * neblock]
            if csize == neblock:
                piece[:] = np.frombuffer(frame, np.uint8, count=neblock, offset=p)
            elif codec == 1:
                _lz4_decompress_into(frame[p:p + csize], piece)
            elif codec == 4:
                _zstd.decompress_into(frame[p:p + csize], piece)
            else:
                raw = zlib.decompress(frame[p:p + csize])
                if len(raw) != neblock:
                    raise ValueError("corrupt zlib stream in a blosc frame")
                piece[:] = np.frombuffer(raw, np.uint8)
            p += csize
        if shuffled:
            _unshuffle(target, dst[lo:lo + bsize], typesize)


# ---------------------------------------------------------------------------------------------------------------------


def _read_attrs(store, path: str) -> dict:
    key = f"{path}/.zattrs" if path else ".zattrs"
    return json.loads(store.get(key)) if store.exists(key) else {}


def _dtype(spec) -> np.dtype:
    if isinstance(spec, str):
        return np.dtype(spec)
    return np.dtype([(f[0], _dtype(f[1])) + ((tuple(f[2]),) if len(f) > 2 else ()) for f in spec])


class Array:
    """a zarr-v2 array opened for reading: `read(i0, i1)` = rows [i0, i1) along axis 0"""

    def __init__(self, store, path: str):
        self.store, self.path = store, path
        meta = json.loads(store.get(f"{path}/.zarray"))
        if meta.get("zarr_format") != 2:
            raise ValueError(f"{path!r}: not a zarr format 2 array")
        self.meta = meta
        self.attrs = _read_attrs(store, path)
        self.shape = tuple(int(s) for s in meta["shape"])
        self.ndim = len(self.shape)
        self.inner = tuple(int(c) for c in meta["chunks"]) if self.ndim else ()
        self.dtype = _dtype(meta["dtype"])
        self._sep = meta.get("dimension_separator", ".")
        if meta.get("order", "C") != "C" and self.ndim > 1:
            raise NotImplementedError(f"{path!r}: Fortran-ordered chunks are not read here")
        self.filters = [f["id"] for f in (meta.get("filters") or [])]
        self.is_string = self.dtype.kind == "O" or self.dtype.kind in "SU"
        if self.dtype.kind == "O" and self.filters != ["vlen-utf8"]:
            raise NotImplementedError(f"{path!r}: object arrays are read with the vlen-utf8 filter only "
                                      f"(found {self.filters})")
        if self.dtype.kind != "O" and self.filters:
            raise NotImplementedError(f"{path!r}: filters {self.filters} are not read here")
        comp = meta.get("compressor")
        self.codec = None if comp is None else comp["id"]
        if self.codec not in (None, "blosc", "zlib", "gzip", "zstd", "lz4"):
            raise NotImplementedError(f"{path!r}: compressor {self.codec!r} is not read here")
        self.fill = meta.get("fill_value")
        self._scratch = threading.local()

    def _fill_value(self):
        f = self.fill
        if self.dtype.kind == "O":
            return f if isinstance(f, str) else ""
        if self.dtype.names or self.dtype.kind in "SV":
            out = np.zeros((), dtype=self.dtype)
            if isinstance(f, str) and f:
                try:
                    out = np.frombuffer(base64.standard_b64decode(f), dtype=self.dtype, count=1)[0]
                except (ValueError, TypeError):
                    pass
            return out
        if isinstance(f, str):
            f = {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}.get(f, 0)
        return np.asarray(0 if f is None else f).astype(self.dtype)

    def _decode(self, raw: bytes, dst: np.ndarray) -> None:
        """one stored chunk -> `dst` (C-contiguous array of the chunk shape; fixed-size dtypes)"""
        flat = dst.reshape(-1).view(np.uint8)
        if self.codec is None:
            flat[:] = np.frombuffer(raw, np.uint8, count=flat.nbytes)
        elif self.codec == "blosc":
            blosc_decompress_into(raw, flat)
        elif self.codec == "zstd":
            _zstd.decompress_into(raw, flat)
        elif self.codec == "lz4":  # numcodecs.LZ4: u32 decoded size, then one lz4 block
            _lz4_decompress_into(raw[4:], flat)
        else:
            out = zlib.decompress(raw, 15 + 32)
            flat[:] = np.frombuffer(out, np.uint8, count=flat.nbytes)

    def _decode_strings(self, raw: bytes) -> np.ndarray:
        n_items = int(np.prod(self.inner)) if self.inner else 1
        if self.codec is not None:
            if self.codec == "blosc":
                (nbytes,) = struct.unpack_from("<I", raw, 4)
                buf = np.empty(nbytes, dtype=np.uint8)
                blosc_decompress_into(raw, buf)
                raw = buf.tobytes()
            elif self.codec == "zstd":
                raw = _zstd.decompress(raw)
            elif self.codec == "lz4":
                (nbytes,) = struct.unpack_from("<I", raw, 0)
                buf = np.empty(nbytes, dtype=np.uint8)
                _lz4_decompress_into(raw[4:], buf)
                raw = buf.tobytes()
            else:
                raw = zlib.decompress(raw, 15 + 32)
        (n,) = struct.unpack_from("<I", raw, 0)
        out = np.empty(n_items, dtype=object)
        out[:] = ""
        pos = 4
        for i in range(min(n, n_items)):
            (m,) = struct.unpack_from("<I", raw, pos)
            out[i] = raw[pos + 4:pos + 4 + m].decode("utf-8")
            pos += 4 + m
        return out.reshape(self.inner)

    def _load(self, out, i0: int, idx) -> None:
        origin = tuple(i * c for i, c in zip(idx, self.inner))
        key = f"{self.path}/" + (self._sep.join(str(i) for i in idx) if idx else "0")
        raw = self.store.get(key) if self.store.exists(key) else None
        a0, a1 = max(origin[0], i0), min(origin[0] + self.inner[0], i0 + out.shape[0])
        if a1 <= a0:
            return
        dst_sel = (slice(a0 - i0, a1 - i0),) + tuple(slice(o, min(o + c, s)) for c, s, o in
                                                      zip(self.inner[1:], self.shape[1:], origin[1:]))
        src_sel = (slice(a0 - origin[0], a1 - origin[0]),) + tuple(slice(0, d.stop - d.start) for d in dst_sel[1:])
        if raw is None:
            out[dst_sel] = self._fill_value()
            return
        if self.dtype.kind == "O":
            out[dst_sel] = self._decode_strings(raw)[src_sel]
            return
        whole = (a0 == origin[0] and a1 == origin[0] + self.inner[0]
                 and all(c == s for c, s in zip(self.inner[1:], self.shape[1:])))
        if whole:
            self._decode(raw, out[a0 - i0:a1 - i0])
            return
        tmp = getattr(self._scratch, "chunk", None)
        if tmp is None:
            tmp = self._scratch.chunk = np.empty(self.inner, dtype=self.dtype)
        self._decode(raw, tmp)
        out[dst_sel] = tmp[src_sel]

    def read(self, i0: int = 0, i1: int | None = None, *, out: np.ndarray | None = None,
             parallel: bool = True) -> np.ndarray:
        if self.ndim == 0:
            res = np.empty((), dtype=self.dtype)
            key = f"{self.path}/0"
            if not self.store.exists(key):
                res[...] = self._fill_value()
            elif self.dtype.kind == "O":
                res[...] = self._decode_strings(self.store.get(key)).reshape(-1)[0]
            else:
                self._decode(self.store.get(key), res.reshape(1))
            return res
        n0 = self.shape[0]
        i1 = n0 if i1 is None else i1
        if not 0 <= i0 <= i1 <= n0:
            raise IndexError(f"rows [{i0}, {i1}) outside an array of {n0} rows")
        shape = (i1 - i0,) + self.shape[1:]
        if out is None:
            out = np.empty(shape, dtype=self.dtype)
        elif out.shape != shape or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise ValueError("`out` must be a C-contiguous array of the range's shape and the array's dtype")
        if 0 in shape:
            return out
        grid = [range(i0 // self.inner[0], (i1 - 1) // self.inner[0] + 1)] + \
               [range(-(-s // c)) for s, c in zip(self.shape[1:], self.inner[1:])]
        tasks = list(itertools.product(*grid))
        if parallel and len(tasks) > 1:
            list(decode_pool().map(lambda t: self._load(out, i0, t), tasks))
        else:
            for t in tasks:
                self._load(out, i0, t)
        return out

    def __getitem__(self, sel):
        if sel is Ellipsis or (isinstance(sel, tuple) and sel == ()):
            return self.read()
        if isinstance(sel, slice) and sel.step in (None, 1):
            i0, i1, _ = sel.indices(self.shape[0])
            return self.read(i0, max(i0, i1))
        raise IndexError("only contiguous row ranges are read from a zarr array")


class Group:
    def __init__(self, store, path: str = ""):
        self.store, self.path = store, path
        key = f"{path}/.zgroup" if path else ".zgroup"
        if not store.exists(key):
            raise KeyError(f"no zarr v2 group at {path!r}")
        self.attrs = _read_attrs(store, path)

    def _child(self, name: str) -> str:
        return f"{self.path}/{name}" if self.path else name

    def keys(self) -> list[str]:
        return [n for n in self.store.subdirs(self.path)
                if self.store.exists(f"{self._child(n)}/.zarray") or self.store.exists(f"{self._child(n)}/.zgroup")]

    def __contains__(self, name: str) -> bool:
        p = self._child(name)
        return self.store.exists(f"{p}/.zarray") or self.store.exists(f"{p}/.zgroup")

    def __getitem__(self, name: str):
        p = self._child(name)
        if self.store.exists(f"{p}/.zarray"):
            return Array(self.store, p)
        if self.store.exists(f"{p}/.zgroup"):
            return Group(self.store, p)
        raise KeyError(f"{name!r} not in {self.path!r}")
