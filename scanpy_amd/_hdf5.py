"""A read-only HDF5 reader: what `.h5ad` files and 10x Genomics `.h5` matrices need, with RANGE reads along axis 0.

The reference reads HDF5 through h5py (`src/scanpy/readwrite.py:235-352` for 10x files, `read_h5ad` re-exported at
`:15-29`); h5py is not importable in this image's interpreter, and the out-of-core path wants rows
[i0, i1) of chunked, deflate-compressed 1-d datasets decoded in parallel into one buffer.  Restated from the HDF5 File
Format Specification (version 3.0) for the subset those files use, and pinned against files written by the HDF5
library itself: the reference's PyTables / h5py-written 10x fixtures and h5py-written AnnData-layout fixtures
(`tests/golden/make_h5_golden.py`), `tests/test_hdf5_cpu.py`.

    superblock        versions 0-3
    object headers    version 1 and version 2 ("OHDR" / "OCHK"), continuation blocks
    groups            symbol table (B-tree v1 + local heap + "SNOD") and compact link messages
    datasets          layout version 1-3: compact, contiguous, chunked (B-tree v1 chunk index);
                      layout version 4: single chunk, implicit and fixed-array chunk indexes
    filters           deflate (1), shuffle (2), fletcher32 (3), lzf (32000, h5py), zstd (32015)
    datatypes         integers, floats, fixed strings, variable-length strings (global heap), enums (h5py bool),
                      compounds of those, object references
    attributes        compact (header messages), versions 1-3

                      dense link storage (fractal heap, scanned in storage order: groups created with
                      `track_order=True` / `libver='latest'` holding more than 8 links)

Not read (a clear NotImplementedError says so): dense ATTRIBUTE storage (more than 8 attributes on a new-style object),
fractal heaps that lost objects, extensible-array and B-tree v2 chunk indexes,
szip / scale-offset / n-bit filters, region references, virtual and external datasets.
"""
from __future__ import annotations

import itertools
import os
import threading
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _Reader:
    """positioned reads on one file descriptor (`os.pread`: no seek state, safe from many threads)"""

    def __init__(self, path):
        self.fd = os.open(os.fspath(path), os.O_RDONLY)
        self.size = os.fstat(self.fd).st_size
        self.base = 0  # superblock base address: every file address is relative to it

    def at(self, addr: int, n: int) -> bytes:
        out = os.pread(self.fd, n, self.base + addr)
        if len(out) != n:
            raise ValueError(f"HDF5: short read of {n} bytes at {addr}")
        return out

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None


class _Inflate:
    """libz's `uncompress` through ctypes, into caller memory.  Not for single-thread speed (it is the same code as
    Python's zlib module) but because `zlib.decompress` allocates its result: eight threads each faulting in fresh
    4 MB buffers decode at 0.4 GB/s here, the same threads inflating into recycled buffers at 2 GB/s."""

    def __init__(self):
        self.fn = None
        try:
            import ctypes

            lib = ctypes.CDLL("libz.so.1")
            lib.uncompress.restype = ctypes.c_int
            lib.uncompress.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulong), ctypes.c_void_p, ctypes.c_ulong]
            self.fn, self._ulong, self._byref = lib.uncompress, ctypes.c_ulong, ctypes.byref
        except OSError:  # pragma: no cover - the image ships libz
            pass

    def into(self, src, dst: np.ndarray) -> int:
        """-> bytes produced (`dst`: C-contiguous uint8, large enough)"""
        if self.fn is None:  # pragma: no cover
            out = zlib.decompress(src)
            dst[:len(out)] = np.frombuffer(out, np.uint8)
            return len(out)
        n = self._ulong(dst.nbytes)
        src = bytes(src) if not isinstance(src, bytes) else src
        if self.fn(dst.ctypes.data, self._byref(n), src, len(src)) != 0:
            raise ValueError("HDF5: corrupt deflate chunk")
        return int(n.value)


_inflate = _Inflate()
_tls = threading.local()


def _scratch(name: str, nbytes: int) -> np.ndarray:
    """this thread's reusable uint8 buffer `name`, at least `nbytes` long"""
    buf = getattr(_tls, name, None)
    if buf is None or buf.nbytes < nbytes:
        buf = np.empty(nbytes, dtype=np.uint8)
        setattr(_tls, name, buf)
    return buf[:nbytes]


_native = None


def _lib():
    """libscanpy_amd.so's host codecs (`scamd_unshuffle`, `scamd_lzf_decompress`), or False when it is not built"""
    global _native
    if _native is None:
        try:
            from ._lib import load

            _native = load()
        except Exception:  # noqa: BLE001 - the readers work without it (numpy un-shuffle; lzf raises)
            _native = False
    return _native


def _unshuffle_into(src, dst: np.ndarray) -> None:
    """inverse of the HDF5 shuffle filter: `src` (bytes or uint8 array) = es byte planes of n bytes, `dst` = C-contiguous
    uint8 [n, es] view of the destination elements.  Native when the library is built (GIL released); else one strided
    numpy write per byte plane (3.4x faster than a single transposed assignment)"""
    n, es = dst.shape
    lib = _lib()
    if lib and dst.flags.c_contiguous:
        ptr = src.ctypes.data if isinstance(src, np.ndarray) else src
        if lib.scamd_unshuffle(ptr, dst.ctypes.data, n, es) != 0:
            raise ValueError("HDF5: un-shuffle failed")
        return
    planes = (src if isinstance(src, np.ndarray) else np.frombuffer(src, np.uint8))[:n * es].reshape(es, n)
    for j in range(es):
        dst[:, j] = planes[j]


def _uint(buf, off: int, n: int) -> int:
    return int.from_bytes(buf[off:off + n], "little")


# ---------------------------------------------------------------------------------------------------------------------
# datatypes


class _Type:
    """a parsed datatype message: `dtype` (numpy, for fixed-size elements), `kind` in {'fixed', 'vlen_str', 'vlen'}"""

    def __init__(self, dtype=None, kind="fixed", size=0, base=None, enum=None, utf8=False, strpad=0):
        self.dtype, self.kind, self.size, self.base, self.enum, self.utf8, self.strpad = \
            dtype, kind, size, base, enum, utf8, strpad


def _parse_type(buf, off: int) -> tuple[_Type, int]:
    """-> (type, offset just past the message)"""
    cv = buf[off]
    cls, ver = cv & 0x0F, cv >> 4
    bits = _uint(buf, off + 1, 3)
    size = _uint(buf, off + 4, 4)
    p = off + 8
    if cls == 0:  # fixed-point
        order = ">" if bits & 1 else "<"
        return _Type(np.dtype(f"{order}{'i' if bits & 8 else 'u'}{size}"), size=size), p + 4
    if cls == 1:  # floating point
        order = ">" if bits & 1 else "<"
        return _Type(np.dtype(f"{order}f{size}"), size=size), p + 12
    if cls == 3:  # fixed-length string
        return _Type(np.dtype(f"S{size}"), size=size, utf8=bool((bits >> 4) & 0xF), strpad=bits & 0xF), p
    if cls == 4:  # bitfield -> unsigned
        return _Type(np.dtype(f"<u{size}"), size=size), p + 4
    if cls == 6:  # compound
        n = bits & 0xFFFF
        names, types, offsets = [], [], []
        for _ in range(n):
            e = buf.index(b"\0", p)
            name = bytes(buf[p:e]).decode("utf-8")
            if ver < 3:
                p += ((e - p) // 8 + 1) * 8
                moff = _uint(buf, p, 4)
                p += 4
                if ver == 1:
                    p += 1 + 3 + 4 + 4 + 16
            else:
                p = e + 1
                nb = 1 if size < 256 else 2 if size < 65536 else 3 if size < (1 << 24) else 4
                moff = _uint(buf, p, nb)
                p += nb
            mt, p = _parse_type(buf, p)
            if mt.kind != "fixed":
                raise NotImplementedError("HDF5: compound members of variable length are not read here")
            names.append(name)
            types.append(np.dtype(bool) if mt.enum == "bool" and mt.size == 1 else mt.dtype)
            offsets.append(moff)
        return _Type(np.dtype({"names": names, "formats": types, "offsets": offsets, "itemsize": size}), size=size), p
    if cls == 8:  # enumeration (h5py stores numpy bool as ENUM {FALSE=0, TRUE=1} of int8)
        n = bits & 0xFFFF
        bt, p = _parse_type(buf, p)
        names = []
        for _ in range(n):
            e = buf.index(b"\0", p)
            names.append(bytes(buf[p:e]).decode("utf-8"))
            p = p + ((e - p) // 8 + 1) * 8 if ver < 3 else e + 1
        values = np.frombuffer(bytes(buf[p:p + n * bt.size]), dtype=bt.dtype)
        p += n * bt.size
        mapping = dict(zip(names, values.tolist()))
        is_bool = {k.upper() for k in mapping} == {"FALSE", "TRUE"}
        return _Type(bt.dtype, size=size, enum=mapping if not is_bool else "bool"), p
    if cls == 9:  # variable length
        bt, p = _parse_type(buf, p)
        if bits & 0xF == 1:
            return _Type(np.dtype(object), kind="vlen_str", size=size, utf8=bool((bits >> 8) & 0xF)), p
        return _Type(np.dtype(object), kind="vlen", size=size, base=bt), p
    if cls == 10:  # array
        nd = buf[p]
        p += 4 if ver < 3 else 1
        dims = [_uint(buf, p + 4 * i, 4) for i in range(nd)]
        p += 4 * nd + (4 * nd if ver < 3 else 0)
        bt, p = _parse_type(buf, p)
        if bt.kind != "fixed":
            raise NotImplementedError("HDF5: arrays of variable-length elements are not read here")
        return _Type(np.dtype((bt.dtype, tuple(dims))), size=size), p
    if cls == 7 and bits & 0xF == 0:  # object reference: the address of an object header
        return _Type(np.dtype(f"<u{size}"), size=size, enum="objref"), p
    raise NotImplementedError(f"HDF5: datatype class {cls} is not read here")


def _parse_space(buf, off: int, L: int) -> tuple[tuple[int, ...] | None, int]:
    """-> (shape, or None for a null dataspace; end offset)"""
    ver, rank, flags = buf[off], buf[off + 1], buf[off + 2]
    if ver == 1:
        p = off + 8
    elif ver == 2:
        if buf[off + 3] == 2:
            return None, off + 4
        p = off + 4
    else:
        raise NotImplementedError(f"HDF5: dataspace message version {ver}")
    shape = tuple(_uint(buf, p + L * i, L) for i in range(rank))
    p += L * rank * (2 if flags & 1 else 1)
    return shape, p


# ---------------------------------------------------------------------------------------------------------------------
# file, objects


class File:
    def __init__(self, path):
        self.path = os.fspath(path)
        self.r = _Reader(path)
        self._gheap: dict[int, dict[int, bytes]] = {}
        self._lock = threading.Lock()
        addr = 0
        while True:  # the superblock sits at 0, 512, 1024, 2048, ...
            if addr + 8 > self.r.size:
                self.r.close()
                raise ValueError(f"{self.path}: not an HDF5 file")
            if os.pread(self.r.fd, 8, addr) == _SIG:
                break
            addr = 512 if addr == 0 else addr * 2
        sb = os.pread(self.r.fd, 128, addr)
        ver = sb[8]
        if ver in (0, 1):
            self.O, self.L = sb[13], sb[14]
            p = 24 + (4 if ver == 1 else 0)
            base = _uint(sb, p, self.O)
            p += 4 * self.O  # base, free-space info, end of file, driver info
            root_header = _uint(sb, p + self.O, self.O)  # symbol table entry: name offset, header address
        elif ver in (2, 3):
            self.O, self.L = sb[9], sb[10]
            base = _uint(sb, 12, self.O)
            root_header = _uint(sb, 12 + 3 * self.O, self.O)
        else:
            self.r.close()
            raise NotImplementedError(f"{self.path}: HDF5 superblock version {ver}")
        self.r.base = base  # (a file with a user block stores base = the superblock's offset)
        self.root = Group(self, root_header, "/")

    # convenience: the file acts as its root group
    def __getitem__(self, name):
        return self.root[name]

    def __contains__(self, name):
        return name in self.root

    def keys(self):
        return self.root.keys()

    @property
    def attrs(self):
        return self.root.attrs

    def close(self):
        self.r.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def deref(self, ref: "Reference"):
        """the dataset or group an object reference points at"""
        kinds = {m[0] for m in _messages(self, ref.addr)}
        name = f"<object at {ref.addr}>"
        return Dataset(self, ref.addr, name) if 0x08 in kinds else Group(self, ref.addr, name)

    # -- global heap (variable-length data)
    def gheap_object(self, addr: int, index: int) -> bytes:
        with self._lock:
            coll = self._gheap.get(addr)
        if coll is None:
            head = self.r.at(addr, 8 + self.L)
            if head[:4] != b"GCOL":
                raise ValueError("HDF5: bad global heap collection")
            size = _uint(head, 8, self.L)
            buf = self.r.at(addr, size)
            coll, p = {}, 8 + self.L
            while p + 8 + self.L <= size:
                idx = _uint(buf, p, 2)
                n = _uint(buf, p + 8, self.L)
                if idx == 0:
                    break
                coll[idx] = buf[p + 8 + self.L:p + 8 + self.L + n]
                p += 8 + self.L + (n + 7) // 8 * 8
            with self._lock:
                self._gheap[addr] = coll
        return coll.get(index, b"")

    def vlen_strings(self, raw: bytes, count: int, utf8: bool = True) -> np.ndarray:
        """`count` variable-length string references (length u32, collection address, object index u32) -> str"""
        out = np.empty(count, dtype=object)
        rec = np.frombuffer(raw, dtype=np.dtype([("n", "<u4"), ("addr", f"<u{self.O}"), ("idx", "<u4")]), count=count)
        undef = _UNDEF & ((1 << (8 * self.O)) - 1)
        colls = {}
        for a in np.unique(rec["addr"]).tolist():
            if a not in (0, undef):
                self.gheap_object(a, 1)  # loads and caches the collection
                colls[a] = self._gheap[a]
        empty: dict = {}
        for i, (n, a, idx) in enumerate(zip(rec["n"].tolist(), rec["addr"].tolist(), rec["idx"].tolist())):
            out[i] = colls.get(a, empty).get(idx, b"")[:n].decode("utf-8", "replace") if n else ""
        return out


def _messages(f: File, addr: int):
    """yield (type, flags, payload bytes) of every message of the object header at `addr`"""
    r, O, L = f.r, f.O, f.L
    head = r.at(addr, 16)
    if head[:4] == b"OHDR":
        flags = head[5]
        p = 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
        nb = 1 << (flags & 3)
        head = r.at(addr, p + nb)
        size0 = _uint(head, p, nb)
        blocks = [(addr + p + nb, size0)]
        track = bool(flags & 0x04)
        while blocks:
            baddr, bsize = blocks.pop(0)
            buf = r.at(baddr, bsize)
            q = 0
            while q + 4 <= bsize:
                mtype, msize, mflags = buf[q], _uint(buf, q + 1, 2), buf[q + 3]
                q += 4 + (2 if track else 0)
                body = buf[q:q + msize]
                q += msize
                if mtype == 0x10:
                    caddr, clen = _uint(body, 0, O), _uint(body, O, L)
                    blocks.append((caddr + 4, clen - 8))  # skip "OCHK", drop the checksum
                elif mtype != 0:
                    yield mtype, mflags, body
        return
    if head[0] != 1:
        raise ValueError(f"HDF5: no object header at {addr}")
    nmsg, hsize = _uint(head, 2, 2), _uint(head, 8, 4)
    blocks = [(addr + 16, hsize)]
    seen = 0
    while blocks and seen < nmsg:
        baddr, bsize = blocks.pop(0)
        buf = r.at(baddr, bsize)
        q = 0
        while q + 8 <= bsize and seen < nmsg:
            mtype, msize, mflags = _uint(buf, q, 2), _uint(buf, q + 2, 2), buf[q + 4]
            body = buf[q + 8:q + 8 + msize]
            q += 8 + msize
            seen += 1
            if mtype == 0x10:
                blocks.append((_uint(body, 0, O), _uint(body, O, L)))
            elif mtype != 0:
                yield mtype, mflags, body


def _parse_link(buf, p: int, O: int):
    """one link message at `p` -> (name, object header address or None for a soft / external link, end offset)"""
    if buf[p] != 1:
        raise ValueError("HDF5: bad link message")
    flags = buf[p + 1]
    p += 2
    ltype = 0
    if flags & 0x08:
        ltype = buf[p]
        p += 1
    if flags & 0x04:
        p += 8
    if flags & 0x10:
        p += 1
    nb = 1 << (flags & 3)
    n = _uint(buf, p, nb)
    p += nb
    name = bytes(buf[p:p + n]).decode("utf-8")
    p += n
    if ltype == 0:
        return name, _uint(buf, p, O), p + O
    if ltype == 1:  # soft link: length + path
        return name, None, p + 2 + _uint(buf, p, 2)
    return name, None, p + 2 + _uint(buf, p + 1, 2) + 1 if ltype >= 64 else p


def _fractal_heap_objects(f: File, addr: int, where: str):
    """The managed objects of a fractal heap, in storage order (dense link storage of groups created with
    `track_order=True` / `libver='latest'`: every object is one link message).

    The name-index B-tree (version 2) is not consulted: direct blocks are filled front to back, so the objects of a
    heap that never lost one are its blocks' contents in order.  The scan must find exactly `number of managed
    objects` link messages; a heap that lost links (stale bytes or holes in its blocks) is refused, not guessed at."""
    O, L = f.O, f.L
    h = f.r.at(addr, 22 + 12 * L + 3 * O + 8)
    if h[:4] != b"FRHP":
        raise ValueError(f"HDF5: bad fractal heap header in {where!r}")
    filt_len = _uint(h, 7, 2)
    flags = h[9]
    p = 10 + 4 + L + O  # max managed object size, next huge id, huge-object B-tree
    p += L + O  # free space, free-space manager
    p += 2 * L  # managed space, allocated managed space
    p += L      # allocation iterator offset
    n_managed = _uint(h, p, L)
    p += L + 4 * L  # number of managed objects; huge size / count, tiny size / count
    width = _uint(h, p, 2)
    start = _uint(h, p + 2, L)
    max_direct = _uint(h, p + 2 + L, L)
    max_bits = _uint(h, p + 2 + 2 * L, 2)
    root = _uint(h, p + 6 + 2 * L, O)
    cur_rows = _uint(h, p + 6 + 2 * L + O, 2)
    if filt_len:
        raise NotImplementedError(f"HDF5: {where!r}: filtered fractal heaps are not read here")
    off_bytes = (max_bits + 7) // 8
    dhead = 5 + O + off_bytes + (4 if flags & 0x02 else 0)
    undef = _UNDEF & ((1 << (8 * O)) - 1)

    def direct_blocks():
        if cur_rows == 0:
            if root != undef:
                yield root, start
            return
        stack = [(root, cur_rows)]
        while stack:
            iaddr, nrows = stack.pop(0)
            n_direct_rows = (max_direct // start).bit_length() + 1  # rows whose blocks are still direct blocks
            ih = 5 + O + off_bytes
            ents = f.r.at(iaddr, ih + nrows * width * O)
            if ents[:4] != b"FHIB":
                raise ValueError(f"HDF5: bad fractal heap indirect block in {where!r}")
            q = ih
            for r in range(nrows):
                size = start if r < 2 else start << (r - 1)
                for _ in range(width):
                    child = _uint(ents, q, O)
                    q += O
                    if child == undef:
                        continue
                    if r < n_direct_rows:
                        yield child, size
                    else:  # a nested indirect block covering `size` bytes of heap space
                        stack.append((child, (size // (start * width)).bit_length()))  # log2(size / (start * width)) + 1

    found = []
    for baddr, bsize in direct_blocks():
        blk = f.r.at(baddr, bsize)
        if blk[:4] != b"FHDB":
            raise ValueError(f"HDF5: bad fractal heap direct block in {where!r}")
        q = dhead
        while q < bsize and blk[q] == 1:  # link messages start with their version, 1
            try:
                _, _, end = _parse_link(blk, q, O)
            except (ValueError, IndexError, UnicodeDecodeError):
                break
            if end > bsize:
                break
            found.append(blk[q:end])
            q = end
    if len(found) != n_managed:
        # a deleted link leaves its bytes behind (one candidate too many) or a hole that ends the scan of its block
        # (candidates missing): only the name-index B-tree knows which objects are live
        raise NotImplementedError(f"HDF5: {where!r}: the sequential scan of its fractal heap found {len(found)} link "
                                  f"messages where the heap holds {n_managed} (links were deleted from this group; "
                                  "the version-2 B-tree index is not read here)")
    return found


def _parse_attribute(f: File, body: bytes):
    ver = body[0]
    nsz, tsz, ssz = _uint(body, 2, 2), _uint(body, 4, 2), _uint(body, 6, 2)
    p = 8 + (1 if ver == 3 else 0)
    pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
    name = bytes(body[p:p + nsz]).split(b"\0", 1)[0].decode("utf-8")
    p += pad(nsz)
    if ver in (2, 3) and body[1] & 0x03:
        raise NotImplementedError(f"HDF5: attribute {name!r} uses a shared datatype / dataspace")
    t, _ = _parse_type(body, p)
    p += pad(tsz)
    shape, _ = _parse_space(body, p, f.L)
    p += pad(ssz)
    if shape is None:
        return name, None
    count = int(np.prod(shape)) if shape else 1
    return name, _decode_elements(f, t, body[p:], count, shape)


def _decode_elements(f: File, t: _Type, raw, count: int, shape):
    if t.kind == "vlen_str":
        arr = f.vlen_strings(raw, count)
    elif t.kind == "vlen":
        raise NotImplementedError("HDF5: variable-length sequences are not read here")
    else:
        arr = np.frombuffer(bytes(raw[:count * t.size]), dtype=t.dtype, count=count)
        arr = _finish_fixed(arr, t)
    arr = arr.reshape(shape)
    return arr[()] if shape == () else arr


class Reference:
    """an HDF5 object reference (`h5py.Reference`): resolve with `File.deref`"""

    def __init__(self, addr: int):
        self.addr = int(addr)

    def __repr__(self) -> str:
        return f"<HDF5 object reference to {self.addr}>"


def _finish_fixed(arr: np.ndarray, t: _Type) -> np.ndarray:
    if t.enum == "bool":
        return arr.astype(bool)
    if t.enum == "objref":
        out = np.empty(arr.shape, dtype=object)
        out.reshape(-1)[:] = [Reference(a) for a in arr.reshape(-1).tolist()]
        return out
    if t.dtype.kind == "S" and t.utf8:
        return np.array([s.decode("utf-8", "replace") for s in arr.tolist()], dtype=object).reshape(arr.shape)
    return arr


class _Node:
    def __init__(self, f: File, addr: int, name: str):
        self.file, self.addr, self.name = f, addr, name
        self._msgs = None
        self._attrs = None

    def _messages(self):
        if self._msgs is None:
            self._msgs = list(_messages(self.file, self.addr))
        return self._msgs

    @property
    def attrs(self) -> dict:
        if self._attrs is None:
            out = {}
            for mtype, _, body in self._messages():
                if mtype == 0x0C:
                    try:
                        k, v = _parse_attribute(self.file, body)
                    except NotImplementedError:  # e.g. a region reference: the attribute is skipped, not the object
                        continue
                    out[k] = v
                elif mtype == 0x15:  # attribute info: dense storage if the fractal heap address is defined
                    flags = body[1]
                    p = 2 + (2 if flags & 1 else 0)
                    if _uint(body, p, self.file.O) != _UNDEF & ((1 << (8 * self.file.O)) - 1):
                        raise NotImplementedError(f"HDF5: {self.name!r} keeps its attributes in dense storage "
                                                  "(fractal heap), which is not read here")
            self._attrs = out
        return self._attrs


class Group(_Node):
    def __init__(self, f: File, addr: int, name: str):
        super().__init__(f, addr, name)
        self._links = None

    def _load_links(self) -> dict[str, int]:
        if self._links is not None:
            return self._links
        f, O, L = self.file, self.file.O, self.file.L
        links: dict[str, int] = {}
        for mtype, _, body in self._messages():
            if mtype == 0x11:  # symbol table: B-tree v1 of "SNOD" leaves + local heap of names
                btree, heap = _uint(body, 0, O), _uint(body, O, O)
                hh = f.r.at(heap, 8 + 2 * L + O)
                if hh[:4] != b"HEAP":
                    raise ValueError("HDF5: bad local heap")
                hdata = f.r.at(_uint(hh, 8 + 2 * L, O), _uint(hh, 8, L))
                stack = [btree]
                while stack:
                    a = stack.pop()
                    nh = f.r.at(a, 8)
                    if nh[:4] == b"TREE":
                        n = _uint(nh, 6, 2)
                        body2 = f.r.at(a + 8 + 2 * O, n * (O + L) + L)
                        # keys (L bytes) and children (O bytes) alternate: key0 child0 key1 ... keyN
                        q = L
                        kids = []
                        for _ in range(n):
                            kids.append(_uint(body2, q, O))
                            q += O + L
                        stack.extend(reversed(kids))
                    elif nh[:4] == b"SNOD":
                        n = _uint(nh, 6, 2)
                        ent = f.r.at(a + 8, n * (2 * O + 24))
                        for i in range(n):
                            e = i * (2 * O + 24)
                            noff, haddr = _uint(ent, e, O), _uint(ent, e + O, O)
                            end = hdata.index(b"\0", noff)
                            links[hdata[noff:end].decode("utf-8")] = haddr
                    else:
                        raise ValueError("HDF5: bad group B-tree node")
            elif mtype == 0x06:  # link message (compact new-style group)
                lname, addr, _ = _parse_link(body, 0, O)
                if addr is not None:
                    links[lname] = addr
            elif mtype == 0x02:  # link info: dense storage if the fractal heap address is defined
                flags = body[1]
                p = 2 + (8 if flags & 1 else 0)
                heap = _uint(body, p, O)
                if heap != _UNDEF & ((1 << (8 * O)) - 1):
                    for obj in _fractal_heap_objects(f, heap, self.name):
                        lname, addr, _ = _parse_link(obj, 0, O)
                        if addr is not None:
                            links[lname] = addr
        self._links = links
        return links

    @property
    def path(self) -> str:
        return self.name

    def keys(self) -> list[str]:
        return sorted(self._load_links())

    def __contains__(self, name: str) -> bool:
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name: str):
        node = self
        for part in [p for p in name.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(name)
            links = node._load_links()
            if part not in links:
                raise KeyError(f"{part!r} not in {node.name!r}")
            addr = links[part]
            child = f"{node.name.rstrip('/')}/{part}"
            kinds = {m[0] for m in _messages(self.file, addr)}
            node = Dataset(self.file, addr, child) if 0x08 in kinds else Group(self.file, addr, child)
        return node


class Dataset(_Node):
    def __init__(self, f: File, addr: int, name: str):
        super().__init__(f, addr, name)
        L = f.L
        self.filters: list[tuple[int, tuple[int, ...]]] = []
        self.fill = None
        self.type = self.shape = None
        self.layout = None
        for mtype, _, body in self._messages():
            if mtype == 0x03:
                self.type, _ = _parse_type(body, 0)
            elif mtype == 0x01:
                self.shape, _ = _parse_space(body, 0, L)
            elif mtype == 0x0B:
                self._parse_filters(body)
            elif mtype == 0x08:
                self._parse_layout(body)
        if self.type is None or self.layout is None:
            raise ValueError(f"HDF5: {name!r} lacks a datatype or a layout")
        self.dtype = np.dtype(bool) if self.type.enum == "bool" else self.type.dtype
        if self.shape is None:
            self.shape = (0,)
        self.ndim = len(self.shape)
        self._index = None
        self._scratch = threading.local()
        self._lock = threading.Lock()

    # -- the small protocol shared with `_zarr3.Array` (readwrite.read_elem, _backed.BackedCsr)
    @property
    def path(self) -> str:
        return self.name

    @property
    def is_string(self) -> bool:
        return self.type.kind == "vlen_str" or (self.type.kind == "fixed" and self.type.dtype.kind == "S")

    @property
    def inner(self) -> tuple[int, ...]:
        """chunk shape (the whole dataset for compact / contiguous layouts)"""
        if self.layout[0] in ("compact", "contiguous"):
            return tuple(max(1, s) for s in self.shape)
        return self._chunk_index()[0]

    # -- header pieces
    def _parse_filters(self, body):
        ver, n = body[0], body[1]
        p = 8 if ver == 1 else 2
        for _ in range(n):
            fid = _uint(body, p, 2)
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = _uint(body, p, 2)
                p += 2
            p += 2  # flags
            nvals = _uint(body, p, 2)
            p += 2
            p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            vals = tuple(_uint(body, p + 4 * i, 4) for i in range(nvals))
            p += 4 * nvals + (4 if ver == 1 and nvals % 2 else 0)
            if fid not in (1, 2, 3, 32000, 32015):
                names = {4: "szip", 5: "nbit", 6: "scaleoffset", 32001: "blosc", 32004: "lz4"}
                raise NotImplementedError(f"HDF5: {self.name!r} uses filter {names.get(fid, fid)}, which is not read "
                                          "here (deflate, shuffle, fletcher32, lzf and zstd are)")
            self.filters.append((fid, vals))

    def _parse_layout(self, body):
        O, L = self.file.O, self.file.L
        ver = body[0]
        if ver in (1, 2):
            nd, cls = body[1], body[2]
            p = 8
            addr = None
            if cls != 0:
                addr = _uint(body, p, O)
                p += O
            dims = [_uint(body, p + 4 * i, 4) for i in range(nd)]
            p += 4 * nd
            if cls == 2:
                self.layout = ("chunked", addr, tuple(dims[:-1]) if len(dims) > 1 else tuple(dims))
                # (versions 1-2 store the element size as the last "dimension")
            elif cls == 1:
                self.layout = ("contiguous", addr, None)
            else:
                n = _uint(body, p, 4)
                self.layout = ("compact", bytes(body[p + 4:p + 4 + n]), None)
            return
        cls = body[1]
        if cls == 0:
            n = _uint(body, 2, 2)
            self.layout = ("compact", bytes(body[4:4 + n]), None)
        elif cls == 1:
            self.layout = ("contiguous", _uint(body, 2, O), _uint(body, 2 + O, L))
        elif cls == 2 and ver == 3:
            nd = body[2]
            addr = _uint(body, 3, O)
            dims = tuple(_uint(body, 3 + O + 4 * i, 4) for i in range(nd))
            self.layout = ("chunked", addr, dims[:-1])
        elif cls == 2 and ver == 4:
            flags, nd, enc = body[2], body[3], body[4]
            dims = tuple(_uint(body, 5 + enc * i, enc) for i in range(nd))
            p = 5 + enc * nd
            itype = body[p]
            p += 1
            if itype == 1:  # single chunk
                size = mask = None
                if flags & 0x02:
                    size, mask = _uint(body, p, L), _uint(body, p + L, 4)
                    p += L + 4
                self.layout = ("single", _uint(body, p, O), (dims[:-1], size, mask))
            elif itype == 2:  # implicit: chunks back to back, unfiltered
                self.layout = ("implicit", _uint(body, p, O), dims[:-1])
            elif itype == 3:  # fixed array
                self.layout = ("farray", _uint(body, p + 1, O), dims[:-1])
            else:
                kinds = {4: "extensible array", 5: "version-2 B-tree"}
                raise NotImplementedError(f"HDF5: {self.name!r} indexes its chunks with a {kinds.get(itype, itype)} "
                                          "(a resizable dataset written with libver='latest'), not read here")
        else:
            raise NotImplementedError(f"HDF5: {self.name!r}: data layout version {ver} class {cls}")

    # -- chunk index
    def _chunk_index(self):
        """-> (chunk shape, {chunk origin tuple: (address, stored size, filter mask)})"""
        with self._lock:
            if self._index is not None:
                return self._index
        f, O, L = self.file, self.file.O, self.file.L
        kind, addr, extra = self.layout
        table: dict[tuple[int, ...], tuple[int, int, int]] = {}
        undef = _UNDEF & ((1 << (8 * O)) - 1)
        if kind == "chunked":
            cshape = extra
            nd = len(cshape)
            stack = [addr] if addr != undef else []
            while stack:
                a = stack.pop()
                head = f.r.at(a, 8 + 2 * O)
                if head[:4] != b"TREE" or head[4] != 1:
                    raise ValueError("HDF5: bad chunk B-tree node")
                level, n = head[5], _uint(head, 6, 2)
                ksz = 8 + 8 * (nd + 1)
                body = f.r.at(a + 8 + 2 * O, n * (ksz + O) + ksz)
                for i in range(n):
                    q = i * (ksz + O)
                    child = _uint(body, q + ksz, O)
                    if level > 0:
                        stack.append(child)
                    else:
                        origin = tuple(_uint(body, q + 8 + 8 * d, 8) for d in range(nd))
                        table[origin] = (child, _uint(body, q, 4), _uint(body, q + 4, 4))
        elif kind == "single":
            cshape, size, mask = extra
            if addr != undef:
                raw = int(np.prod(cshape)) * self.type.size
                table[(0,) * len(cshape)] = (addr, raw if size is None else size, mask or 0)
        elif kind == "implicit":
            cshape = extra
            raw = int(np.prod(cshape)) * self.type.size
            grid = [-(-s // c) for s, c in zip(self.shape, cshape)]
            for lin, idx in enumerate(np.ndindex(*grid)):
                table[tuple(i * c for i, c in zip(idx, cshape))] = (addr + lin * raw, raw, 0)
        elif kind == "farray":
            cshape = extra
            raw = int(np.prod(cshape)) * self.type.size
            hd = f.r.at(addr, 12 + L + O)
            if hd[:4] != b"FAHD":
                raise ValueError("HDF5: bad fixed array header")
            client, esz, bits = hd[5], hd[6], hd[7]
            nent = _uint(hd, 8, L)
            dblk = _uint(hd, 8 + L, O)
            grid = [-(-s // c) for s, c in zip(self.shape, cshape)]
            if dblk != undef:
                # data block "FADB": signature, version, client, header address; paged blocks (more entries than
                # one page holds) add a page-initialised bitmap and a checksum, then pages that each end in a checksum
                page = 1 << bits
                paged = nent > page
                npages = -(-nent // page) if paged else 1
                off = dblk + 4 + 1 + 1 + O
                bitmap = None
                if paged:
                    nbm = (npages + 7) // 8
                    bitmap = f.r.at(off, nbm)
                    off += nbm + 4
                entries = []
                left = nent
                for pg in range(npages):
                    m = min(page, left) if paged else nent
                    live = bitmap is None or bool(bitmap[pg // 8] & (0x80 >> (pg % 8)))
                    buf = f.r.at(off, m * esz) if live else b""
                    for i in range(m):
                        e = i * esz
                        if not live:
                            entries.append((undef, 0, 0))
                        elif client == 1:  # filtered chunks: address, stored size (esz - O - 4 bytes), filter mask
                            nb = esz - O - 4
                            entries.append((_uint(buf, e, O), _uint(buf, e + O, nb), _uint(buf, e + O + nb, 4)))
                        else:
                            entries.append((_uint(buf, e, O), raw, 0))
                    off += m * esz + (4 if paged else 0)
                    left -= m
                for lin, idx in enumerate(np.ndindex(*grid)):
                    if lin < len(entries) and entries[lin][0] != undef:
                        table[tuple(i * c for i, c in zip(idx, cshape))] = entries[lin]
        else:
            raise AssertionError(kind)
        with self._lock:
            self._index = (tuple(cshape), table)
        return self._index

    def _fill_value(self):
        return np.zeros((), dtype=self.type.dtype)

    def _decode_chunk(self, raw, mask: int, dst: np.ndarray) -> None:
        """undo the filter pipeline of one stored chunk INTO `dst` (C-contiguous uint8 of the chunk's byte size).
        Intermediate results live in this thread's scratch buffers: nothing is allocated per chunk."""
        nbytes = dst.nbytes
        live = [k for k in range(len(self.filters)) if not mask & (1 << k)]
        cur = raw  # bytes, memoryview or uint8 array
        for pos in range(len(live) - 1, -1, -1):
            k = live[pos]
            fid, vals = self.filters[k]
            last = pos == 0
            # bytes this stage must produce: the chunk plus a checksum for every fletcher32 stage still to be undone
            want = nbytes + 4 * sum(1 for j in live[:pos] if self.filters[j][0] == 3)
            if fid == 3:
                cur = memoryview(cur)[:len(cur) - 4] if not isinstance(cur, np.ndarray) else cur[:-4]
                continue
            if fid == 2:
                es = vals[0] if vals else self.type.size
                n = want // es
                if es <= 1 or n <= 1:
                    continue
                target = dst if last else _scratch("shuffle", want)
                src = cur if isinstance(cur, (bytes, np.ndarray)) else bytes(cur)
                _unshuffle_into(src, target[:n * es].reshape(n, es))
                if n * es < want:  # leftover bytes of a size that is no multiple of the element are stored as they are
                    tail = src if isinstance(src, np.ndarray) else np.frombuffer(src, np.uint8)
                    target[n * es:] = tail[n * es:want]
                cur = target
                continue
            target = dst if last else _scratch("inflate", want)
            src = cur.tobytes() if isinstance(cur, np.ndarray) else cur
            if fid == 1:
                got = _inflate.into(src, target)
            elif fid == 32015:
                from ._zarr3 import _zstd

                _zstd.decompress_into(bytes(src), target)
                got = want
            else:  # 32000: h5py's lzf filter
                lib = _lib()
                if not lib:
                    raise RuntimeError("HDF5: lzf chunks are decoded by libscanpy_amd.so, which is not built "
                                       "(python -m scanpy_amd._build)")
                src = bytes(src)
                got = lib.scamd_lzf_decompress(src, len(src), target.ctypes.data, target.nbytes)
                if got < 0:
                    raise ValueError(f"HDF5: corrupt lzf chunk in {self.name!r}")
            if got != want:
                raise ValueError(f"HDF5: chunk of {self.name!r} decodes to {got} bytes, expected {want}")
            cur = target
        if cur is not dst:
            have = len(cur) if not isinstance(cur, np.ndarray) else cur.nbytes
            if have < nbytes:
                raise ValueError(f"HDF5: chunk of {self.name!r} holds {have} bytes, expected {nbytes}")
            dst[:] = np.frombuffer(cur, np.uint8, count=nbytes) if not isinstance(cur, np.ndarray) else cur[:nbytes]

    # -- reading
    def _raw_rows(self, i0: int, i1: int, out: np.ndarray | None = None, parallel: bool = True) -> np.ndarray:
        """rows [i0, i1) as an array of the STORAGE dtype (fixed-size elements; vlen = the heap references), decoded
        into `out` when it is given (same dtype and shape, C-contiguous)"""
        t = self.type
        esz = t.size
        store = np.dtype((np.void, esz)) if t.kind != "fixed" else t.dtype
        shape = (i1 - i0,) + tuple(self.shape[1:])
        if out is None or out.dtype != store or out.shape != shape or not out.flags.c_contiguous:
            out = np.empty(shape, dtype=store)
        if out.size == 0:
            return out
        kind = self.layout[0]
        undef = _UNDEF & ((1 << (8 * self.file.O)) - 1)
        row = int(np.prod(self.shape[1:])) * esz if self.ndim > 1 else esz
        flat = out.reshape(-1).view(np.uint8)
        if kind == "compact":
            flat[:] = np.frombuffer(self.layout[1], np.uint8)[i0 * row:i1 * row]
            return out
        if kind == "contiguous":
            addr = self.layout[1]
            if addr == undef:
                flat[:] = 0
            else:
                flat[:] = np.frombuffer(self.file.r.at(addr + i0 * row, (i1 - i0) * row), np.uint8)
            return out
        cshape, table = self._chunk_index()
        n_elem = int(np.prod(cshape))
        cbytes = n_elem * esz
        c0 = cshape[0]
        grid = [range(i0 // c0, (i1 - 1) // c0 + 1)] + [range(-(-s // c)) for s, c in zip(self.shape[1:], cshape[1:])]
        tasks = [tuple(g * c for g, c in zip(idx, cshape)) for idx in itertools.product(*grid)]
        one_d = self.ndim == 1

        def load(origin):
            a0, a1 = max(origin[0], i0), min(origin[0] + c0, i1)
            if a1 <= a0:
                return
            dst = (slice(a0 - i0, a1 - i0),) + tuple(slice(o, min(o + c, s)) for o, c, s in
                                                      zip(origin[1:], cshape[1:], self.shape[1:]))
            ent = table.get(origin)
            if ent is None:
                out[dst] = np.zeros((), dtype=store)
                return
            addr, size, mask = ent
            raw = self.file.r.at(addr, size)
            whole = one_d and a0 == origin[0] and a1 == origin[0] + c0
            if whole:  # the chunk lies inside the range: decode straight into its place
                self._decode_chunk(raw, mask if self.filters else ~0, out[dst].view(np.uint8))
                return
            tmp = _scratch("chunk", cbytes)
            self._decode_chunk(raw, mask if self.filters else ~0, tmp)
            chunk = tmp.view(store).reshape(cshape)
            src = (slice(a0 - origin[0], a1 - origin[0]),) + tuple(slice(0, d.stop - d.start) for d in dst[1:])
            out[dst] = chunk[src]

        if len(tasks) > 1 and parallel:
            from ._zarr3 import decode_pool

            list(decode_pool().map(load, tasks))
        else:
            for t0 in tasks:
                load(t0)
        return out

    def read(self, i0: int = 0, i1: int | None = None, *, out: np.ndarray | None = None,
             parallel: bool = True) -> np.ndarray:
        """rows [i0, i1) along axis 0 (everything for a scalar dataset)"""
        t = self.type
        if self.ndim == 0:
            kind = self.layout[0]
            if kind == "compact":
                raw = self.layout[1]
            elif kind == "contiguous":
                raw = self.file.r.at(self.layout[1], t.size)
            else:
                raise NotImplementedError(f"HDF5: scalar dataset {self.name!r} with a chunked layout")
            return np.asarray(_decode_elements(self.file, t, raw, 1, ()))
        n0 = self.shape[0]
        i1 = n0 if i1 is None else i1
        if not 0 <= i0 <= i1 <= n0:
            raise IndexError(f"rows [{i0}, {i1}) outside a dataset of {n0} rows")
        direct = out is not None and t.kind == "fixed" and t.enum is None and not (t.dtype.kind == "S" and t.utf8)
        raw = self._raw_rows(i0, i1, out if direct else None, parallel)
        if direct and raw is out:
            return out
        if t.kind == "vlen_str":
            res = self.file.vlen_strings(raw.tobytes(), raw.size).reshape(raw.shape)
        elif t.kind == "vlen":
            raise NotImplementedError("HDF5: variable-length sequences are not read here")
        else:
            res = _finish_fixed(raw, t)
        if out is not None:
            out[...] = res
            return out
        return res

    def __getitem__(self, sel):
        if sel is Ellipsis or (isinstance(sel, tuple) and sel == ()):
            r = self.read()
            return r[()] if self.ndim == 0 else r
        if isinstance(sel, slice) and sel.step in (None, 1):
            i0, i1, _ = sel.indices(self.shape[0])
            return self.read(i0, max(i0, i1))
        raise IndexError("only contiguous row ranges are read from an HDF5 dataset")
