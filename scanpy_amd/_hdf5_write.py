"""A write-once HDF5 writer: what `write_h5ad` needs, laid out the way h5py's default settings lay it out.

The reference writes results with `adata.write(filename, compression=...)` (`src/scanpy/readwrite.py:657-740`, through
h5py, which this image's interpreter does not have).  This module produces the same classic container from the HDF5
File Format Specification -- superblock version 0, version-1 object headers, symbol-table groups (B-tree v1 + local
heap + "SNOD" nodes), contiguous or chunked (B-tree v1) datasets with the shuffle + deflate pipeline, variable-length
UTF-8 strings in global heap collections, h5py's bool enum -- message bytes modelled on what h5py 3.3 / HDF5 1.10.6
emits for the same objects.  Pinned by reading the files back with the HDF5 library itself where it is available (the
image's conda Python has h5py: `tests/test_hdf5_write_cpu.py`) and with `scanpy_amd/_hdf5.py` everywhere.

The file is written front to back in post-order (children before the objects that point at them), so nothing is ever
patched except the superblock, and big datasets stream chunk by chunk (deflate runs on the decode thread pool).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_UNDEF = b"\xff" * 8
GROUP_LEAF_K = 16       # symbol-table leaves ("SNOD") hold up to 2K entries
GROUP_INTERNAL_K = 32   # group B-tree nodes hold up to 2K children
CHUNK_K = 32            # chunk B-tree nodes hold up to 2K entries (the library's default for superblock version 0)
CHUNK_BYTES = 4 << 20   # raw bytes per chunk of a compressed dataset


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# ---------------------------------------------------------------------------------------------------------------------
# datatype / dataspace / attribute messages

_BOOL_ENUM = (bytes([0x18, 0x02, 0, 0]) + struct.pack("<I", 1) + bytes([0x10, 0x08, 0, 0]) + struct.pack("<IHH", 1, 0, 8)
              + b"FALSE\0\0\0" + b"TRUE\0\0\0\0" + bytes([0, 1]))
_VLEN_UTF8 = (bytes([0x19, 0x01, 0x01, 0]) + struct.pack("<I", 16) + bytes([0x10, 0, 0, 0])
              + struct.pack("<IHH", 1, 0, 8))


def _dtype_message(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt == np.dtype(bool):
        return _BOOL_ENUM
    if dt.kind in "iu":
        return (bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0]) + struct.pack("<IHH", dt.itemsize, 0,
                                                                                 8 * dt.itemsize))
    if dt.kind == "f":
        layout = {2: (15, 10, 5, 0, 10, 15), 4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023)}[dt.itemsize]
        sign, eloc, esize, mloc, msize, bias = layout
        return (bytes([0x11, 0x20, sign, 0]) + struct.pack("<IHH", dt.itemsize, 0, 8 * dt.itemsize)
                + bytes([eloc, esize, mloc, msize]) + struct.pack("<I", bias))
    if dt.names:  # compound, version 1 member layout
        out = bytes([0x16, len(dt.names) & 0xFF, len(dt.names) >> 8, 0]) + struct.pack("<I", dt.itemsize)
        for name in dt.names:
            ft, off = dt.fields[name][0], dt.fields[name][1]
            nm = name.encode("utf-8") + b"\0"
            out += nm + b"\0" * (-len(nm) % 8) + struct.pack("<I", off) + bytes(1 + 3 + 4 + 4 + 16) + _dtype_message(ft)
        return out
    if dt.kind == "S":
        return bytes([0x13, 0x00, 0, 0]) + struct.pack("<I", dt.itemsize)
    raise TypeError(f"dtype {dt} has no HDF5 encoding here")


def _space_message(shape) -> bytes:
    if shape is None:  # null dataspace (h5py.Empty): version 2, type 2
        return bytes([2, 0, 0, 2])
    rank = len(shape)
    if rank == 0:
        return bytes([1, 0, 0, 0, 0, 0, 0, 0])
    dims = b"".join(struct.pack("<Q", int(s)) for s in shape)
    return bytes([1, rank, 1, 0, 0, 0, 0, 0]) + dims + dims  # (maximum dimensions = dimensions)


def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
    body = _pad8(body)
    if len(body) > 0xFFF8:
        raise ValueError("an HDF5 header message is limited to 64 KiB (too many columns / too long an attribute)")
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


# ---------------------------------------------------------------------------------------------------------------------
# the file


class H5Writer:
    def __init__(self, path):
        self.f = open(path, "wb")
        self.f.write(b"\0" * 96)  # superblock, filled in by `close`
        self.pos = 96

    def put(self, data: bytes, align: int = 8) -> int:
        """append -> address"""
        pad = -self.pos % align
        if pad:
            self.f.write(b"\0" * pad)
            self.pos += pad
        addr = self.pos
        self.f.write(data)
        self.pos += len(data)
        return addr

    # -- variable-length strings
    def strings(self, values) -> bytes:
        """store `values` in global heap collections -> the concatenated (length, collection address, index) elements"""
        enc = [str(v).encode("utf-8") for v in values]
        refs = bytearray()
        i = 0
        while i < len(enc):
            body = bytearray()
            items = []
            idx = 1
            while i < len(enc) and idx < 0xFFFF and (not items or len(body) + 16 + len(enc[i]) + 8 < (1 << 20)):
                data = enc[i]
                body += struct.pack("<HHIQ", idx, 1, 0, len(data)) + _pad8(data)
                items.append((idx, len(data)))
                idx += 1
                i += 1
            size = max(4096, 16 + len(body) + 16)
            size += -size % 8
            free = size - 16 - len(body)
            # object 0 = the free space; its size field counts its own 16-byte header (the library advances by it)
            body += struct.pack("<HHIQ", 0, 0, 0, free) + b"\0" * (free - 16)
            addr = self.put(b"GCOL" + bytes([1, 0, 0, 0]) + struct.pack("<Q", size) + bytes(body))
            for j, n in items:
                refs += struct.pack("<IQI", n, addr, j)
        return bytes(refs)

    # -- attributes
    def _attribute(self, name: str, value) -> bytes:
        nm = name.encode("utf-8") + b"\0"
        if value is None:
            dt, space, data = _dtype_message(np.dtype("<f4")), _space_message(None), b""
        elif isinstance(value, str):
            dt, space, data = _VLEN_UTF8, _space_message(()), self.strings([value])
        else:
            arr = np.asarray(value)
            if arr.dtype.kind in "OU":
                dt, space, data = _VLEN_UTF8, _space_message(arr.shape), self.strings(arr.reshape(-1).tolist())
            else:
                if arr.dtype.byteorder == ">":
                    arr = arr.astype(arr.dtype.newbyteorder("<"))
                dt, space = _dtype_message(arr.dtype), _space_message(arr.shape)
                data = (arr.astype(np.int8) if arr.dtype == bool else arr).tobytes()
        body = (struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(space)) + _pad8(nm) + _pad8(dt) + _pad8(space) + data)
        return _message(0x0C, body)

    def _header(self, messages: list[bytes]) -> int:
        body = b"".join(messages)
        return self.put(struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body)

    # -- datasets
    def dataset(self, data, attrs: dict | None = None, *, compression: str | None = None, level: int = 4,
                chunk_rows: int | None = None) -> int:
        """write one dataset (numeric / bool / record array, or a sequence of str) -> its object header address"""
        if hasattr(data, "pieces"):  # a 1-d column that arrives in consecutive pieces (an on-disk matrix being copied)
            return self._streamed(data, attrs, compression, level)
        is_text = isinstance(data, str) or (isinstance(data, np.ndarray) and data.dtype.kind in "OU") \
            or (not isinstance(data, np.ndarray) and not np.isscalar(data) and len(data) > 0
                and isinstance(data[0], str))
        if is_text:
            arr = np.asarray(data, dtype=object)
            shape = arr.shape
            raw_dtype = None
            raw = np.frombuffer(self.strings(arr.reshape(-1).tolist()), dtype=np.uint8)
            dt_msg, esize = _VLEN_UTF8, 16
        else:
            arr = np.asarray(data)
            if arr.dtype.byteorder == ">":
                arr = arr.astype(arr.dtype.newbyteorder("<"))
            arr = np.asarray(arr, order="C")
            shape = arr.shape
            raw_dtype = arr.dtype
            raw = None
            dt_msg, esize = _dtype_message(arr.dtype), arr.dtype.itemsize
        msgs = [_message(0x01, _space_message(shape)), _message(0x03, dt_msg, 1)]
        n_elem = int(np.prod(shape)) if shape else 1
        chunked = compression is not None and len(shape) >= 1 and n_elem > 0 and not is_text
        if not chunked:
            msgs.append(_message(0x05, bytes([2, 2, 2, 1, 0, 0, 0, 0]), 1))
            flat = raw if is_text else (arr.astype(np.int8) if raw_dtype == bool else arr)
            # big arrays go to the file through the buffer protocol (no `tobytes` copy of a 400 MB column)
            payload = flat.tobytes() if flat.nbytes < (1 << 20) else memoryview(flat.reshape(-1)).cast("B")
            addr = self.put(payload) if len(payload) else None
            msgs.append(_message(0x08, bytes([3, 1]) + (struct.pack("<Q", addr) if addr is not None else _UNDEF)
                                 + struct.pack("<Q", len(payload))))
        else:
            if compression != "gzip":
                raise ValueError("compression must be None or 'gzip'")
            row = int(np.prod(shape[1:])) * esize
            c0 = chunk_rows or max(1, min(shape[0], CHUNK_BYTES // max(1, row)))
            cshape = (c0,) + tuple(shape[1:])
            store = (arr.astype(np.int8) if raw_dtype == bool else arr)
            btree = self._chunks(store, cshape, esize, level)
            msgs.append(_message(0x05, bytes([2, 3, 0, 1, 0, 0, 0, 0]), 1))
            pipeline = (bytes([1, 2, 0, 0, 0, 0, 0, 0])
                        + struct.pack("<HHHH", 2, 8, 1, 1) + b"shuffle\0" + struct.pack("<II", esize, 0)
                        + struct.pack("<HHHH", 1, 8, 1, 1) + b"deflate\0" + struct.pack("<II", level, 0))
            msgs.append(_message(0x0B, pipeline, 1))
            msgs.append(_message(0x08, bytes([3, 2, len(shape) + 1]) + struct.pack("<Q", btree)
                                 + b"".join(struct.pack("<I", c) for c in cshape) + struct.pack("<I", esize)))
        for k, v in (attrs or {}).items():
            msgs.append(self._attribute(k, v))
        return self._header(msgs)

    def _streamed(self, col, attrs, compression, level: int) -> int:
        """`col`: .shape (n,), .dtype, .pieces() -> consecutive 1-d arrays; never more than a piece + a chunk in memory"""
        dt = np.dtype(col.dtype)
        n, esize = int(col.shape[0]), dt.itemsize
        msgs = [_message(0x01, _space_message((n,))), _message(0x03, _dtype_message(dt), 1)]
        if compression is None or n == 0:
            addr, total = None, 0
            for piece in col.pieces():
                piece = np.ascontiguousarray(piece, dtype=dt)
                if piece.size == 0:
                    continue
                a = self.put(memoryview(piece).cast("B"), align=8 if addr is None else 1)
                addr = a if addr is None else addr
                total += piece.nbytes
            assert total == n * esize, "streamed column shorter than announced"
            msgs.append(_message(0x05, bytes([2, 2, 2, 1, 0, 0, 0, 0]), 1))
            msgs.append(_message(0x08, bytes([3, 1]) + (struct.pack("<Q", addr) if addr is not None else _UNDEF)
                                 + struct.pack("<Q", total)))
        else:
            c0 = max(1, min(n, CHUNK_BYTES // esize))
            buf = np.empty(c0, dtype=dt)
            fill, done = 0, 0
            entries = []

            def flush(count: int) -> None:
                nonlocal done
                if count < c0:
                    buf[count:] = 0  # edge chunks are stored whole
                b = buf.view(np.uint8).reshape(-1, esize)
                blob = zlib.compress(np.ascontiguousarray(b.T).tobytes() if esize > 1 else b.tobytes(), level)
                entries.append((len(blob), (done,), self.put(blob, align=1)))
                done += c0

            for piece in col.pieces():
                piece = np.ascontiguousarray(piece, dtype=dt)
                p = 0
                while p < piece.size:
                    take = min(c0 - fill, piece.size - p)
                    buf[fill:fill + take] = piece[p:p + take]
                    fill += take
                    p += take
                    if fill == c0:
                        flush(c0)
                        fill = 0
            if fill:
                flush(fill)
            assert done >= n, "streamed column shorter than announced"
            btree = self._chunk_btree(entries, 1, c0)
            msgs.append(_message(0x05, bytes([2, 3, 0, 1, 0, 0, 0, 0]), 1))
            pipeline = (bytes([1, 2, 0, 0, 0, 0, 0, 0])
                        + struct.pack("<HHHH", 2, 8, 1, 1) + b"shuffle\0" + struct.pack("<II", esize, 0)
                        + struct.pack("<HHHH", 1, 8, 1, 1) + b"deflate\0" + struct.pack("<II", level, 0))
            msgs.append(_message(0x0B, pipeline, 1))
            msgs.append(_message(0x08, bytes([3, 2, 2]) + struct.pack("<Q", btree) + struct.pack("<II", c0, esize)))
        for k, v in (attrs or {}).items():
            msgs.append(self._attribute(k, v))
        return self._header(msgs)

    def _chunks(self, arr: np.ndarray, cshape, esize: int, level: int) -> int:
        """chunks along axis 0 (the other axes whole): shuffle + deflate each, write them, then their B-tree"""
        from ._zarr3 import decode_pool

        n0, c0 = arr.shape[0], cshape[0]
        nd = arr.ndim

        def encode(i0: int) -> bytes:
            piece = arr[i0:i0 + c0]
            if piece.shape[0] < c0:  # edge chunks are stored whole
                full = np.zeros(cshape, dtype=arr.dtype)
                full[:piece.shape[0]] = piece
                piece = full
            b = np.ascontiguousarray(piece).view(np.uint8).reshape(-1, esize)
            shuffled = np.ascontiguousarray(b.T).tobytes() if esize > 1 else b.tobytes()
            return zlib.compress(shuffled, level)

        starts = list(range(0, n0, c0))
        entries = []  # (stored size, offsets tuple, address)
        for lo in range(0, len(starts), 64):
            batch = starts[lo:lo + 64]
            for i0, blob in zip(batch, decode_pool().map(encode, batch)):
                entries.append((len(blob), (i0,) + (0,) * (nd - 1), self.put(blob, align=1)))
        return self._chunk_btree(entries, nd, c0)

    def _chunk_btree(self, entries, nd: int, c0: int) -> int:
        """B-tree v1 (node type 1) over written chunks: entries = [(stored size, origin tuple, address)] in order"""
        ksz = 8 + 8 * (nd + 1)

        def key(size: int, offsets) -> bytes:
            return struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in offsets) + struct.pack("<Q", 0)

        end_key = key(0, (entries[-1][1][0] + c0,) + (0,) * (nd - 1))
        level_items = [(key(sz, off), addr) for sz, off, addr in entries]  # (first key of the child, child address)
        node_size = 24 + 2 * CHUNK_K * (ksz + 8) + ksz
        lvl = 0
        while True:
            groups = [level_items[i:i + 2 * CHUNK_K] for i in range(0, len(level_items), 2 * CHUNK_K)]
            base = self.put(b"", align=8)
            addrs = [base + j * node_size for j in range(len(groups))]
            nxt = []
            for j, grp in enumerate(groups):
                left = struct.pack("<Q", addrs[j - 1]) if j else _UNDEF
                right = struct.pack("<Q", addrs[j + 1]) if j + 1 < len(groups) else _UNDEF
                body = b"TREE" + bytes([1, lvl]) + struct.pack("<H", len(grp)) + left + right
                for k_, child in grp:
                    body += k_ + struct.pack("<Q", child)
                body += groups[j + 1][0][0] if j + 1 < len(groups) else end_key
                self.put(body + b"\0" * (node_size - len(body)), align=1)
                nxt.append((grp[0][0], addrs[j]))
            if len(groups) == 1:
                return addrs[0]
            level_items, lvl = nxt, lvl + 1

    # -- groups
    def group(self, children: dict[str, int], attrs: dict | None = None) -> int:
        """write a group over already written children (name -> object header address) -> its header address"""
        names = sorted(children, key=lambda s: s.encode("utf-8"))
        heap = bytearray(8)  # offset 0: the empty string
        offsets = {}
        for nm in names:
            offsets[nm] = len(heap)
            heap += _pad8(nm.encode("utf-8") + b"\0")
        heap_data = self.put(bytes(heap))
        heap_addr = self.put(b"HEAP" + bytes(4) + struct.pack("<QQQ", len(heap), 1, heap_data))  # free list: none (1)
        # leaves
        leaf_size = 8 + 2 * GROUP_LEAF_K * 40
        leaves = [names[i:i + 2 * GROUP_LEAF_K] for i in range(0, len(names), 2 * GROUP_LEAF_K)]
        items = []  # (largest name offset in the subtree, address)
        for grp in leaves:
            body = b"SNOD" + bytes([1, 0]) + struct.pack("<H", len(grp))
            for nm in grp:
                body += struct.pack("<QQII16x", offsets[nm], children[nm], 0, 0)
            items.append((offsets[grp[-1]] if grp else 0, self.put(body + b"\0" * (leaf_size - len(body)))))
        node_size = 24 + 2 * GROUP_INTERNAL_K * 16 + 8
        lvl = 0
        if not names:  # an empty group: a B-tree node without entries, as the library writes it
            items = []
            body = b"TREE" + bytes([0, 0]) + struct.pack("<H", 0) + _UNDEF + _UNDEF + struct.pack("<Q", 0)
            btree = self.put(body + b"\0" * (node_size - len(body)))
        while names:
            groups = [items[i:i + 2 * GROUP_INTERNAL_K] for i in range(0, len(items), 2 * GROUP_INTERNAL_K)]
            base = self.put(b"", align=8)
            addrs = [base + j * node_size for j in range(len(groups))]
            nxt = []
            for j, grp in enumerate(groups):
                left = struct.pack("<Q", addrs[j - 1]) if j else _UNDEF
                right = struct.pack("<Q", addrs[j + 1]) if j + 1 < len(groups) else _UNDEF
                first_key = groups[j - 1][-1][0] if j else 0  # largest name to the left of this node (0 = "")
                body = b"TREE" + bytes([0, lvl]) + struct.pack("<H", len(grp)) + left + right + struct.pack("<Q", first_key)
                for mx, child in grp:
                    body += struct.pack("<QQ", child, mx)
                self.put(body + b"\0" * (node_size - len(body)), align=1)
                nxt.append((grp[-1][0], addrs[j]))
            if len(groups) == 1:
                btree = addrs[0]
                break
            items, lvl = nxt, lvl + 1
        msgs = [_message(0x11, struct.pack("<QQ", btree, heap_addr))]
        for k, v in (attrs or {}).items():
            msgs.append(self._attribute(k, v))
        self._last_group = (btree, heap_addr)
        return self._header(msgs)

    def close(self, root: int) -> None:
        btree, heap = self._last_group  # the root group is the last group written
        eof = self.pos
        sb = (b"\x89HDF\r\n\x1a\n" + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
              + struct.pack("<Q", 0) + _UNDEF + struct.pack("<Q", eof) + _UNDEF
              + struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap))
        assert len(sb) == 96
        self.f.seek(0)
        self.f.write(sb)
        self.f.close()


# ---------------------------------------------------------------------------------------------------------------------
# a tree of nodes, written in post-order


class Node:
    """group (children: name -> Node) or dataset (data set) with attributes"""

    def __init__(self, attrs=None, data=None, *, is_group=False, compression=None):
        self.attrs, self.data, self.is_group, self.compression = dict(attrs or {}), data, is_group, compression
        self.children: dict[str, Node] = {}

    def child(self, path: str) -> "Node":
        node = self
        for part in [p for p in path.split("/") if p]:
            node = node.children[part]
        return node

    def add(self, path: str, node: "Node") -> None:
        parts = [p for p in path.split("/") if p]
        self.child("/".join(parts[:-1])).children[parts[-1]] = node


def write_tree(path, root: Node, *, level: int = 4) -> None:
    w = H5Writer(path)

    def emit(node: Node) -> int:
        if not node.is_group:
            return w.dataset(node.data, node.attrs, compression=node.compression, level=level)
        kids = {name: emit(ch) for name, ch in node.children.items()}
        return w.group(kids, node.attrs)

    try:
        addr = emit(root)
    except BaseException:
        w.f.close()
        raise
    w.close(addr)
