"""Host-side rate of streaming an `.h5ad` CSR matrix by row chunks (decode only, no device).

step 1 (needs h5py: in the build container, `/opt/conda/bin/python3.9 tools/h5ad_stream_probe.py write N /tmp/p.h5ad`)
writes a synthetic N x 2000 CSR matrix the way anndata lays it out (gzip level 4 + shuffle chunks);
step 2 (`python tools/h5ad_stream_probe.py read /tmp/p.h5ad`) streams it through `_ChunkedRows` with recycled buffers."""
import sys
import time

import numpy as np

if sys.argv[1] == "write":
    import h5py

    n, path = int(sys.argv[2]), sys.argv[3]
    comp = sys.argv[4] if len(sys.argv) > 4 else "gzip"
    rng = np.random.default_rng(0)
    r = 100
    width = 2000 // r
    indices = (np.arange(r, dtype=np.int32) * width)[None, :] + rng.integers(0, width, (n, r), dtype=np.int32)
    data = np.round(np.log1p(rng.lognormal(0, 1, (n, r))), 3).astype(np.float32)
    indptr = np.arange(n + 1, dtype=np.int64) * r
    kw = dict(chunks=(1 << 20,), shuffle=True)
    if comp == "gzip":
        kw.update(compression="gzip", compression_opts=4)
    with h5py.File(path, "w") as f:
        f.attrs["encoding-type"], f.attrs["encoding-version"] = "anndata", "0.1.0"
        x = f.create_group("X")
        x.attrs["encoding-type"], x.attrs["encoding-version"] = "csr_matrix", "0.1.0"
        x.attrs["shape"] = np.array([n, 2000], dtype=np.int64)
        x.create_dataset("data", data=data.reshape(-1), **kw)
        x.create_dataset("indices", data=indices.reshape(-1), **kw)
        x.create_dataset("indptr", data=indptr, **{**kw, "chunks": (min(n + 1, 1 << 18),)})
        for name in ("obs", "var"):
            g = f.create_group(name)
            g.attrs["encoding-type"], g.attrs["encoding-version"] = "dataframe", "0.2.0"
            g.attrs["_index"] = "_index"
            g.attrs["column-order"] = np.array([], dtype=h5py.string_dtype())
            m = n if name == "obs" else 2000
            d = g.create_dataset("_index", data=np.array([f"{name}{i}" for i in range(m)], dtype=h5py.string_dtype()),
                                 chunks=(min(m, 65536),), compression="gzip")
            d.attrs["encoding-type"], d.attrs["encoding-version"] = "string-array", "0.2.0"
    print("written", path)
else:
    import scanpy_amd as sc
    from scanpy_amd.preprocessing._pca_solver import _ChunkedRows

    path = sys.argv[2]
    t = time.perf_counter()
    a = sc.read_h5ad(path, backed="r")
    print(f"open {time.perf_counter() - t:.2f} s  {a.X}")
    n = a.shape[0]
    raw = a.X.nnz * 8 + 8 * (n + 1)

    class Be:
        upload_copies = True

        def upload(self, c):
            return c.data.size

    for step in (250_000, 1_000_000):
        rows = _ChunkedRows(a.X.row_chunks(step), a.shape[1])
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            assert sum(rows.handles(Be())) == a.X.nnz
            ts.append(time.perf_counter() - t)
        print(f"step {step}: first {ts[0]:.3f} s, best {min(ts):.3f} s = {raw / min(ts) / 1e9:.2f} GB/s decoded "
              f"({n / min(ts) / 1e6:.2f} M cells/s)")
    t = time.perf_counter()
    print("absmax", a.X.absmax(), f"{time.perf_counter() - t:.3f} s")
