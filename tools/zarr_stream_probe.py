"""Host-side rate of the out-of-core path: write a synthetic CSR store, then stream it by row chunks (decode only, no
device) -- what `pp.pca` on a `read_zarr(backed='r')` matrix can be fed with.  usage: zarr_stream_probe.py [n_obs] [dir]"""
import sys
import time

import numpy as np

import scanpy_amd as sc
from scanpy_amd.datasets import synthetic_planted

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
root = sys.argv[2] if len(sys.argv) > 2 else "/tmp/zs"
x, _ = synthetic_planted(n, 2000, seed=0)
raw = x.data.nbytes + x.indices.nbytes + 8 * (n + 1)
t = time.perf_counter()
sc.write_zarr(f"{root}/probe.zarr", sc.AnnData(x))
tw = time.perf_counter() - t
t = time.perf_counter()
a = sc.read_zarr(f"{root}/probe.zarr", backed="r")
to = time.perf_counter() - t
for step in (250_000, 1_000_000):
    t = time.perf_counter()
    tot = 0
    for c in a.X.row_chunks(step):
        r = c.load()
        tot += r.data.size
    tr = time.perf_counter() - t
    assert tot == x.nnz
    print(f"step {step}: stream {tr:.3f} s = {raw / tr / 1e9:.2f} GB/s decoded ({n / tr / 1e6:.2f} M cells/s)")
r = a.X.rows(0, n)
assert np.array_equal(r.data, x.data) and np.array_equal(r.indices, x.indices)
print(f"n={n} nnz={x.nnz} raw={raw / 1e9:.2f} GB  write {tw:.2f} s  open {to * 1e3:.1f} ms")
