"""PCA stage timing at 1M x 2k (device resident): whole fit + the Gram kernel alone, best of 5.
    python tools/pca_stage_probe.py            (env knobs are read by the library)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from scanpy_amd import _kernels as K
from scanpy_amd.preprocessing._pca_solver import GpuBackend, pca_fit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, _ = bench.make_matrix(n, 2000, 0, "planted")
be = GpuBackend()
h = be.upload(x)
for name, fn in (("pca_fit", lambda: pca_fit(h, 50, backend=be)), ("gram", lambda: K.csr_gram(h[0], h[1], h[2], n, 2000, 36)),
                 ("pca_csr (one C call)", lambda: K.pca_csr(h[0], h[1], h[2], n, 2000, 50))):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {best * 1e3:.2f} ms", flush=True)
