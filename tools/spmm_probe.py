"""The scores SpMM alone at 1M x 2k: loadings of 50 columns against the same loadings padded to 64 (rows of B aligned to
128-byte lines), best of 5.    python tools/spmm_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from scanpy_amd import _kernels as K
from scanpy_amd.preprocessing._pca_solver import GpuBackend

n, g = 1_000_000, 2000
x, _ = bench.make_matrix(n, g, 0, "planted")
be = GpuBackend()
h = be.upload(x)
torch.manual_seed(0)
v50 = torch.randn(g, 50, device="cuda", dtype=torch.float32)
v64 = torch.zeros(g, 64, device="cuda", dtype=torch.float32)
v64[:, :50] = v50
for name, b in (("l=50", v50), ("l=64 (padded)", v64), ("l=32", v50[:, :32].contiguous())):
    fn = lambda: K.spmm(h[0], h[1], h[2], n, g, b)
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"spmm {name}: {best * 1e3:.2f} ms", flush=True)
