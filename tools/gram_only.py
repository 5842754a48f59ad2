"""The Gram kernel alone at 1M x 2k (device resident), best of 5 -- the unit of every Gram A/B (env knobs are read by the library).
    python tools/gram_only.py [n] [g] [structure]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from scanpy_amd import _kernels as K
from scanpy_amd.preprocessing._pca_solver import GpuBackend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
structure = sys.argv[3] if len(sys.argv) > 3 else "planted"
x, _ = bench.make_matrix(n, g, 0, structure)
be = GpuBackend()
h = be.upload(x)
fn = lambda: K.csr_gram(h[0], h[1], h[2], n, g, 36)
fn(); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"gram n={n} g={g} {structure}: {best * 1e3:.2f} ms", flush=True)
