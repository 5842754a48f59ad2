"""The Gram entry on matrices denser than the bench's 5 %: rows with more than 16 entries per 128-gene tile take the CSR tail of
the packed kernel.    python tools/gram_density_probe.py [n] [density ...]     (SCAMD_GRAM_LEGACY=1: the round-2 kernel)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from scipy import sparse
from scanpy_amd import _kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
for dens in [float(a) for a in sys.argv[2:]] or [0.05, 0.08, 0.12, 0.2]:
    rng = np.random.default_rng(0)
    g = 2000
    r = int(g * dens)
    cols = np.sort(np.argsort(rng.random((n, g)), axis=1)[:, :r].astype(np.int32), axis=1) if n * g <= 4e8 else None
    indptr = torch.arange(0, n * r + 1, r, dtype=torch.int64).cuda()
    indices = torch.from_numpy(cols.reshape(-1)).cuda()
    data = torch.from_numpy(np.log1p(np.exp(rng.standard_normal(n * r))).astype(np.float32)).cuda()
    fn = lambda: K.csr_gram(indptr, indices, data, n, g, 30)
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"gram n={n} density {dens}: {best * 1e3:.2f} ms ({r} entries per row, {r * 128 / g:.1f} per tile)", flush=True)
