"""Wall time of sc.pp.pca one-shot vs chunked (resident / streamed) on a host CSR (probe, not a test)."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import scanpy_amd as sc
from scanpy_amd.datasets import synthetic_planted

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
x, _ = synthetic_planted(n, 2000, seed=0)
ref = None
for label, kw, env in (("warm-up", {}, "1"), ("one-shot", {}, "1"), ("chunked resident", dict(chunked=True, chunk_size=500_000), "1"),
                       ("chunked streamed", dict(chunked=True, chunk_size=500_000), "0")):
    os.environ["SCAMD_PCA_CHUNK_RESIDENT"] = env
    a = sc.AnnData(x)
    t = time.perf_counter()
    sc.pp.pca(a, **kw)
    dt = time.perf_counter() - t
    if ref is None:
        ref = a.varm["PCs"]
    print(f"{label:18s} {dt * 1e3:8.0f} ms   identical={np.array_equal(ref, a.varm['PCs'])}", flush=True)
