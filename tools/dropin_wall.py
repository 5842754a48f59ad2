"""Wall time of the drop-in calls on a host AnnData (H2D, kernels, D2H, scipy/pandas slot construction), per call."""
from __future__ import annotations

import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import scanpy_amd as sc  # noqa: E402
from scanpy_amd.datasets import synthetic_planted  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    x, _ = synthetic_planted(n, 2000, seed=0)
    for rep in range(2):
        adata = sc.AnnData(x)
        t = {}
        t0 = time.perf_counter()
        sc.pp.pca(adata)
        t["pca"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        sc.pp.neighbors(adata)
        t["neighbors"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        sc.tl.leiden(adata, flavor="igraph", n_iterations=-1)
        t["leiden"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        sc.tl.umap(adata)
        t["umap"] = time.perf_counter() - t0
        tot = t["pca"] + t["neighbors"] + t["leiden"]
        print(f"rep {rep}: " + "  ".join(f"{k} {v * 1e3:.0f} ms" for k, v in t.items()) +
              f"  | pca+neighbors+leiden {tot * 1e3:.0f} ms = {n / tot:.0f} cells/s host-in/host-out; "
              f"{adata.obs['leiden'].nunique()} clusters", flush=True)


if __name__ == "__main__":
    main()
