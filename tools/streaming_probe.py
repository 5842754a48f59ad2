"""Out-of-core PCA on the GPU box (not yet measured there): writes a synthetic n x 2000 CSR matrix as a
zarr-v3 store and as an .h5ad (gzip + shuffle), then times `sc.pp.pca` on the in-memory matrix, on both backed
matrices (chunks resident after the first pass, the default when they fit 40 % of the free HBM) and with
SCAMD_PCA_CHUNK_RESIDENT=0 (every pass streams from disk), and checks that all of them give the same bits.

    python tools/streaming_probe.py [n_obs=1000000] [workdir=/tmp/scamd_stream]      -> one JSON line
"""
from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import scanpy_amd as sc  # noqa: E402
from scanpy_amd.datasets import synthetic_planted  # noqa: E402


def timed(fn, reps=2):
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best, out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    work = Path(sys.argv[2] if len(sys.argv) > 2 else "/tmp/scamd_stream")
    work.mkdir(parents=True, exist_ok=True)
    x, _ = synthetic_planted(n, 2000, seed=0)
    res = {"n_obs": n, "nnz": int(x.nnz), "csr_gb": round((x.nnz * 8 + 8 * (n + 1)) / 1e9, 3)}
    t = time.perf_counter()
    sc.write_zarr(work / "x.zarr", sc.AnnData(x))
    res["write_zarr_s"] = round(time.perf_counter() - t, 2)
    t = time.perf_counter()
    sc.write_h5ad(work / "x.h5ad", sc.AnnData(x), compression="gzip")
    res["write_h5ad_gzip_s"] = round(time.perf_counter() - t, 2)

    def fit(adata, **kw):
        sc.pp.pca(adata, **kw)
        return adata.obsm["X_pca"], adata.varm["PCs"]

    mem = sc.AnnData(x)
    res["pca_in_memory_s"], ref = timed(lambda: fit(mem))
    for name, reader, path in (("zarr", sc.read_zarr, work / "x.zarr"), ("h5ad", sc.read_h5ad, work / "x.h5ad")):
        for resident in ("1", "0"):
            os.environ["SCAMD_PCA_CHUNK_RESIDENT"] = resident
            for step in (250_000, 1_000_000):
                key = f"pca_backed_{name}_{'resident' if resident == '1' else 'streamed'}_chunk{step // 1000}k_s"

                def run(reader=reader, path=path, step=step):
                    return fit(reader(path, backed="r"), chunk_size=step)

                res[key], got = timed(run)
                res[key] = round(res[key], 3)
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), key
    os.environ["SCAMD_PCA_CHUNK_RESIDENT"] = "0"
    os.environ["SCAMD_PIN_STAGING"] = "1"  # page-locked staging buffers (opt-in): does the streamed fit gain?
    for name, reader, path in (("zarr", sc.read_zarr, work / "x.zarr"), ("h5ad", sc.read_h5ad, work / "x.h5ad")):
        key = f"pca_backed_{name}_streamed_pinned_chunk1000k_s"
        res[key], got = timed(lambda reader=reader, path=path: fit(reader(path, backed="r"), chunk_size=1_000_000))
        res[key] = round(res[key], 3)
        assert np.array_equal(got[0], ref[0]), key
    os.environ.pop("SCAMD_PIN_STAGING", None)
    os.environ.pop("SCAMD_PCA_CHUNK_RESIDENT", None)
    res["pca_in_memory_s"] = round(res["pca_in_memory_s"], 3)
    res["bitwise_equal"] = True
    print(json.dumps(res))


if __name__ == "__main__":
    main()
