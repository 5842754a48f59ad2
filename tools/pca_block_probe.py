import sys, time, os
from pathlib import Path
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from scanpy_amd.preprocessing._pca_solver import GpuBackend, pca_fit
n=1_000_000
be=GpuBackend()
for structure in ("planted","none"):
    x,_=bench.make_matrix(n,2000,0,structure)
    h=be.upload(x); del x
    ref=None
    for blk in ("128","96","112"):
        os.environ["SCAMD_DENSE_BLOCK"]=blk
        r=pca_fit(h,50,backend=be); torch.cuda.synchronize()
        best=1e9
        for _ in range(4):
            t0=time.perf_counter(); r=pca_fit(h,50,backend=be); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
        comp=np.abs(r.components)
        if ref is None: ref=comp
        print(structure, "block", blk, f"{best*1e3:.2f} ms", {k:r.info[k] for k in ("n_outer","n_gemm","chol_retries","residual","block_size") if k in r.info}, "max |dV| vs block 128:", float(np.abs(comp-ref).max()), flush=True)
    del h
