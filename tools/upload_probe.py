import sys, time, os
sys.path.insert(0,'/root/repo')
import torch, numpy as np
import bench
from scanpy_amd.preprocessing._pca_solver import GpuBackend
x,_=bench.make_matrix(1_000_000,2000,0,"planted")
be=GpuBackend()
for _ in range(2): h=be.upload(x); torch.cuda.synchronize()
best=1e9
for _ in range(5):
    t0=time.perf_counter(); h=be.upload(x); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
print("threads", os.environ.get("SCAMD_UPLOAD_THREADS","8"), "piece", os.environ.get("SCAMD_UPLOAD_PIECE_MB","32"), f"upload {best*1e3:.1f} ms = {x.data.nbytes*2/best/1e9:.1f} GB/s", flush=True)
