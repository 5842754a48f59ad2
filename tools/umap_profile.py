"""Where does sc.tl.umap spend its wall time at 1M?  (probe, not a test)"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from scanpy_amd import _kernels as K
from scanpy_amd._pipeline import run_path
from scanpy_amd.datasets import synthetic_planted
from scanpy_amd.preprocessing._pca_solver import GpuBackend
from scanpy_amd.tools import _umap
from scipy import sparse

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, _ = synthetic_planted(n, 2000, seed=0)
be = GpuBackend()
res = run_path(be.upload(x), n, backend=be)
conn = sparse.csr_matrix((res.conn_data.cpu().numpy(), res.conn_indices.cpu().numpy(), res.conn_indptr.cpu().numpy()), shape=(n, n))
def T(label, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t) * 1e3:.0f} ms", flush=True); return r
csr = T("csr_matrix + canonical check", lambda: (lambda c: (c, c.has_canonical_format))(sparse.csr_matrix(conn))[0])
dev = torch.device("cuda")
up = T("upload", lambda: (torch.from_numpy(csr.indptr.astype(np.int64)).to(dev), torch.from_numpy(csr.indices.astype(np.int32)).to(dev), torch.from_numpy(np.ascontiguousarray(csr.data, dtype=np.float32)).to(dev)))
ip, ix, w, eps = T("prune device", lambda: _umap.prune_and_schedule_device(*up, n, 200))
ini = T("spectral init", lambda: _umap._spectral_init(ip, ix, w, n, 2, 0))
y = torch.rand((n, 2), device=dev) * 10
a, b = _umap.find_ab_params(1.0, 0.5)
T("optimize 200 epochs", lambda: K.umap_optimize_(ip, ix, eps, n, y, n_epochs=200, a=a, b=b, seed=0))
T("spmm l=7", lambda: K.spmm(ip, ix, w, n, n, torch.randn((n, 7), device=dev).contiguous()))
from scanpy_amd.preprocessing._pca_solver import _cholqr2
v = torch.randn((n, 7), dtype=torch.float64, device=dev)
T("cholqr2", lambda: _cholqr2(v))
T("f64->f32->f64", lambda: v.to(torch.float32).contiguous().to(torch.float64))
