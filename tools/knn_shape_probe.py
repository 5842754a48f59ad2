"""kNN on random embeddings of several (n, d): does the search answer, how many queries go through the fallback tiers?
    python tools/knn_shape_probe.py 120000x12 120000x16 70000x12 ..."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from scanpy_amd import _kernels as K
from scanpy_amd import _lib

rng = np.random.default_rng(3)
for spec in sys.argv[1:]:
    parts = spec.split("x")
    n, d = int(parts[0]), int(parts[1])
    kind = parts[2] if len(parts) > 2 else "normal"
    x = rng.normal(size=(n, d)).astype(np.float32)
    if kind == "blobs":
        x += 8.0 * rng.normal(size=(32, d)).astype(np.float32)[rng.integers(0, 32, n)]
    try:
        idx, dist, nfb = K.knn(torch.from_numpy(x).cuda(), 15)
        torch.cuda.synchronize()
        lib = _lib.load()
        q = rng.choice(n, 200, replace=False)
        d2 = ((x[q, None, :].astype(np.float64) - x[None, :, :].astype(np.float64)) ** 2).sum(-1) if n <= 200000 else None
        ok = float("nan")
        if d2 is not None:
            ref = np.sort(np.argsort(d2, axis=1)[:, :15], axis=1)
            ok = (np.sort(idx[q].cpu().numpy(), axis=1) == ref).all(1).mean()
        print(spec, "ok", ok, "fallback", nfb, "second tier", lib.scamd_knn_last_second_tier_queries(), "select ms", lib.scamd_knn_last_select_ms(),
              "pairs", lib.scamd_knn_last_select_pairs() / (float(n) * n), flush=True)
    except Exception as e:  # noqa: BLE001
        print(spec, "ERROR", str(e)[:160], flush=True)
