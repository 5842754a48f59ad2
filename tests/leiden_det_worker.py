"""Worker of tests/test_gpu_leiden_determinism.py: ONE process = the path's own fuzzy graph of a structure-less / weakly
structured matrix, then Leiden on it under every kernel variant.  Prints one JSON object (sha1 of every result)."""
from __future__ import annotations

import hashlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import os as _os  # noqa: E402

if _os.environ.get("SCAMD_TESTS_ON_EMULATOR") == "1":  # the host-emulated kernel library, tests/emu/README.md
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    import patch_torch  # noqa: E402

    patch_torch.activate()

VARIANTS = [
    {},                                                        # default kernels
    {},                                                        # again: dirty workspace, same process
    {"SCAMD_LEIDEN_QUAD": "0"},                                # one wave per vertex everywhere
    {"SCAMD_LEIDEN_QUAD": "1"},                                # 16 lanes per vertex everywhere
    {"SCAMD_LEIDEN_QUAD": "2"},                                # 32 lanes per vertex everywhere
    {"SCAMD_LEIDEN_AGG_WAVE_MAX": "48", "SCAMD_LEIDEN_AGG_MID_MAX": "256", "SCAMD_LEIDEN_AGG_PASS_KEYS": "512"},
]


def sha(t) -> str:
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def main():
    n, structure = int(sys.argv[1]), sys.argv[2]
    variants = VARIANTS if len(sys.argv) < 4 or sys.argv[3] == "all" else VARIANTS[:1]
    import torch

    import bench
    from scanpy_amd import _kernels as K
    from scanpy_amd._pipeline import run_path
    from scanpy_amd.preprocessing._pca_solver import GpuBackend

    x, _ = bench.make_matrix(n, 2000, 0, structure)
    backend = GpuBackend()
    res = run_path(backend.upload(x), n, backend=backend)
    ip, ix, w = res.conn_indptr, res.conn_indices, res.conn_data
    out = {"graph": {"indptr": sha(ip), "indices": sha(ix), "data": sha(w)}, "x_pca": sha(res.x_pca),
           "knn_idx": sha(res.knn_indices), "path": {"labels": sha(res.labels), "q": res.modularity, "nc": res.n_communities},
           "runs": []}
    keys = sorted({k for v in VARIANTS for k in v})
    for env in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        labels, q, nc = K.leiden(ip, ix, w, n)
        torch.cuda.synchronize()
        out["runs"].append({"env": env, "labels": sha(labels), "q": q, "nc": nc})
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
