"""Probe (not a test): is the path bitwise reproducible where it is hard (structure none / weak)?

    python tools/leiden_determinism_probe.py 200000 none 3 outdir

Builds the path's own fuzzy graph, prints sha1 of every stage's output (so two PROCESSES / boxes can be diffed line by
line), then runs Leiden `reps` times on the same resident graph with SCAMD_LEIDEN_DEBUG=1, one trace file per run,
and reports the first trace line at which two runs diverge."""
from __future__ import annotations

import hashlib
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def sha(t) -> str:
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


class StderrTo:
    """redirect the C library's stderr (fd 2) into a file for the duration of the block"""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        sys.stderr.flush()
        self.saved = os.dup(2)
        self.f = open(self.path, "w")
        os.dup2(self.f.fileno(), 2)

    def __exit__(self, *a):
        os.dup2(self.saved, 2)
        os.close(self.saved)
        self.f.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    structure = sys.argv[2] if len(sys.argv) > 2 else "none"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out = Path(sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/ldet")
    out.mkdir(parents=True, exist_ok=True)
    import torch

    import bench
    from scanpy_amd import _kernels as K
    from scanpy_amd._pipeline import run_path
    from scanpy_amd.preprocessing._pca_solver import GpuBackend

    x, _ = bench.make_matrix(n, 2000, 0, structure)
    backend = GpuBackend()
    h = backend.upload(x)
    os.environ.pop("SCAMD_LEIDEN_DEBUG", None)
    tag = f"{structure}_{n}"
    for rep in range(2):
        res = run_path(h, n, backend=backend)
        print(f"{tag} path#{rep}: x_pca {sha(res.x_pca)} knn_idx {sha(res.knn_indices)} knn_dist {sha(res.knn_distances)} "
              f"conn_indptr {sha(res.conn_indptr)} conn_indices {sha(res.conn_indices)} conn_data {sha(res.conn_data)} "
              f"labels {sha(res.labels)} nc {res.n_communities} Q {res.modularity!r}", flush=True)
    ip, ix, w = res.conn_indptr, res.conn_indices, res.conn_data
    os.environ["SCAMD_LEIDEN_DEBUG"] = "1"
    traces = []
    for rep in range(reps):
        f = out / f"trace_{tag}_{os.getpid()}_{rep}.log"
        with StderrTo(f):
            labels, q, nc = K.leiden(ip, ix, w, n)
            torch.cuda.synchronize()
        # wall times differ run to run: keep only the counters of each line
        lines = [ln.split(": local moving")[0] if ": local moving" in ln else ln for ln in f.read_text().splitlines()]
        lines = [ln for ln in lines if "aggregate" not in ln or "->" in ln]
        lines = [ln.split(" aggregate ")[0] + " -> " + ln.split("->")[1] if " aggregate " in ln else ln for ln in lines]
        traces.append(lines)
        print(f"{tag} leiden#{rep}: labels {sha(labels)} nc {nc} Q {q!r} trace_lines {len(lines)}", flush=True)
    os.environ.pop("SCAMD_LEIDEN_DEBUG", None)
    for rep in range(1, reps):
        a, b = traces[0], traces[rep]
        for i, (la, lb) in enumerate(zip(a, b)):
            if la != lb:
                print(f"{tag} run 0 vs {rep}: first divergence at trace line {i}:\n   {la}\n   {lb}")
                for j in range(max(0, i - 6), i):
                    print(f"   (before) {a[j]}")
                break
        else:
            print(f"{tag} run 0 vs {rep}: traces identical ({len(a)} vs {len(b)} lines)")


if __name__ == "__main__":
    main()
