"""Where does the GPU Leiden sit in the oracle's distribution on an AMBIGUOUS graph?  (not a test)

The `weak` sample of bench.py (overlapping programmes): the CPU chain builds the graph; the CPU oracle runs seeds 0..4, the
GPU runs seeds 0..4 under several settings of its knobs (class sub-rounds of the local moving / of the refinement, the
early stop of the local moving, one-workgroup small levels).  Prints modularity and ARI against the planted truth of every
run, pairwise ARI medians, and the run time -- which phase (if any) costs recovered truth.

    python tools/leiden_weak_probe.py [n_cells] [structure]
"""
from __future__ import annotations

import itertools
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    structure = sys.argv[2] if len(sys.argv) > 2 else "weak"
    import bench
    import scanpy_amd as sc
    from oracle import compare as cmp
    from oracle import leiden as ol

    ol.build()
    x, truth = bench.make_matrix(n, 2000, 0, structure)
    chain = bench.cpu_chain(x, 50, 15)
    conn = chain["conn"]
    print(f"graph: n={n} nnz={conn.nnz} ({structure}); CPU chain seconds {chain['seconds']}", flush=True)
    oracle = [(chain["labels"], chain["modularity"])]
    for s in range(1, 5):
        t0 = time.perf_counter()
        oracle.append(ol.leiden(conn, resolution=1.0, n_iterations=-1, seed=s))
        print(f"oracle seed {s}: {time.perf_counter() - t0:.1f} s", flush=True)
    o_truth = [cmp.ari(m, truth) for m, _ in oracle]
    o_pair = [cmp.ari(a[0], b[0]) for a, b in itertools.combinations(oracle, 2)]
    print("oracle: Q", [round(q, 5) for _, q in oracle], "ARI vs truth", [round(v, 3) for v in o_truth],
          "clusters", [int(m.max()) + 1 for m, _ in oracle], f"pairwise ARI median {np.median(o_pair):.3f} min {min(o_pair):.3f}", flush=True)

    def gpu_runs(env: dict, seeds=range(5)):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            out = []
            for s in seeds:
                a = sc.AnnData(x[:, :1])
                t0 = time.perf_counter()
                sc.tl.leiden(a, adjacency=conn, flavor="igraph", n_iterations=-1, random_state=s)
                dt = time.perf_counter() - t0
                out.append((a.obs["leiden"].cat.codes.to_numpy(), float(a.uns["leiden"]["modularity"]), dt))
            return out
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    variants = {
        "default": {},
        "lm_classes=1": {"SCAMD_LEIDEN_LM_CLASSES": "1"},
        "lm_classes=32": {"SCAMD_LEIDEN_LM_CLASSES": "32"},
        "rf_classes=32": {"SCAMD_LEIDEN_RF_CLASSES": "32"},
        "rf_classes=1": {"SCAMD_LEIDEN_RF_CLASSES": "1"},
        "lm_stop=0": {"SCAMD_LEIDEN_LM_STOP_PERMILLE": "0"},
        "lm_stop=2": {"SCAMD_LEIDEN_LM_STOP_PERMILLE": "2"},
        "lm_stop=5": {"SCAMD_LEIDEN_LM_STOP_PERMILLE": "5"},
        "lm_stop=10": {"SCAMD_LEIDEN_LM_STOP_PERMILLE": "10"},
        "lm32+rf32+stop0": {"SCAMD_LEIDEN_LM_CLASSES": "32", "SCAMD_LEIDEN_RF_CLASSES": "32", "SCAMD_LEIDEN_LM_STOP_PERMILLE": "0"},
    }
    if len(sys.argv) > 3:
        variants = {k: v for k, v in variants.items() if k in sys.argv[3].split(",")}
    report = {"n": n, "structure": structure, "oracle": {"Q": [q for _, q in oracle], "ari_truth": o_truth, "pairwise_median": float(np.median(o_pair))}}
    for name, env in variants.items():
        runs = gpu_runs(env)
        g_truth = [cmp.ari(m, truth) for m, _, _ in runs]
        g_cross = [cmp.ari(m, o[0]) for m, _, _ in runs for o in oracle]
        g_pair = [cmp.ari(a[0], b[0]) for a, b in itertools.combinations(runs, 2)]
        print(f"gpu {name:18s} Q {[round(q, 5) for _, q, _ in runs]} ARI vs truth {[round(v, 3) for v in g_truth]} (median {np.median(g_truth):.3f}; "
              f"oracle median {np.median(o_truth):.3f}) clusters {[int(m.max()) + 1 for m, _, _ in runs]} vs-oracle median {np.median(g_cross):.3f} "
              f"own pairwise median {np.median(g_pair):.3f}  {np.mean([t for _, _, t in runs]) * 1e3:.0f} ms", flush=True)
        report[name] = {"Q": [q for _, q, _ in runs], "ari_truth": g_truth, "vs_oracle_median": float(np.median(g_cross)),
                        "pairwise_median": float(np.median(g_pair)), "ms": float(np.mean([t for _, _, t in runs]) * 1e3)}
    print(json.dumps(report))


if __name__ == "__main__":
    main()
