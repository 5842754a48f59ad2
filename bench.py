#!/usr/bin/env python3
"""bench.py -- cells/sec through pca + neighbors + leiden on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path (PCA 50 comps -> exact kNN k=15 -> umap connectivities ->
Leiden res 1.0) over the synthetic planted-cluster log-normal CSR (1M cells x 2k genes, ~5 % nnz: BASELINE
configs[2], the 1x MI355X roofline configuration; the 1M cells are row-sharded over the N ranks for N > 1,
i.e. strong scaling of configs[3]).  The CSR shard is resident in HBM before the timed region starts; the
timed region ends with the labels on the device.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      knn_select_reg_kernel (FP32 MFMA): achieved = 2 * 50 flop per evaluated (query, candidate) pair
                (scamd_knn_last_select_pairs; the exact cell-pruned search skips provably empty cells) / its HIP-event
                duration (scamd_knn_last_select_ms), peak = 157.3 TFLOP/s (MI355X_MICROARCH.md).
  cpu_baseline  the reference's CPU call chain (sklearn PCA arpack + sklearn brute kNN = reference calls; oracle
                fuzzy set + oracle Leiden) on a bounded sample of the same matrix, on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-obs", type=int, default=1_000_000)
    ap.add_argument("--n-vars", type=int, default=2000)
    ap.add_argument("--n-comps", type=int, default=50)
    ap.add_argument("--n-neighbors", type=int, default=15)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="cells in the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def cpu_baseline(n_sample: int, n_vars: int, n_comps: int, k: int, seed: int) -> dict:
    """Reference CPU chain on the first `n_sample` cells of the same synthetic matrix (rank 0, N=1 only)."""
    import numpy as np

    from oracle import connectivities as oc
    from oracle import knn as oknn
    from oracle import leiden as ol
    from oracle import pca as opca
    from scanpy_amd.datasets import synthetic_planted

    ol.build()
    x, _ = synthetic_planted(n_sample, n_vars, seed=seed)
    t0 = time.perf_counter()
    ref = opca.pca_reference(x, n_comps)
    t1 = time.perf_counter()
    idx, dist, _ = oknn.knn_sklearn(ref["X_pca"].astype(np.float32), k, n_jobs=-1)
    t2 = time.perf_counter()
    conn, _, _ = oc.fuzzy_simplicial_set(idx, dist, n_sample, k)
    t3 = time.perf_counter()
    ol.leiden(conn, resolution=1.0, n_iterations=-1, seed=0)
    t4 = time.perf_counter()
    total = t4 - t0
    return {
        "value": n_sample / total,
        "unit": "cells/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": (f"first {n_sample} cells x {n_vars} genes of the same synthetic CSR; sklearn PCA(arpack) "
                   f"{t1 - t0:.2f}s + sklearn brute kNN(n_jobs=-1) {t2 - t1:.2f}s (the reference's own calls) + oracle "
                   f"fuzzy_simplicial_set {t3 - t2:.2f}s + oracle Leiden {t4 - t3:.2f}s; brute kNN is O(n^2), so the "
                   "CPU rate at the full 1M cells is far lower than at this sample size"),
        "seconds": total,
    }


def _profiled_traffic(mode: str):
    """HBM-side bytes per launch of the roofline kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE are separate profiling runs, they cannot be taken inside this process); None if the committed passes
    are of a different sweep mode than the one that ran (or absent)."""
    f = ROOT / "profiles" / "knn_select_traffic.json"
    try:
        d = json.loads(f.read_text())
        return d["bytes_per_launch"] if d.get("mode") == mode else None
    except (OSError, KeyError, ValueError):
        return None


def _baseline_config(n: int, g: int, world: int) -> str:
    """which BASELINE.json `configs` entry the run corresponds to (the label is informational)"""
    if (n, g) == (1_000_000, 2000):
        return "BASELINE configs[2]" if world == 1 else f"BASELINE configs[3] over {world} GPUs"
    if (n, g) == (100_000, 2000):
        return "BASELINE configs[1]"
    if (n, g) == (10_000_000, 4000):
        return "BASELINE configs[4] sizes, exact kNN, single resolution"
    return "custom size"


def upstream_chain(handle, reps: int = 3) -> dict:
    """SURVEY 8(f).2 rows, measured beside the path (NOT part of `value`): the device passes of
    normalize_total(1e4) -> log1p -> highly_variable_genes('seurat') statistics -> scale(zero_center=False) on the
    resident CSR, each against the HBM roofline with its algorithmic bytes (csrc/preprocess.hip header)."""
    import torch

    from scanpy_amd import _kernels as K

    ip, ix, dt, n, g = handle[:5]
    nnz = dt.numel()
    passes = [
        ("row_sums", 4 * nnz + 8 * (n + 1) + 4 * n),
        ("row_divide", 8 * nnz + 8 * (n + 1) + 4 * n),
        ("log1p", 8 * nnz),
        ("col_stats_expm1", 8 * nnz + 8 * (n + 1) + 24 * g),
        ("scale_csr", 12 * nnz + 8 * (n + 1) + 8 * g),
    ]
    best = [float("inf")] * len(passes)
    for _ in range(reps):
        d = dt.clone()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(passes) + 1)]
        ev[0].record()
        sums = K.pp_row_sums(ip, ix, d, n)
        ev[1].record()
        K.pp_row_divide_(ip, d, n, sums / 1e4)
        ev[2].record()
        K.pp_log1p_(d)
        ev[3].record()
        s, sq, _ = K.pp_col_stats(ip, ix, d, n, g, expm1_scale=1.0, count_positive=False)
        ev[4].record()
        K.pp_scale_csr_(ip, ix, d, n, torch.ones(g, dtype=torch.float64, device=d.device), max_value=10.0)
        ev[5].record()
        torch.cuda.synchronize()
        for i in range(len(passes)):
            best[i] = min(best[i], ev[i].elapsed_time(ev[i + 1]))
    out = {"note": "device passes of pp.normalize_total / log1p / highly_variable_genes / scale on the same CSR; "
                   "best of %d; outside `value`" % reps, "peak_GBps": 8000.0, "passes": {}}
    for (name, nbytes), ms in zip(passes, best):
        gbps = nbytes / (ms * 1e-3) / 1e9
        out["passes"][name] = {"ms": ms, "algorithmic_bytes": nbytes, "GBps": gbps, "frac": gbps / 8000.0}
    out["total_ms"] = sum(best)
    return out


def umap_layout(res, n: int, n_epochs: int = 200) -> dict:
    """SURVEY 8(f).1 row, measured beside the path (NOT part of `value`): the UMAP layout kernel on the fuzzy graph the
    timed path just produced (resident), random start, `n_epochs` synchronous epochs."""
    import torch

    from scanpy_amd import _kernels as K
    from scanpy_amd.tools._umap import find_ab_params, prune_and_schedule_device

    ip, ix, w, eps = prune_and_schedule_device(res.conn_indptr, res.conn_indices, res.conn_data, n, n_epochs)
    a, b = find_ab_params(1.0, 0.5)
    gen = torch.Generator(device="cpu").manual_seed(0)
    y = (torch.rand((n, 2), generator=gen) * 10.0).to(w.device).contiguous()
    fired = float(torch.floor(float(n_epochs - 1) / eps[eps > 0].to(torch.float64)).sum())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.umap_optimize_(ip, ix, eps, n, y, n_epochs=n_epochs, a=a, b=b, seed=0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nnz = int(ix.numel())
    # per epoch: 16 B of index + schedule per stored sample; a fired sample gathers one neighbour and ~5 negatives (8 B
    # each at n_components = 2) and rewrites 8 B of schedule; plus the embedding in and out
    bytes_total = n_epochs * (16.0 * nnz + 16.0 * n) + fired * (8.0 + 6 * 8.0)
    return {"note": "scamd_umap_optimize_f32 on the path's own fuzzy graph, random start; outside `value`",
            "n_epochs": n_epochs, "stored_samples": nnz, "fired_samples": fired, "ms": ms, "ms_per_epoch": ms / n_epochs,
            "gathers_per_s": fired * 6.0 / (ms * 1e-3), "algorithmic_GBps": bytes_total / (ms * 1e-3) / 1e9,
            "frac_of_8TBps": bytes_total / (ms * 1e-3) / 1e9 / 8000.0, "finite": bool(torch.isfinite(y).all())}


def main() -> None:
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # SCAMD_BENCH_ONE_DEVICE=1: validation mode, all ranks share cuda:0 and the collectives go through gloo (RCCL refuses
    # two ranks on one device); the numbers of such a run are meaningless, the code path is the multi-rank one
    one_device = os.environ.get("SCAMD_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist

    from scanpy_amd import _lib
    from scanpy_amd._pipeline import run_path, shard_bounds
    from scanpy_amd.datasets import synthetic_planted
    from scanpy_amd.preprocessing._pca_solver import GpuBackend, NoComm, TorchDistComm

    comm = NoComm()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        comm = TorchDistComm()

    n = args.n_obs
    lo, hi = shard_bounds(n, world, rank)
    t_gen = time.perf_counter()
    x, _ = synthetic_planted(n, args.n_vars, seed=args.seed, row_range=(lo, hi))
    t_gen = time.perf_counter() - t_gen
    backend = GpuBackend()
    t_h2d = time.perf_counter()
    handle = backend.upload(x)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter() - t_h2d
    nnz_local = x.nnz
    del x

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    kw = dict(comm=comm, backend=backend, n_comps=args.n_comps, n_neighbors=args.n_neighbors, resolution=1.0,
              n_iterations=-1, seed=0)
    for _ in range(args.warmup):
        run_path(handle, n, **kw)
    lib = _lib.load()
    select_ms, select_pairs, stage_acc, res = [], [], {}, None
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = run_path(handle, n, timing=True, **kw)
        select_ms.append(float(lib.scamd_knn_last_select_ms()))
        select_pairs.append(float(lib.scamd_knn_last_select_pairs()))
        for kname, v in res.stage_ms.items():
            stage_acc[kname] = stage_acc.get(kname, 0.0) + v
    sync_all()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        if one_device:
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    value = n * args.steps / elapsed

    if rank == 0:
        sel = sum(select_ms) / max(len(select_ms), 1)
        n_query = hi - lo
        pairs = sum(select_pairs) / max(len(select_pairs), 1)  # (query, candidate) pairs the kernel evaluated
        brute_pairs = float(n_query) * float(n)
        # algorithmic flop of the launch = 2 * d flop per EVALUATED pair: the exact cell-pruned search skips the
        # cells that provably hold no neighbour, what it does evaluate runs on the FP32 MFMA pipe
        flops = 2.0 * pairs * args.n_comps
        achieved = flops / (sel * 1e-3) / 1e12 if sel > 0 else None
        peak = 157.3
        out = {
            "metric": "cells/sec through pca+neighbors+leiden, 1M x 2k CSR" if (n, args.n_vars) == (1_000_000, 2000) else f"cells/sec through pca+neighbors+leiden, {n} x {args.n_vars} CSR",
            "value": value,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"synthetic planted log-normal CSR {n} cells x {args.n_vars} genes (~5% nnz), PCA {args.n_comps} "
                             f"(exact Gram + dense eigensolve, arpack accuracy) + exact kNN k={args.n_neighbors} (cell-pruned brute force) + umap "
                             "connectivities + Leiden res=1.0 n_iterations=-1 (" + _baseline_config(n, args.n_vars, world) + ")"),
                "n_obs": n,
                "n_vars": args.n_vars,
                "nnz_per_rank": int(nnz_local),
                "parallelism": f"cells row-sharded x{world}; leiden on rank 0",
            },
            "roofline": {
                "kernel": "knn_select_reg_kernel<25,64,3> (v_mfma_f32_32x32x2_f32), exact cell-pruned sweep",
                "bound": "mfma",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": (achieved / peak) if achieved else None,
                # the committed PMC passes are of the single-GPU 1M x 1M launch: not quoted for any other shape
                "traffic": (_profiled_traffic("ivf" if 0 < pairs < brute_pairs else "brute")
                            if (n, args.n_comps, world) == (1_000_000, 50, 1) else None),
                "launch_ms": sel,
                "algorithmic_flop_per_launch": flops,
                "pairs_evaluated_fraction": pairs / brute_pairs if brute_pairs > 0 else None,
                "brute_force_equivalent_tflops": 2.0 * brute_pairs * args.n_comps / (sel * 1e-3) / 1e12 if sel > 0 else None,
            },
            "stage_ms_per_step": {kname: v / max(args.steps, 1) for kname, v in stage_acc.items()},
            "result": {"n_communities": res.n_communities, "modularity": res.modularity, **res.info},
            "setup_s": {"generate": t_gen, "h2d": t_h2d},
        }
        if world == 1:
            out["upstream_chain"] = upstream_chain(handle)
            out["umap_layout"] = umap_layout(res, n)
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, n), args.n_vars, args.n_comps, args.n_neighbors, args.seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
