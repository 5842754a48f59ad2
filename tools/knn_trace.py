"""Timeline of the pruned kNN sweep: per-block start / end / tiles swept, from SCAMD_KNN_TRACE (debug dump of
knn_select_reg_kernel<.., IVF>).  Usage (GPU box): python tools/knn_trace.py [n] [structure]
Answers: how much of the launch is a tail (few blocks left running), how uneven the blocks are, tiles per microsecond
of a block in the full-occupancy phase vs alone."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def parse_trace(path):
    """the dump of SCAMD_KNN_TRACE (csrc/knn.hip, IvfArgs::trace): per block 8 x uint64 {start, end (100 MHz ticks), tiles
    swept, xcc << 32 | hw id, end of the prologue, end of the pre-pass, ticks inside the cells' sweeps, cells swept}, then
    the blocks' cells as int32 -> (uint64 [n_blocks, 8], int32 [n_blocks])"""
    raw = np.fromfile(path, dtype=np.uint8)
    nb = raw.size // 68
    if nb * 68 != raw.size:
        raise ValueError(f"{path}: {raw.size} bytes is not a whole number of 68-byte block records")
    return raw[: nb * 64].view(np.uint64).reshape(nb, 8), raw[nb * 64:].view(np.int32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    structure = sys.argv[2] if len(sys.argv) > 2 else "planted"
    import torch

    import bench
    from scanpy_amd import _kernels as K
    from scanpy_amd import _lib
    from scanpy_amd.preprocessing._pca_solver import GpuBackend, pca_fit

    x, _ = bench.make_matrix(n, 2000, 0, structure)
    backend = GpuBackend()
    res = pca_fit(backend.upload(x), 50, backend=backend)
    emb = res.scores
    no_insert = os.environ.pop("SCAMD_KNN_DEBUG_NO_INSERT", None)  # (only the traced call runs in the debug mode)
    K.knn(emb, 15)
    path = "/tmp/knn_trace.bin"
    os.environ["SCAMD_KNN_TRACE"] = path
    if no_insert:
        os.environ["SCAMD_KNN_DEBUG_NO_INSERT"] = no_insert
    try:
        K.knn(emb, 15)
    except _lib.ScamdError as e:  # SCAMD_KNN_DEBUG_NO_INSERT=1: the lists are empty, the float64 scan overflows
        print("knn raised (expected in the no-insert debug mode):", str(e)[:120])
    os.environ.pop("SCAMD_KNN_TRACE")
    lib = _lib.load()
    sel_ms, pairs = float(lib.scamd_knn_last_select_ms()), float(lib.scamd_knn_last_select_pairs())
    tr, cell = parse_trace(path)
    nb = tr.shape[0]
    t0, t1, tiles = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64), tr[:, 2].astype(np.int64)
    xcc = (tr[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0  # 100 MHz -> microseconds
    dur = t1 - t0
    span = t1.max()
    print(f"launch {sel_ms:.2f} ms (events), trace span {span / 1e3:.2f} ms, {nb} blocks, {pairs:.3e} pairs, "
          f"{np.unique(cell).size} cells")
    print(f"block duration us: min {dur.min():.0f} p10 {np.percentile(dur, 10):.0f} median {np.median(dur):.0f} "
          f"p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}; tiles per block: min {tiles.min()} median "
          f"{int(np.median(tiles))} max {tiles.max()}")
    # concurrency over time
    grid = np.linspace(0, span, 201)
    active = ((t0[None, :] <= grid[:, None]) & (t1[None, :] > grid[:, None])).sum(axis=1)
    print("active blocks at 0,5,..100% of the span:", active[::10].tolist())
    full = active.max()
    busy = np.trapezoid(active, grid) / (full * span)
    print(f"max concurrent blocks {full}; block-slot utilisation over the span {busy:.3f}")
    tail_start = grid[np.argmax(active < 0.9 * full) if (active < 0.9 * full).any() else -1]
    last = np.where(active >= 0.9 * full)[0].max()
    print(f"span with >= 90% of the slots busy ends at {grid[last] / 1e3:.2f} ms of {span / 1e3:.2f} ms "
          f"({1 - grid[last] / span:.1%} tail)")
    rate = tiles / np.maximum(dur, 1e-9)
    mid = (t0 > 0.1 * span) & (t1 < 0.6 * span)
    late = t0 > grid[last]
    print(f"tiles per us per block: steady-state blocks {np.median(rate[mid]):.3f} (n={mid.sum()}), "
          f"blocks started in the tail {np.median(rate[late]) if late.any() else float('nan'):.3f} (n={late.sum()})")
    # work per XCD
    for xc in range(8):
        m = xcc == xc
        if m.any():
            print(f"  xcc {xc}: {m.sum()} blocks, {tiles[m].sum()} tiles, last end {t1[m].max() / 1e3:.2f} ms")
    # order of work: are long blocks late?
    order = np.argsort(t0)
    q = np.array_split(order, 10)
    # where a block's time goes: prologue (query operand, order table), pre-pass + sort, the cells' sweeps, the rest
    # (stopping rule between the cells, final write)
    pro = (tr[:, 4].astype(np.int64) - tr[:, 0].astype(np.int64)) / 100.0
    pre = (tr[:, 5].astype(np.int64) - tr[:, 4].astype(np.int64)) / 100.0
    swp = tr[:, 6].astype(np.int64) / 100.0
    ncell = tr[:, 7].astype(np.int64)
    rest = dur - pro - pre - swp
    print(f"per block (median / mean, us): prologue {np.median(pro):.1f} / {pro.mean():.1f}, pre-pass + sort {np.median(pre):.1f} / {pre.mean():.1f}, "
          f"sweeps {np.median(swp):.1f} / {swp.mean():.1f}, rest {np.median(rest):.1f} / {rest.mean():.1f}; cells swept {np.median(ncell):.0f} / {ncell.mean():.1f} "
          f"(tiles per cell {tiles.sum() / max(ncell.sum(), 1):.1f})")
    print(f"share of the blocks' time: prologue {pro.sum() / dur.sum():.3f}, pre-pass {pre.sum() / dur.sum():.3f}, sweeps {swp.sum() / dur.sum():.3f}, "
          f"rest {rest.sum() / dur.sum():.3f}; tiles per us inside the sweeps {tiles.sum() / swp.sum():.3f}")
    # a sweep's fixed cost: least squares of the per-block sweep time on (cells, tiles)
    A = np.stack([ncell.astype(np.float64), tiles.astype(np.float64)], axis=1)
    coef, *_ = np.linalg.lstsq(A, swp, rcond=None)
    print(f"sweep time ~ {coef[0]:.2f} us per cell + {coef[1]:.3f} us per tile ({1.0 / coef[1]:.3f} tiles per us once a sweep runs)")
    print("median tiles per block by start-time decile:", [int(np.median(tiles[i])) for i in q])
    print("median duration (us) by start-time decile:", [int(np.median(dur[i])) for i in q])


if __name__ == "__main__":
    main()
