"""Probe (not a test): how much does vertex locality buy the Leiden kernels?  Runs the path at 1M to get the fuzzy graph,
times Leiden on it as is and after renumbering the vertices so that communities are contiguous."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from scanpy_amd import _kernels as K  # noqa: E402
from scanpy_amd._pipeline import run_path  # noqa: E402
from scanpy_amd.datasets import synthetic_planted  # noqa: E402
from scanpy_amd.preprocessing._pca_solver import GpuBackend  # noqa: E402


def timed_leiden(ip, ix, w, n, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        lab, q, nc = K.leiden(ip, ix, w, n)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3, lab, q, nc


def permute(ip, ix, w, n, order):
    """order[new] = old.  CSR of the renumbered graph."""
    inv = torch.empty_like(order)
    inv[order] = torch.arange(n, device=order.device, dtype=order.dtype)
    deg = (ip[1:] - ip[:-1])
    new_deg = deg[order]
    nip = torch.zeros(n + 1, dtype=torch.int64, device=ip.device)
    nip[1:] = torch.cumsum(new_deg, 0)
    rows_new = torch.repeat_interleave(torch.arange(n, device=ip.device), new_deg)
    pos_in_row = torch.arange(int(nip[-1]), device=ip.device) - nip[:-1][rows_new]
    src = ip[:-1][order][rows_new] + pos_in_row
    return nip, inv[ix[src].long()].to(torch.int32).contiguous(), w[src].contiguous()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    x, _ = synthetic_planted(n, 2000, seed=0)
    be = GpuBackend()
    res = run_path(be.upload(x), n, backend=be)
    ip, ix, w = res.conn_indptr, res.conn_indices, res.conn_data
    ms, lab, q, nc = timed_leiden(ip, ix, w, n)
    print(f"original order : {ms:.1f} ms  Q={q:.6f} nc={nc}")
    order = torch.argsort(lab.long(), stable=True)
    nip, nix, nw = permute(ip, ix, w, n, order)
    ms2, lab2, q2, nc2 = timed_leiden(nip, nix, nw, n)
    print(f"community order: {ms2:.1f} ms  Q={q2:.6f} nc={nc2}")
    rnd = torch.randperm(n, device=ip.device)
    nip, nix, nw = permute(ip, ix, w, n, rnd)
    ms3, _, q3, nc3 = timed_leiden(nip, nix, nw, n)
    print(f"random order   : {ms3:.1f} ms  Q={q3:.6f} nc={nc3}")


if __name__ == "__main__":
    main()
