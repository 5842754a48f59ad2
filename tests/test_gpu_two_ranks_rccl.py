"""Self-enabling check of the RCCL path: skipped on a one-GPU lease, otherwise `bench.py --gpus 2` is
launched exactly as the driver launches it (torch.distributed.run, one rank per GPU, backend "nccl" = RCCL over xGMI) on
a 200k-cell matrix and must reproduce the one-rank run bit for bit: same labels (sha1), same modularity, same number of
communities -- the int64 Gram all-reduce, the embedding all-gather, the all-to-all of the directed graph edges and the
gather of the CSR rows to rank 0 all sit between the two.  (One-device variants of the same path, gloo collectives staged
through the host: tests/test_gpu_sharded_one_device.py; CPU, world size 2: tests/test_sharded_gloo.py.)"""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
ARGS = ["--steps", "1", "--warmup", "0", "--n-obs", "200000", "--cpu-sizes", "", "--no-noise-variant", "--no-side",
        "--h2h-reps", "0", "--no-verify"]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(cmd) -> dict:
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=str(ROOT), env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_ranks_over_rccl_reproduce_one_rank():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on one node (the RCCL path has no one-device form)")
    one = _bench([sys.executable, "bench.py", "--gpus", "1", *ARGS])
    two = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                  "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", *ARGS])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    r1, r2 = one["result"], two["result"]
    print("1 rank:", r1["n_communities"], r1["modularity"], r1["labels_sha"], "| 2 ranks:", r2["n_communities"], r2["modularity"],
          r2["labels_sha"], "ms/step", one["ms_per_step"], two["ms_per_step"])
    assert r1["labels_sha"] == r2["labels_sha"]
    assert r1["modularity"] == r2["modularity"] and r1["n_communities"] == r2["n_communities"]
