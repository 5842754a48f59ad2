"""Multi-resolution Leiden replicas: resolutions round-robin over the ranks, every rank ends with all
labelings, identical to the single-process sweep.  CPU: gloo + stand-in kernels; GPU: real kernels, ranks share cuda:0."""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent


def _launch(world: int, tmp: Path, mode: str):
    init = tmp / f"mr_init_{mode}_{world}"
    procs = [subprocess.Popen([sys.executable, str(HERE / "multires_worker.py"), str(r), str(world), str(init), str(tmp), mode])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [dict(np.load(tmp / f"multires_{mode}_rank{r}_of{world}.npz")) for r in range(world)]


def _check(tmp_path, mode):
    one = _launch(1, tmp_path, mode)[0]
    two = _launch(2, tmp_path, mode)
    keys = [k for k in one if not k.startswith("q_")]
    assert len(keys) == 5
    n_clusters = [int(one[k].max()) + 1 for k in keys]
    assert n_clusters == sorted(n_clusters) and n_clusters[0] < n_clusters[-1]  # finer with the resolution
    for r in two:
        for k in keys:
            np.testing.assert_array_equal(r[k], one[k])
            assert float(r[f"q_{k}"]) == float(one[f"q_{k}"])


def test_resolution_owner_round_robin():
    from scanpy_amd.tools._leiden_multires import resolution_owner

    assert [resolution_owner(i, 3) for i in range(7)] == [0, 1, 2, 0, 1, 2, 0]


def test_multires_two_ranks_cpu(tmp_path):
    _check(tmp_path, "cpu")


@pytest.mark.gpu
def test_multires_two_ranks_one_gpu(tmp_path):
    _check(tmp_path, "gpu")
