"""-m gpu: the row-sharded path with 2 and 3 ranks sharing cuda:0 (real kernels, gloo collectives through host memory)
against the single-rank run -- SURVEY.md 8(e) on hardware, minus RCCL itself (see tests/dist_gpu_worker.py)."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def _launch(world: int, tmp: Path, n: int, g: int, k: int):
    init = tmp / f"gpu_init_{world}"
    procs = [subprocess.Popen([sys.executable, str(HERE / "dist_gpu_worker.py"), str(r), str(world), str(init), str(tmp),
                               str(n), str(g), str(k)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [dict(np.load(tmp / f"gpu_rank{r}_of{world}.npz")) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_match_single_rank(tmp_path, world):
    n, g, k = 70001, 400, 30  # n >= 65536: the cell-pruned kNN sweep, query-sharded; odd n: ragged shards
    if os.environ.get("SCAMD_TESTS_ON_EMULATOR") == "1":  # (host-emulated kernels, tests/emu: a size they finish;
        n, g = 6001, 200                                   #  run with SCAMD_KNN_IVF=1 to keep the pruned sweep)
    one = _launch(1, tmp_path, n, g, k)[0]
    many = _launch(world, tmp_path, n, g, k)
    # PCA: the int64 Gram matrix is additive over shards -> the SAME model on every rank and for every world size
    for r in many:
        np.testing.assert_array_equal(r["components"], one["components"])
        np.testing.assert_array_equal(r["variance"], one["variance"])
    scores = np.concatenate([r["scores"] for r in many], axis=0)
    np.testing.assert_array_equal(scores, one["scores"])
    # kNN: every rank answers its own rows against all candidates -> identical lists
    idx = np.concatenate([r["knn_idx"] for r in many], axis=0)
    dist = np.concatenate([r["knn_dist"] for r in many], axis=0)
    np.testing.assert_array_equal(idx, one["knn_idx"])
    np.testing.assert_array_equal(dist, one["knn_dist"])
    # graph + Leiden on rank 0 only, labels broadcast: identical to the single-rank labels
    assert bool(many[0]["has_graph"]) and not any(bool(r["has_graph"]) for r in many[1:])
    # the graph itself: per-rank membership strengths + all-to-all of the directed edges + per-rank merge give the rows of
    # the single-device fuzzy set bit for bit (same float32 expression on the same operands)
    for key in ("conn_indptr", "conn_indices", "conn_data"):
        np.testing.assert_array_equal(many[0][key], one[key], err_msg=key)
    for r in many:
        np.testing.assert_array_equal(r["labels"], one["labels"])
        assert int(r["nc"]) == int(one["nc"]) and float(r["q"]) == float(one["q"])


def test_bench_line_of_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), both ranks on cuda:0
    with gloo collectives (SCAMD_BENCH_ONE_DEVICE=1: RCCL refuses two ranks on one device): the N > 1 code path of the
    bench -- sharded generation, barriers, max-over-ranks timing, the per-rank stage table -- must produce its JSON line."""
    import json

    env = dict(os.environ, SCAMD_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(HERE.parent / "bench.py"), "--gpus", "2", "--n-obs", "70000", "--n-vars", "500",
           "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(HERE.parent))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    mg = line["multi_gpu"]
    assert len(mg["per_rank_stage_ms"]) == 2 and all(r["knn"] > 0 and r["pca"] > 0 for r in mg["per_rank_stage_ms"])
    assert mg["per_rank_stage_ms"][0]["leiden"] > 0 and mg["per_rank_stage_ms"][1]["leiden"] == 0  # rank 0 only
    assert 0 < mg["rank0_only_share_of_step"] < 1 and mg["collective_bytes"]["embedding_all_gather_f32"] == 4.0 * 70000 * 50
