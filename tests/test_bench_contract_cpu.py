"""The bench line the driver parses: the committed result of the last measured run (profiles/r*_bench_1M.json, written
by `python bench.py` on an MI355X) carries every field of the contract with consistent values."""
from __future__ import annotations

import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_committed_bench_line_has_the_contract_fields():
    latest = sorted((ROOT / "profiles").glob("r*_bench_1M.json"))[-1]  # profiles are named per round: take the newest
    d = json.loads(latest.read_text().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "cells/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    assert "model" not in d["config"]
    # value = cells per second over the timed steps
    assert abs(d["value"] - d["config"]["n_obs"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "cells/s" and c["sample"]
    assert abs(sum(d["stage_ms_per_step"].values()) - d["ms_per_step"]) / d["ms_per_step"] < 0.05


def test_bench_default_arguments():
    import importlib.util
    import sys

    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
        a = mod.parse_args()
    finally:
        sys.argv = argv
    assert (a.gpus, a.n_obs, a.n_vars, a.n_comps, a.n_neighbors) == (1, 1_000_000, 2000, 50, 15)
    assert a.steps >= 1 and a.warmup >= 0
    assert mod._baseline_config(1_000_000, 2000, 1) == "BASELINE configs[2]"
    assert "configs[3]" in mod._baseline_config(1_000_000, 2000, 8)
