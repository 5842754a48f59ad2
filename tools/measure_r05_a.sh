#!/bin/bash
# Round-5 first GPU call: the new tests first (approximate kNN, exact Leiden guarantees), the whole GPU suite, smoke, the
# bench line (20 steps; structure_none + structure_weak, knn_approx curves, Leiden block), kernel stats of the same path.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/measure_r05_a.sh r05a'
set -u
TAG="${1:-r05a}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout -k 5 600 python -m pytest tests/test_gpu_knn_approx.py tests/test_gpu_leiden_guarantees.py -m gpu -q -s -p no:faulthandler > "$OUT/pytest_new.log" 2>&1 < /dev/null
echo "new tests rc=$?"; grep -E "recall|improving|passed|failed|Error" "$OUT/pytest_new.log" | cut -c1-400 | tail -30
timeout -k 5 900 python -m pytest tests -m gpu -q -p no:faulthandler > "$OUT/pytest_gpu.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log" | cut -c1-300
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 < /dev/null
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout -k 5 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
echo "bench rc=$?"; tail -2 "$OUT/bench.err" | cut -c1-400
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_prof.log" 2>&1 < /dev/null
echo "bench prof rc=$?"
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_kernel_stats.csv" \;
cd "$R"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
    print("bench:", round(d["value"]), "cells/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()})
    print("h2h", d.get("value_host_to_host"), d["host_to_host"]["best"])
    print("leiden", d["leiden"])
    print("knn_approx", json.dumps(d.get("knn_approx", {}).get("runs")))
    for st in ("none", "weak"):
        sn = d["structure_" + st]
        print(st, round(sn["ms_per_step"], 1), {k: round(v, 1) for k, v in sn["stage_ms"].items()}, sn["n_communities"], sn["labels_sha"], sn["leiden_guarantees"])
        print("   leiden", sn["leiden"]); print("   approx", json.dumps(sn.get("knn_approx", {}).get("runs")))
    r = d["roofline"]; print({k: r[k] for k in ("engine", "achieved", "peak", "frac", "launch_ms", "traffic", "pairs_evaluated_fraction")})
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"))
    print("failed gates", d["parity"]["failed_gates"], "variants", d.get("variant_failed_gates"))
    print("properties", d["full_size_properties"]["failed_gates"], d["full_size_properties"]["leiden"], "labels", d["result"]["labels_sha"])
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
