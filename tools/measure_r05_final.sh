#!/bin/bash
# Round-5 final measurement of a state: counters of the select kernel (-> profiles/knn_select_traffic.json, regenerated in
# the same call as the bench line that quotes it), GPU suite, smoke, the bench line (20 steps; CPU legs, parity, both
# structure variants, approximate-search curves), kernel stats of the same path.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/measure_r05_final.sh r05z'
set -u
TAG="${1:-r05z}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
cd /tmp
i=0
for P in "$PMC1" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex 'knn_select_reg' --pmc $P -d /tmp/pmc_${TAG}_$i -o knn -- python "$R/tools/knn_only.py" 1000000 1 > "$OUT/pmc$i.log" 2>&1 < /dev/null
  echo "pmc$i rc=$? $(grep 'knn n=' "$OUT/pmc$i.log" | tail -1 | sed 's/.*select/select/' | cut -c1-40)"
  find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} "$OUT/knn_pmc$i.csv" \;
done
cd "$R"
test -s "$OUT/knn_pmc2.csv" && test -s "$OUT/knn_pmc3.csv" && python tools/make_traffic_json.py "$OUT/knn_pmc2.csv" "$OUT/knn_pmc3.csv" > /dev/null && cp profiles/knn_select_traffic.json "$OUT/knn_select_traffic.json"
timeout -k 5 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
echo "bench rc=$?"; tail -2 "$OUT/bench.err" | cut -c1-300
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_prof.log" 2>&1 < /dev/null
echo "bench prof rc=$?"
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_kernel_stats.csv" \;
cd "$R"
timeout -k 5 1200 python -m pytest tests -m gpu -q -p no:faulthandler > "$OUT/pytest_gpu.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -2 "$OUT/pytest_gpu.log" | cut -c1-200
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 < /dev/null
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
# the realistic case as the timed workload (overlapping programmes: the exact search evaluates every pair), and the driver's own
# invocation (no flags) under `time`
timeout -k 5 600 python bench.py --structure weak --steps 5 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_weak.json" 2> "$OUT/bench_weak.err" < /dev/null
echo "bench weak rc=$? $(python -c "import json,sys; d=json.loads([l for l in open('$OUT/bench_weak.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, 'roofline', round(d['roofline']['frac'],3), d['full_size_properties']['failed_gates'])" 2>&1 | tail -1)"
( time timeout -k 5 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" < /dev/null ) 2> "$OUT/bench_default.time"
echo "bench default rc=$? $(grep real "$OUT/bench_default.time")"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
    print("bench:", round(d["value"]), "cells/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()})
    print("h2h", d.get("value_host_to_host"), d["host_to_host"]["best"])
    print("leiden", {k: v for k, v in d["leiden"].items() if k != "note"})
    print("knn_approx", json.dumps(d.get("knn_approx", {}).get("runs")))
    for st in ("none", "weak"):
        sn = d["structure_" + st]
        print(st, round(sn["ms_per_step"], 1), {k: round(v, 1) for k, v in sn["stage_ms"].items()}, sn["n_communities"], sn["labels_sha"], sn["leiden_guarantees"])
    r = d["roofline"]; print({k: r[k] for k in ("engine", "achieved", "peak", "frac", "launch_ms", "traffic", "algorithmic_bytes_per_launch", "pairs_evaluated_fraction")})
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"))
    print("failed gates", d["parity"]["failed_gates"], "variants", d.get("variant_failed_gates"), "properties", d["full_size_properties"]["failed_gates"], "labels", d["result"]["labels_sha"])
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
