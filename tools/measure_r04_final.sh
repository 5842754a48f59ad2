#!/bin/bash
# Round-4 final measurement: counters of the select kernel (-> profiles/knn_select_traffic.json), GPU suite, smoke, the
# bench line (20 steps, CPU legs and parity included), kernel stats of the same command.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/measure_r04_final.sh r04z'
set -u
TAG="${1:-r04z}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
cd /tmp
i=0
for P in "$PMC1" "SKIP" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  [ "$P" = "SKIP" ] && continue
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex 'knn_select_reg' --pmc $P -d /tmp/pmc_${TAG}_$i -o knn -- python "$R/tools/knn_only.py" 1000000 1 > "$OUT/pmc$i.log" 2>&1 < /dev/null
  echo "pmc$i rc=$? $(grep 'knn n=' "$OUT/pmc$i.log" | tail -1 | sed 's/.*select/select/' | cut -c1-40)"
  find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} "$OUT/knn_pmc$i.csv" \;
done
cd "$R"
test -s "$OUT/knn_pmc3.csv" && test -s "$OUT/knn_pmc4.csv" && python tools/make_traffic_json.py "$OUT/knn_pmc3.csv" "$OUT/knn_pmc4.csv" > /dev/null && cp profiles/knn_select_traffic.json "$OUT/knn_select_traffic.json"
python - "$OUT" <<'PY'
import collections, csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/knn_pmc*.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "knn_select_reg" in r.get("Kernel_Name", "")]
    if not rows:
        print(f, "no rows"); continue
    gmax = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(float)
    for r in rows:
        if int(r["Grid_Size"]) == gmax:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[-1], gmax, {k: f"{v:.4g}" for k, v in acc.items()})
PY
timeout -k 5 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
echo "bench rc=$?"; tail -2 "$OUT/bench.err" | cut -c1-300
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_prof.log" 2>&1 < /dev/null
echo "bench prof rc=$?"
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_kernel_stats.csv" \;
cd "$R"
timeout -k 5 900 python -m pytest tests -m gpu -q -p no:faulthandler > "$OUT/pytest_gpu.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -2 "$OUT/pytest_gpu.log" | cut -c1-200
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 < /dev/null
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
cd "$R"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
    print("bench:", round(d["value"]), "cells/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()})
    print("h2h", d.get("value_host_to_host"), d["host_to_host"]["best"])
    sn = d["structure_none"]; print("none", sn["ms_per_step"], sn["stage_ms"], sn["n_communities"], sn["labels_sha"])
    r = d["roofline"]; print({k: r[k] for k in ("engine", "achieved", "peak", "frac", "launch_ms", "traffic", "algorithmic_bytes_per_launch", "pairs_evaluated_fraction")})
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"))
    print("failed gates", d["parity"]["failed_gates"], "weak", json.dumps(d["parity"]["weak"])[:900])
    print("properties", d["full_size_properties"]["failed_gates"], d["full_size_properties"].get("enforced"), "labels", d["result"]["labels_sha"])
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY

# informational A/Bs with whatever GPU time is left (every run bounded; nothing below feeds the numbers above)
for K in "SCAMD_KNN_THR_MARGIN=6" "SCAMD_KNN_THR_MARGIN=10" "SCAMD_KNN_PREPASS_TILES=32" "SCAMD_KNN_PREPASS_TILES=8" "SCAMD_KNN_IVF_WPS=2" ""; do
  echo "[$K] $(env $K timeout -k 5 60 python tools/knn_only.py 1000000 3 2>&1 | grep 'knn n=' | tail -1 | cut -c1-200)" | tee -a "$OUT/knn_knobs.log"
done
