#!/bin/bash
# Round-4 first GPU call: HEAD's first profile (kernel trace + stats of the bench command), the bench line, then the
# counter passes of the dominant kernel -- counters restricted to that kernel (--kernel-include-regex) and preceded by a
# small probe, so a profiler fault costs one short timeout and skips the rest.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/measure_r04_a.sh r04a'
set -u
TAG="${1:-r04a}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"

timeout -k 5 700 python -m pytest tests -m gpu -q -p no:faulthandler > "$OUT/pytest_gpu.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -2 "$OUT/pytest_gpu.log" | cut -c1-200

timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 < /dev/null
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"

cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_prof.log" 2>&1 < /dev/null
echo "bench prof rc=$?"
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_kernel_stats.csv" \;
cd "$R"

timeout -k 5 500 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
echo "bench rc=$?"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
    print("bench:", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()},
          "failed gates:", d.get("parity", {}).get("failed_gates"), "properties:", d.get("full_size_properties", {}).get("failed_gates"))
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY

# counter passes, the select kernel only
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
PMC3="FETCH_SIZE"
PMC4="WRITE_SIZE"
cd /tmp
timeout -k 5 150 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex 'knn_select_reg' --pmc $PMC1 -d /tmp/pmc_${TAG}_0 -o knn -- python "$R/tools/knn_only.py" 60000 1 > "$OUT/pmc0.log" 2>&1 < /dev/null
rc=$?
echo "pmc probe rc=$rc"; tail -3 "$OUT/pmc0.log" | cut -c1-300
if [ $rc -eq 0 ]; then
  i=0
  for P in "$PMC1" "$PMC2" "$PMC3" "$PMC4"; do
    i=$((i+1))
    timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex 'knn_select_reg' --pmc $P -d /tmp/pmc_${TAG}_$i -o knn -- python "$R/tools/knn_only.py" 1000000 1 > "$OUT/pmc$i.log" 2>&1 < /dev/null
    echo "pmc$i rc=$?"
    find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} "$OUT/knn_pmc$i.csv" \;
    grep "knn n=" "$OUT/pmc$i.log" | tail -1
  done
fi
cd "$R"
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
