#!/bin/bash
# Round-4 fifth GPU call: kNN knobs A/B on ONE box (boxes differ by 5-10 %), whole-call wall times of the kNN stage.
set -u
TAG="${1:-r04e}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for knob in "" "SCAMD_KNN_XCD_ORDER=0" "SCAMD_KNN_IVF_WPS=3" "SCAMD_KNN_THR_MARGIN=8" "SCAMD_KNN_THR_MARGIN=6" "SCAMD_KNN_THR_MARGIN=4" "SCAMD_KNN_THR_MARGIN=6 SCAMD_KNN_PREPASS_TILES=64" "SCAMD_KNN_THR_MARGIN=6 SCAMD_KNN_IVF_WPS=3" ""; do
  env $knob timeout -k 5 200 python tools/knn_only.py 1000000 4 > "$OUT/knob.log" 2>&1 < /dev/null
  echo "knob [$knob]: $(grep 'knn n=' "$OUT/knob.log" | sort -t' ' -k7 -n | head -1 | sed 's/.*select/select/' | cut -c1-60) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
done
for st in weak none; do
  for knob in "" "SCAMD_KNN_THR_MARGIN=6"; do
    env $knob timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 $st > "$OUT/knob.log" 2>&1 < /dev/null
    echo "$st [$knob]: $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*select/select/' | cut -c1-60) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
  done
done
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_prof.log" 2>&1 < /dev/null
echo "bench prof rc=$?"
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_kernel_stats.csv" \;
cd "$R"
python - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/bench_kernel_stats.csv")))
calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
ld = [r for r in rows if "ld_" in r["Name"] or "scan_" in r["Name"]]
print("per pass: launches", calls / 4, "kernel ms", tot / 4e6, "leiden launches", sum(int(r["Calls"]) for r in ld) / 4, "leiden kernel ms", sum(float(r["TotalDurationNs"]) for r in ld) / 4e6)
for r in rows[:12]:
    print(f"   {r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/4e6:8.3f} ms/pass")
PY
