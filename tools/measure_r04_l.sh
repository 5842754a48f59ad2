#!/bin/bash
# Round-4: two minima per lane in the threshold pre-pass (A/B on one box), faster fallback scan / block order (kernel trace).
set -u
TAG="${1:-r04l}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for knob in "" "SCAMD_KNN_PREPASS_MIN2=0" "" "SCAMD_KNN_PREPASS_MIN2=0" "SCAMD_KNN_THR_MARGIN=10" "SCAMD_KNN_THR_MARGIN=6"; do
  env $knob timeout -k 5 200 python tools/knn_only.py 1000000 4 > "$OUT/knob.log" 2>&1 < /dev/null
  echo "knob [$knob]: $(grep 'knn n=' "$OUT/knob.log" | sort -t' ' -k7 -n | head -1 | sed 's/.*select/select/' | cut -c1-75) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
done
for st in weak none; do
  for knob in "" "SCAMD_KNN_PREPASS_MIN2=0"; do
    env $knob timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 $st > "$OUT/knob.log" 2>&1 < /dev/null
    echo "$st [$knob]: $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*select/select/' | cut -c1-75) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
  done
done
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kn_${TAG} -o knn -- python "$R/tools/knn_only.py" 1000000 4 > "$OUT/knn_only_prof.log" 2>&1 < /dev/null
find /tmp/kn_${TAG} -name '*kernel_stats.csv' -exec cp {} "$OUT/knn_only_kernel_stats.csv" \;
python - "$OUT/knn_only_kernel_stats.csv" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if any(t in r["Name"] for t in ("knn_", "ivf_")):
        tot += float(r["TotalDurationNs"])
        if float(r["AverageNs"]) > 3e4: print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
print("   kNN kernels total per call:", tot / 5e6, "ms")
PY
cd "$R"
timeout -k 5 600 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_kernels.py tests/test_gpu_knn_certificate.py -q -k "knn" -p no:faulthandler > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "knn tests rc=$?"; tail -2 "$OUT/pytest_knn.log" | cut -c1-200
