#!/bin/bash
# Round-5 fifth GPU call: coarse rows tiered by work (member entries) as well as by table size -- Leiden alone on the three
# structures, A/B of the wave-tier threshold on the weak graph; Leiden / pipeline GPU tests.
set -u
TAG="${1:-r05e}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
for ST in planted weak none; do
  echo "[] $(timeout -k 5 300 python tools/leiden_only.py 1000000 $ST 3 2>&1 | grep 'leiden n=' | tail -1 | cut -c1-330)" | tee -a "$OUT/leiden_work_tiers.log"
done
for K in "SCAMD_LEIDEN_AGG_WAVE_WORK=100000000 SCAMD_LEIDEN_AGG_MID_WORK=100000000" "SCAMD_LEIDEN_AGG_WAVE_WORK=512" "SCAMD_LEIDEN_AGG_WAVE_WORK=8192" "SCAMD_LEIDEN_AGG_MID_WORK=16384" "SCAMD_LEIDEN_AGG_MID_WORK=262144"; do
  echo "[$K] $(env $K timeout -k 5 300 python tools/leiden_only.py 1000000 weak 3 2>&1 | grep 'leiden n=' | tail -1 | cut -c1-100)" | tee -a "$OUT/leiden_work_tiers.log"
done
for K in "SCAMD_LEIDEN_AGG_WAVE_WORK=100000000 SCAMD_LEIDEN_AGG_MID_WORK=100000000" "SCAMD_LEIDEN_AGG_WAVE_WORK=512" "SCAMD_LEIDEN_AGG_WAVE_WORK=8192"; do
  echo "[$K] $(env $K timeout -k 5 300 python tools/leiden_only.py 1000000 planted 5 2>&1 | grep 'leiden n=' | tail -1 | cut -c1-100)" | tee -a "$OUT/leiden_work_tiers.log"
done
timeout -k 5 900 python -m pytest tests/test_gpu_leiden.py tests/test_gpu_leiden_guarantees.py tests/test_gpu_leiden_determinism.py tests/test_gpu_pipeline.py -m gpu -q -p no:faulthandler > "$OUT/pytest_leiden.log" 2>&1 < /dev/null
echo "leiden tests rc=$?"; tail -3 "$OUT/pytest_leiden.log" | cut -c1-300
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o leiden -- python "$R/tools/leiden_only.py" 1000000 weak 1 > "$OUT/leiden_weak_prof.log" 2>&1 < /dev/null
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/leiden_weak_kernel_stats.csv" \;
python - "$OUT/leiden_weak_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.1f} ms {100*float(r['TotalDurationNs'])/tot:5.1f} % {int(r['Calls']):6d} calls avg {float(r['AverageNs'])/1e3:8.1f} us max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:60]}")
PY
