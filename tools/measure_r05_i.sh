#!/bin/bash
# Round-5 ninth GPU call: kernel stats of Leiden alone on the weak / structure-less / planted graphs after the aggregation changes.
set -u
TAG="${1:-r05i}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
for ST in ${STRUCTURES:-weak none planted}; do
  cd /tmp
  timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$ST -o leiden -- python "$R/tools/leiden_only.py" 1000000 $ST 1 > "$OUT/leiden_${ST}_prof.log" 2>&1 < /dev/null
  find /tmp/prof_${TAG}_$ST -name '*kernel_stats.csv' -exec cp {} "$OUT/leiden_${ST}_kernel_stats.csv" \;
  python - "$OUT/leiden_${ST}_kernel_stats.csv" $ST <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[2], "total", round(tot / 1e6, 1), "ms over 3 Leiden calls + one pass of the path")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.1f} ms {100*float(r['TotalDurationNs'])/tot:5.1f} % {int(r['Calls']):6d} calls avg {float(r['AverageNs'])/1e3:8.1f} us max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:60]}")
PY
done
