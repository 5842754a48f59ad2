#!/bin/bash
# Round-5 seventh GPU call: kernel stats of Leiden alone on the structure-less 1M graph (where the time is now).
set -u
TAG="${1:-r05g}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o leiden -- python "$R/tools/leiden_only.py" 1000000 none 1 > "$OUT/leiden_none_prof.log" 2>&1 < /dev/null
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/leiden_none_kernel_stats.csv" \;
python - "$OUT/leiden_none_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total", tot / 1e6, "ms over 3 Leiden calls + one pass of the path")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.1f} ms {100*float(r['TotalDurationNs'])/tot:5.1f} % {int(r['Calls']):6d} calls avg {float(r['AverageNs'])/1e3:8.1f} us max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:60]}")
PY
cd "$R"
SCAMD_LEIDEN_DEBUG=1 timeout -k 5 300 python tools/leiden_only.py 1000000 none 1 2> "$OUT/leiden_none_trace.log" | tail -1 | cut -c1-100
grep -E "level [0-9]+ n=|aggregate [0-9.]+ ms|small levels [0-9.]+ ms|iteration [0-9]+:" "$OUT/leiden_none_trace.log" | sed -n '1,60p' | cut -c1-200
