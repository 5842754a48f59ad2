#!/bin/bash
# kernel stats of the Gram entry alone: tools/gram_prof.sh <tag> [ENV=VAL ...] -> gpurun_out/<tag>/gram_kernel_stats.csv + a summary
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp; env "$@" timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gprof_$TAG -o gram -- python "$R/tools/gram_only.py" > "$OUT/gram_prof.log" 2>&1 < /dev/null )
f=$(find /tmp/gprof_$TAG -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/gram_kernel_stats.csv"; python - "$OUT/gram_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{float(r['AverageNs']) / 1e3:9.1f} us x {r['Calls']:>3}  {r['Name'][:70]}")
PY
fi
grep "gram n=" "$OUT/gram_prof.log"
