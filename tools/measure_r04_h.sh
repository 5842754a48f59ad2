#!/bin/bash
# Round-4 eighth GPU call: the early-stop rule of the local moving (SCAMD_LEIDEN_LM_STOP_PERMILLE) -- quality on the weak
# 100k sample against the oracle's seeds, time on the three 1M structures.
set -u
TAG="${1:-r04h}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout -k 5 900 python tools/leiden_weak_probe.py 100000 weak "default,lm_stop=0,lm_stop=2,lm_stop=5,lm_stop=10" > "$OUT/weak_probe.log" 2>&1 < /dev/null
echo "weak probe rc=$?"; grep -E "^(oracle:|gpu )" "$OUT/weak_probe.log" | cut -c1-330
for st in planted weak none; do
  for v in 20 10 5 2 0; do
    SCAMD_LEIDEN_LM_STOP_PERMILLE=$v timeout -k 5 300 python tools/leiden_only.py 1000000 $st 3 > "$OUT/lknob.log" 2>&1 < /dev/null
    echo "leiden $st [stop=$v]: $(tail -1 "$OUT/lknob.log" | cut -c1-160)"
  done
done
