#!/bin/bash
# Round-4, last GPU seconds: the Leiden paper's guarantees on the GPU's stable partitions at full size.
set -u
TAG="${1:-r04v}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$R"
timeout -k 5 50 python tools/leiden_guarantees_probe.py 300000 none weak planted 2>&1 | grep -v "^/opt\|Warning" | tee "$OUT/leiden_guarantees.log" | cut -c1-400
timeout -k 5 40 python tools/leiden_guarantees_probe.py 1000000 planted 2>&1 | grep -v "^/opt\|Warning" | tee -a "$OUT/leiden_guarantees.log" | cut -c1-400
