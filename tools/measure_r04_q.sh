#!/bin/bash
# Round-4 diagnosis of the select kernel (bf16x3 engine): per-block timeline of the pruned sweep as it runs, and the same
# launch with every survivor dropped (SCAMD_KNN_DEBUG_NO_INSERT=1: wrong lists, same tiles) -- what the insertions cost.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/measure_r04_q.sh r04q'
set -u
TAG="${1:-r04q}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout -k 5 150 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline.log" 2>&1 < /dev/null
echo "timeline rc=$?"; grep -v "^/opt\|Warning" "$OUT/knn_timeline.log" | head -30 | cut -c1-300
SCAMD_KNN_DEBUG_NO_INSERT=1 timeout -k 5 150 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline_no_insert.log" 2>&1 < /dev/null
echo "no-insert rc=$?"; grep -v "^/opt\|Warning" "$OUT/knn_timeline_no_insert.log" | head -30 | cut -c1-300
