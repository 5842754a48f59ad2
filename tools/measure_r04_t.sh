#!/bin/bash
# Round-4 last GPU minute: the per-block breakdown of the shipped sweep (three blocks per CU) with and without insertions.
set -u
TAG="${1:-r04t}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$R"
timeout -k 5 45 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline.log" 2>&1 < /dev/null
echo "timeline rc=$?"; grep "^launch\|^per block\|^share\|^sweep time\|utilisation\|tiles per us\|tail" "$OUT/knn_timeline.log" | cut -c1-330
SCAMD_KNN_DEBUG_NO_INSERT=1 timeout -k 5 45 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline_no_insert.log" 2>&1 < /dev/null
echo "no-insert rc=$?"; grep "^launch\|^per block\|^share\|^sweep time\|utilisation\|tiles per us" "$OUT/knn_timeline_no_insert.log" | cut -c1-330
