#!/bin/bash
# Round-5 GPU call: the select kernel at HEAD with every survivor dropped (SCAMD_KNN_DEBUG_NO_INSERT=1 through tools/knn_trace.py:
# wrong results by design, the MFMA-side ceiling of the launch) next to the normal launch -- what the list insertions cost now.
set -u
TAG="${1:-r05ni}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
timeout -k 5 200 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline.log" 2>&1 < /dev/null; echo "timeline rc=$?"
SCAMD_KNN_DEBUG_NO_INSERT=1 timeout -k 5 200 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline_no_insert.log" 2>&1 < /dev/null; echo "no-insert rc=$?"
grep -v Warning "$OUT/knn_timeline.log" | grep -E "^launch|utilisation|per block|share|sweep time"
grep -v Warning "$OUT/knn_timeline_no_insert.log" | grep -E "^launch|utilisation|per block|share|sweep time"
