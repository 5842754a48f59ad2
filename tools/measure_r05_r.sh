#!/bin/bash
# Round-5 GPU call: raw per-block trace of the persistent pruned sweep (for the analysis of the launch's last millisecond)
set -u
TAG="${1:-r05r}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
timeout -k 5 300 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline.log" 2>&1 < /dev/null; echo "timeline rc=$?"
cp /tmp/knn_trace.bin "$OUT/knn_trace.bin"
