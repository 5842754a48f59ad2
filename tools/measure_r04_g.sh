#!/bin/bash
# Round-4 seventh GPU call: pre-pass over more cells (A/B on one box), GPU transfer test, full GPU suite.
set -u
TAG="${1:-r04g}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for knob in "" "SCAMD_KNN_PREPASS_CELLS=2" "SCAMD_KNN_PREPASS_CELLS=3" "SCAMD_KNN_PREPASS_CELLS=4" "SCAMD_KNN_PREPASS_CELLS=6" "SCAMD_KNN_PREPASS_CELLS=8" "SCAMD_KNN_PREPASS_CELLS=16" ""; do
  env $knob timeout -k 5 200 python tools/knn_only.py 1000000 4 > "$OUT/knob.log" 2>&1 < /dev/null
  echo "knob [$knob]: $(grep 'knn n=' "$OUT/knob.log" | sort -t' ' -k7 -n | head -1 | sed 's/.*select/select/' | cut -c1-75) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
done
for st in weak; do
  for knob in "" "SCAMD_KNN_PREPASS_CELLS=4"; do
    env $knob timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 $st > "$OUT/knob.log" 2>&1 < /dev/null
    echo "$st [$knob]: $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*select/select/' | cut -c1-75) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
  done
done
timeout -k 5 900 python -m pytest tests -m gpu -q -p no:faulthandler > "$OUT/pytest_gpu.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log" | cut -c1-200
