#!/bin/bash
# What would a coarse (hi.hi only) first stage of the select kernel's sweep cost?  Launch time with all survivors dropped,
# 12 MFMAs per sub-tile (NO_INSERT=1) against 4 (NO_INSERT=2), per structure.   bash tools/knn_coarse_probe.sh <tag>
TAG="${1:-r06_coarse}"; R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
for ST in weak planted; do
  for V in 1 2; do
    echo "== $ST NO_INSERT=$V"
    SCAMD_KNN_DEBUG_NO_INSERT=$V timeout -k 5 300 python tools/knn_trace.py 1000000 $ST 2>&1 | grep "^launch\|^block duration\|raised" | cut -c1-200
  done
done | tee "$OUT/coarse_probe.log"
