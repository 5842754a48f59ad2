#!/bin/bash
# select-kernel time with / without the coarse first stage (SCAMD_KNN_COARSE=0 / 1 / unset = the host's choice), per structure
TAG="${1:-r06_coarse_ab}"; R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
{
for ST in weak none planted; do
  for V in 0 1 auto; do
    echo "== $ST coarse=$V"
    if [ $V = auto ]; then timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 $ST 2>&1 | grep "knn n=" | cut -c1-230
    else SCAMD_KNN_COARSE=$V timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 $ST 2>&1 | grep "knn n=" | cut -c1-230; fi
  done
done
} | tee "$OUT/coarse_ab.log"
