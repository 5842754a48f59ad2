#!/bin/bash
# local-moving class sub-rounds per sweep (8 / 4 / 2) on the 1M graphs, five seeds each: time, Q, communities
#   bash tools/leiden_classes_ab.sh <tag> [structure ...]
TAG="${1:-r06_classes}"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
for ST in ${@:-weak none planted}; do
  for C in 8 4 2; do echo "== $ST lm_classes $C"; SCAMD_LEIDEN_LM_CLASSES=$C timeout -k 5 400 python tools/oracle_iters_probe.py 1000000 $ST none 0,1,2,3,4 2>&1 | grep "^gpu seed" | cut -c1-150; done
done | tee "$OUT/classes.log"
