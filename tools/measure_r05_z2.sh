#!/bin/bash
# Round-5 GPU call: fewer class sub-rounds in the sweeps of the local moving that follow a sweep with few active vertices
# (SCAMD_LEIDEN_SMALL_SWEEP_ACT / _CLASSES) -- time and modularity of Leiden alone on the three structures.
set -u
TAG="${1:-r05z2}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
for ST in none weak; do
  for K in "X=1 Y=1" "SCAMD_LEIDEN_SMALL_SWEEP_ACT=512 SCAMD_LEIDEN_SMALL_SWEEP_CLASSES=2" "SCAMD_LEIDEN_SMALL_SWEEP_ACT=4096 SCAMD_LEIDEN_SMALL_SWEEP_CLASSES=2" "SCAMD_LEIDEN_SMALL_SWEEP_ACT=4096 SCAMD_LEIDEN_SMALL_SWEEP_CLASSES=4"; do
    echo "[$ST $K] $(env $K timeout -k 5 300 python tools/leiden_only.py 1000000 $ST 2 2>&1 | grep 'leiden n=' | tail -1 | sed 's/mean.*over 2; //' | cut -c1-200)" | tee -a "$OUT/leiden_small_sweeps.log"
  done
done
for K in "X=1 Y=1" "SCAMD_LEIDEN_SMALL_SWEEP_ACT=4096 SCAMD_LEIDEN_SMALL_SWEEP_CLASSES=2"; do
  echo "[planted $K] $(env $K timeout -k 5 300 python tools/leiden_only.py 1000000 planted 3 2>&1 | grep 'leiden n=' | tail -1 | sed 's/mean.*over 3; //' | cut -c1-200)" | tee -a "$OUT/leiden_small_sweeps.log"
done
