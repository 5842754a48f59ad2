#!/bin/bash
# Round-5 GPU call: coarse rows split into parts for several workgroups -- Leiden alone with and without, three structures; tests.
set -u
TAG="${1:-r05l}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
for ST in planted weak none; do
  for K in "" "SCAMD_LEIDEN_AGG_SPLIT_WORK=2000000000" "SCAMD_LEIDEN_AGG_SPLIT_CHUNK=32768 SCAMD_LEIDEN_AGG_SPLIT_WORK=65536"; do
    echo "[$ST $K] $(env $K timeout -k 5 300 python tools/leiden_only.py 1000000 $ST 3 2>&1 | grep 'leiden n=' | tail -1 | cut -c1-110)" | tee -a "$OUT/leiden_split_rows.log"
  done
done
timeout -k 5 900 python -m pytest tests/test_gpu_leiden.py tests/test_gpu_leiden_determinism.py tests/test_gpu_leiden_guarantees.py -m gpu -q -p no:faulthandler > "$OUT/pytest_leiden.log" 2>&1 < /dev/null
echo "leiden tests rc=$?"; tail -2 "$OUT/pytest_leiden.log" | cut -c1-300
