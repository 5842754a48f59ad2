#!/bin/bash
# Round-5 GPU call: Leiden's host read-backs through a pinned page with one synchronisation per round trip, against the build
# before (tools/ab/libscanpy_amd_old.so): Leiden alone on the three structures, the Leiden GPU tests.
set -u
TAG="${1:-r05u}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
LIB=scanpy_amd/_lib/libscanpy_amd.so
cp $LIB /tmp/new.so
for ST in planted weak none; do
  for V in new old; do
    if [ $V = old ]; then cp tools/ab/libscanpy_amd_old.so $LIB; else cp /tmp/new.so $LIB; fi
    echo "[$ST $V] $(timeout -k 5 300 python tools/leiden_only.py 1000000 $ST 3 2>&1 | grep 'leiden n=' | tail -1 | cut -c1-110)" | tee -a "$OUT/leiden_ab.log"
  done
done
cp /tmp/new.so $LIB
timeout -k 5 900 python -m pytest tests/test_gpu_leiden.py tests/test_gpu_leiden_determinism.py tests/test_gpu_leiden_guarantees.py -m gpu -q -p no:faulthandler > "$OUT/pytest_leiden.log" 2>&1 < /dev/null
echo "leiden tests rc=$?"; tail -2 "$OUT/pytest_leiden.log" | cut -c1-300
