#!/bin/bash
# Round-5 GPU call: padding query slots silenced (threshold -1e30) against the build before (tools/ab/libscanpy_amd_old.so)
# on ONE box: select kernel alone (planted, weak), the timeline, the kNN GPU tests.
set -u
TAG="${1:-r05s}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
LIB=scanpy_amd/_lib/libscanpy_amd.so
cp $LIB /tmp/new.so
for ROUND in 1 2; do
  for V in new old; do
    if [ $V = old ]; then cp tools/ab/libscanpy_amd_old.so $LIB; else cp /tmp/new.so $LIB; fi
    echo "[$V] $(timeout -k 5 300 python tools/knn_only.py 1000000 4 2>&1 | grep 'knn n=' | tail -3 | sed 's/.*select \([0-9.]*\) ms.*fallback=\(.*\)/\1 ms fb=\2/' | tr '\n' '|')" | tee -a "$OUT/knn_ab.log"
  done
done
for V in new old; do
  if [ $V = old ]; then cp tools/ab/libscanpy_amd_old.so $LIB; else cp /tmp/new.so $LIB; fi
  echo "[weak $V] $(timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 weak 2>&1 | grep 'knn n=' | tail -1 | sed 's/.*select \([0-9.]*\) ms.*fallback=\(.*\)/\1 ms fb=\2/')" | tee -a "$OUT/knn_ab.log"
done
cp /tmp/new.so $LIB
timeout -k 5 300 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline.log" 2>&1 < /dev/null; echo "timeline rc=$?"
cp /tmp/knn_trace.bin "$OUT/knn_trace.bin"
grep -v Warning "$OUT/knn_timeline.log" | tail -22
timeout -k 5 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_knn_approx.py tests/test_gpu_knn_certificate.py tests/test_gpu_parity_hard.py -m gpu -q -x -p no:faulthandler > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "knn tests rc=$?"; tail -2 "$OUT/pytest_knn.log" | cut -c1-300
