#!/bin/bash
# Round-5 GPU call: persistent launch of the pruned kNN sweep (work taken off per-XCD queues, stealing when a queue is dry)
# against the launch of one workgroup per slot (SCAMD_KNN_PERSISTENT=0) and against the previous build
# (tools/ab/libscanpy_amd_old.so), on ONE box; then the per-block timeline and the kNN GPU tests.
set -u
TAG="${1:-r05q}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
LIB=scanpy_amd/_lib/libscanpy_amd.so
cp $LIB /tmp/new.so
for ROUND in 1 2; do
  for V in "new" "new SCAMD_KNN_PERSISTENT=0" "new SCAMD_KNN_PERSISTENT=512" "old"; do
    set -- $V
    if [ $1 = old ]; then cp tools/ab/libscanpy_amd_old.so $LIB; else cp /tmp/new.so $LIB; fi
    echo "[$V] $(env ${2:-X=1} timeout -k 5 300 python tools/knn_only.py 1000000 4 2>&1 | grep 'knn n=' | tail -3 | sed 's/.*select \([0-9.]*\) ms.*fallback=\(.*\)/\1 ms fb=\2/' | tr '\n' '|')" | tee -a "$OUT/knn_ab.log"
  done
done
cp /tmp/new.so $LIB
for ST in weak; do
  for K in X=1 SCAMD_KNN_PERSISTENT=0; do
    echo "[$ST $K] $(env $K timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 $ST 2>&1 | grep 'knn n=' | tail -1 | sed 's/.*select \([0-9.]*\) ms.*fallback=\(.*\)/\1 ms fb=\2/')" | tee -a "$OUT/knn_ab.log"
  done
done
timeout -k 5 300 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline.log" 2>&1 < /dev/null; echo "timeline rc=$?"
grep -v Warning "$OUT/knn_timeline.log" | tail -22
timeout -k 5 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_knn_approx.py tests/test_gpu_knn_certificate.py tests/test_gpu_parity_hard.py -m gpu -q -x -p no:faulthandler > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "knn tests rc=$?"; tail -2 "$OUT/pytest_knn.log" | cut -c1-300
