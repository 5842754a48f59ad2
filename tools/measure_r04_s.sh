#!/bin/bash
# Round-4 A/B: the bf16 engine's tiles by LDS-DMA into a ring of three buffers (two requests in flight across the barrier,
# running on into the next cell), against the same with every cell starting cold (SCAMD_KNN_CELL_PRELOAD=0) and the build
# cut for three blocks per CU (SCAMD_KNN_IVF_WPS=3: 5 spilled registers now that the staging registers are gone); the
# per-block breakdown; the kNN tests on the hardware (a read placed before its wait would show there, not on a CPU).
#   /usr/local/graft/bin/gpurun --timeout 700 -- 'bash tools/measure_r04_s.sh r04s'
set -u
TAG="${1:-r04s}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for K in "" "SCAMD_KNN_CELL_PRELOAD=0" "SCAMD_KNN_IVF_WPS=3" ""; do
  echo "[$K] $(env $K timeout -k 5 150 python tools/knn_only.py 1000000 3 2>&1 | grep 'knn n=' | tail -1 | cut -c1-260)"
done
timeout -k 5 150 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline.log" 2>&1 < /dev/null
echo "timeline rc=$?"; grep "^launch\|^per block\|^share\|^sweep time\|utilisation\|tiles per us" "$OUT/knn_timeline.log" | cut -c1-330
SCAMD_KNN_DEBUG_NO_INSERT=1 timeout -k 5 150 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline_no_insert.log" 2>&1 < /dev/null
echo "no-insert rc=$?"; grep "^launch\|^sweep time\|tiles per us" "$OUT/knn_timeline_no_insert.log" | cut -c1-330
timeout -k 5 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_hard.py tests/test_gpu_knn_certificate.py -m gpu -q -p no:faulthandler -k "knn" > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -2 "$OUT/pytest_knn.log" | cut -c1-200
