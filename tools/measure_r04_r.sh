#!/bin/bash
# Round-4 A/B: the next cell's first tile requested under the current cell's last one (SCAMD_KNN_CELL_PRELOAD, default 1),
# with the per-block breakdown of the extended trace (prologue / pre-pass / sweeps / rest, fixed cost per cell).
#   /usr/local/graft/bin/gpurun --timeout 700 -- 'bash tools/measure_r04_r.sh r04r'
set -u
TAG="${1:-r04r}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for P in 1 0; do
  SCAMD_KNN_CELL_PRELOAD=$P timeout -k 5 150 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline_preload$P.log" 2>&1 < /dev/null
  echo "preload=$P rc=$?"; grep "^launch\|^per block\|^share\|^sweep time\|utilisation\|tiles per us" "$OUT/knn_timeline_preload$P.log" | cut -c1-330
done
for P in 1 0 1 0; do
  echo "preload=$P $(SCAMD_KNN_CELL_PRELOAD=$P timeout -k 5 150 python tools/knn_only.py 1000000 3 2>&1 | grep 'knn n=' | tail -1 | cut -c1-260)"
done
timeout -k 5 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_hard.py tests/test_gpu_knn_certificate.py -m gpu -q -p no:faulthandler -k "knn" > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -2 "$OUT/pytest_knn.log" | cut -c1-200
