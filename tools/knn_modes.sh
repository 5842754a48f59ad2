#!/bin/bash
# timing of the kNN select kernel variants (SCAMD_KNN_MODE) on the GPU box
TAG=${1:-knn_modes}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "knn or mfma" > $OUT/pytest_knn.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest_knn.log
for M in ${MODES:-1 0 3 2}; do
  echo "MODE $M" | tee -a $OUT/modes.log
  SCAMD_KNN_MODE=$M timeout 300 python tools/knn_only.py 1000000 2 2>&1 | grep knn | tee -a $OUT/modes.log
done
