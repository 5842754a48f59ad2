#!/bin/bash
# select-kernel time of the coarse first stage: off / on, and the two forms of its pipeline (product = pipelined; the A/B build
# tools/ab/libscanpy_amd_coarse_steplocal.so = test + refine inside the step), per structure.  Modifies the box's scratch copy.
TAG="${1:-r06_coarse_ab}"; R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
cp scanpy_amd/_lib/libscanpy_amd.so /tmp/libscanpy_amd_product.so
one() { SCAMD_KNN_COARSE=$2 timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 $1 2>&1 | grep "knn n=" | tail -1 | cut -c1-110; }
{
for ST in weak none planted; do
  echo "== $ST plain";            one $ST 0
  echo "== $ST coarse pipelined"; one $ST 1
  if [ -s tools/ab/libscanpy_amd_coarse_steplocal.so ]; then
    cp tools/ab/libscanpy_amd_coarse_steplocal.so scanpy_amd/_lib/libscanpy_amd.so
    echo "== $ST coarse step-local"; one $ST 1
    cp /tmp/libscanpy_amd_product.so scanpy_amd/_lib/libscanpy_amd.so
  fi
done
} | tee "$OUT/coarse_ab.log"
