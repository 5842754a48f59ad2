#!/bin/bash
# What would a coarse (hi.hi only) first stage of the select kernel's sweep cost?  Launch time with all survivors dropped
# (SCAMD_KNN_DEBUG_NO_INSERT=1), the product library (12 MFMAs per sub-tile) against a PROBE build of knn.hip
# (-DSCAMD_KNN_PROBE_HH: 4 MFMAs; tools/ab/libscanpy_amd_probe_hh.so, built by hand: see DESIGN.md section 8), per structure.
#   bash tools/knn_coarse_probe.sh <tag>        (on the GPU box: the scratch copy of the repository is modified)
TAG="${1:-r06_coarse}"; R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
cp scanpy_amd/_lib/libscanpy_amd.so /tmp/libscanpy_amd_product.so
{
timeout 300 python tools/knn_only.py 1000000 1 2>&1 | grep "knn n=" | cut -c1-120
for LIB in product probe_hh; do
  [ $LIB = probe_hh ] && cp tools/ab/libscanpy_amd_probe_hh.so scanpy_amd/_lib/libscanpy_amd.so
  for ST in weak planted; do
    echo "== $LIB $ST NO_INSERT=1"
    SCAMD_KNN_DEBUG_NO_INSERT=1 timeout -k 5 300 python tools/knn_trace.py 1000000 $ST 2>&1 | grep "^launch\|^block duration" | cut -c1-200
  done
done
cp /tmp/libscanpy_amd_product.so scanpy_amd/_lib/libscanpy_amd.so
} | tee "$OUT/coarse_probe.log"
