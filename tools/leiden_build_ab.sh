#!/bin/bash
# A/B of BUILDS of the library on one box: tools/leiden_build_ab.sh <variant> ... (tools/ab/libscanpy_amd_<variant>.so), Leiden alone
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
cp scanpy_amd/_lib/libscanpy_amd.so /tmp/lib_keep.so
for V in "$@"; do
  cp tools/ab/libscanpy_amd_$V.so scanpy_amd/_lib/libscanpy_amd.so
  for ST in ${STRUCTS:-planted weak}; do echo "== $V $ST"; timeout -k 5 280 python tools/leiden_only.py 1000000 $ST 3 2>&1 | grep "leiden n=" | cut -c1-110; done
done
cp /tmp/lib_keep.so scanpy_amd/_lib/libscanpy_amd.so
