#!/bin/bash
# Round-5 GPU call: A/B of two builds of libscanpy_amd.so on ONE box (tools/ab/libscanpy_amd_old.so = the build before the
# change, copied there by hand; *.so is git-ignored but travels with gpurun) -- the kNN select kernel alone, then its
# per-block timeline.
set -u
TAG="${1:-r05p}"
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
    echo "[$V] $(timeout -k 5 300 python tools/knn_only.py 1000000 4 2>&1 | grep 'knn n=' | tail -2 | cut -c1-120 | tr '\n' '|')" | tee -a "$OUT/knn_ab.log"
  done
done
cp /tmp/new.so $LIB
timeout -k 5 300 python tools/knn_trace.py 1000000 planted > "$OUT/knn_timeline.log" 2>&1 < /dev/null; echo "timeline rc=$?"
grep -v Warning "$OUT/knn_timeline.log" | tail -24
