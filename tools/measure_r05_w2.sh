#!/bin/bash
# Round-5 GPU call: the list insertion laid out in line (likely hint on its guard) against the build before, one box.
set -u
TAG="${1:-r05w2}"
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
cp /tmp/new.so $LIB
