#!/bin/bash
# Round-5 GPU call: per-sweep trace of the local moving (active vertices / moves per sweep and level) on the weak and the
# structure-less 1M graphs -- how much of an iteration is spent in sweeps whose active lists hold a handful of vertices.
set -u
TAG="${1:-r05o}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
for ST in ${STRUCTURES:-none weak}; do
  SCAMD_LEIDEN_DEBUG=1 timeout -k 5 400 python tools/leiden_only.py 1000000 $ST 1 > "$OUT/lm_trace_$ST.out" 2> "$OUT/lm_trace_$ST.err" < /dev/null
  echo "$ST rc=$? lines $(wc -l < "$OUT/lm_trace_$ST.err")"
  grep '\[leiden\]' "$OUT/lm_trace_$ST.err" | gzip > "$OUT/lm_trace_$ST.log.gz"
  rm -f "$OUT/lm_trace_$ST.err"
done
