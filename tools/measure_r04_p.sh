#!/bin/bash
# Round-4 A/B in one box: Leiden with the packed refinement records (HEAD's library) against the library built from the
# commit before (scanpy_amd/_lib/prev/libscanpy_amd.so, swapped in on the box's scratch copy), then the Leiden tests and a
# short bench line (labels_sha must not move: every field of the records is an integer).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/measure_r04_p.sh r04p'
set -u
TAG="${1:-r04p}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
L="$R/scanpy_amd/_lib"
for S in planted none; do
  timeout -k 5 120 python tools/leiden_only.py 1000000 $S 5 2>&1 | grep "^leiden" | sed "s/^/new  /"
done
if [ -f "$L/prev/libscanpy_amd.so" ]; then
  cp "$L/libscanpy_amd.so" /tmp/new.so && cp "$L/prev/libscanpy_amd.so" "$L/libscanpy_amd.so"
  for S in planted none; do
    timeout -k 5 120 python tools/leiden_only.py 1000000 $S 5 2>&1 | grep "^leiden" | sed "s/^/prev /"
  done
  cp /tmp/new.so "$L/libscanpy_amd.so"
fi
for S in planted; do
  timeout -k 5 120 python tools/leiden_only.py 1000000 $S 5 2>&1 | grep "^leiden" | sed "s/^/new  /"
done
timeout -k 5 300 python -m pytest tests/test_gpu_leiden.py tests/test_gpu_leiden_determinism.py tests/test_gpu_pipeline.py -m gpu -q -p no:faulthandler > "$OUT/pytest_leiden.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -2 "$OUT/pytest_leiden.log" | cut -c1-200
timeout -k 5 300 python bench.py --steps 10 --warmup 3 --cpu-sizes 0 --no-side --h2h-reps 0 > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("bench", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, d["result"]["labels_sha"],
          "none", d.get("structure_none", {}).get("labels_sha"), d.get("structure_none", {}).get("ms_per_step"), "properties", d["full_size_properties"]["failed_gates"])
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
