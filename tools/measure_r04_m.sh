#!/bin/bash
# Round-4: Leiden with the re-queue of sub-round c and the decisions of sub-round c + 1 in one launch (A/B on one box),
# Leiden + determinism tests, the new kNN variant test.
set -u
TAG="${1:-r04m}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for st in planted none; do
  for knob in "" "SCAMD_LEIDEN_FUSE=0" "" "SCAMD_LEIDEN_FUSE=0"; do
    env $knob timeout -k 5 300 python tools/leiden_only.py 1000000 $st 3 > "$OUT/lknob.log" 2>&1 < /dev/null
    echo "leiden $st [$knob]: $(tail -1 "$OUT/lknob.log" | cut -c1-160)"
  done
done
timeout -k 5 900 python -m pytest tests/test_gpu_leiden.py tests/test_gpu_leiden_determinism.py tests/test_gpu_metrics.py tests/test_gpu_parity_hard.py -q -p no:faulthandler > "$OUT/pytest_leiden.log" 2>&1 < /dev/null
echo "tests rc=$?"; tail -2 "$OUT/pytest_leiden.log" | cut -c1-200
timeout -k 5 300 python bench.py --steps 10 --warmup 3 --cpu-sizes 0 --no-side --h2h-reps 0 > "$OUT/bench_short.json" 2> "$OUT/bench_short.err" < /dev/null
python - "$OUT/bench_short.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("bench", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "labels", d["result"]["labels_sha"],
      "properties", d.get("full_size_properties", {}).get("failed_gates"), "none", round(d["structure_none"]["ms_per_step"], 1), d["structure_none"]["labels_sha"])
PY
