#!/bin/bash
# Round-4 fourth GPU call (the third lost its kNN legs to a deleted line): certificate + kNN tests, kNN knob sweep, per-level
# Leiden trace, Leiden x5 under SCAMD_GUARD, Leiden knobs on planted / none, Leiden determinism tests.
set -u
TAG="${1:-r04d}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout -k 5 900 python -m pytest tests/test_gpu_knn_certificate.py -q -s -p no:faulthandler > "$OUT/pytest_cert.log" 2>&1 < /dev/null
echo "cert rc=$?"; grep -E "cert_scale|differing|passed|failed|Error" "$OUT/pytest_cert.log" | cut -c1-220 | tail -12
timeout -k 5 600 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_kernels.py tests/test_gpu_leiden.py tests/test_gpu_leiden_determinism.py -q -p no:faulthandler > "$OUT/pytest_knn_leiden.log" 2>&1 < /dev/null
echo "knn + leiden tests rc=$?"; tail -2 "$OUT/pytest_knn_leiden.log" | cut -c1-200
for knob in "" "SCAMD_KNN_THR_MARGIN=8" "SCAMD_KNN_THR_MARGIN=6" "SCAMD_KNN_PREPASS_TILES=64" "SCAMD_KNN_PREPASS_TILES=16" "SCAMD_KNN_XCD_ORDER=0"; do
  env $knob timeout -k 5 200 python tools/knn_only.py 1000000 3 > "$OUT/knob.log" 2>&1 < /dev/null
  echo "knob [$knob]: $(grep 'knn n=' "$OUT/knob.log" | tail -1 | cut -c30-140)"
done
SCAMD_LEIDEN_DEBUG=1 timeout -k 5 300 python tools/leiden_only.py 1000000 planted 1 > "$OUT/leiden_trace_planted.log" 2>&1 < /dev/null
echo "leiden trace rc=$?"; grep -E "^\[leiden\] (level|small|iteration)" "$OUT/leiden_trace_planted.log" | tail -34 | cut -c1-170
SCAMD_GUARD=1 timeout -k 5 400 python tools/leiden_only.py 1000000 planted 5 > "$OUT/leiden_guard.log" 2>&1 < /dev/null
echo "leiden x5 under SCAMD_GUARD rc=$?"; tail -1 "$OUT/leiden_guard.log" | cut -c1-200
for st in planted none; do
  for knob in "" "SCAMD_LEIDEN_LM_STOP_PERMILLE=0" "SCAMD_LEIDEN_RF_CLASSES=32"; do
    env $knob timeout -k 5 300 python tools/leiden_only.py 1000000 $st 3 > "$OUT/lknob.log" 2>&1 < /dev/null
    echo "leiden $st [$knob]: $(tail -1 "$OUT/lknob.log" | cut -c1-160)"
  done
done
timeout -k 5 300 python bench.py --steps 10 --warmup 3 --cpu-sizes 0 --no-side --h2h-reps 0 > "$OUT/bench_short.json" 2> "$OUT/bench_short.err" < /dev/null
echo "bench rc=$?"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.loads([l for l in open(out + "/bench_short.json") if l.startswith("{")][-1])
    print("bench:", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()},
          "properties:", d.get("full_size_properties", {}).get("failed_gates"), "labels", d["result"]["labels_sha"], d.get("structure_none", {}).get("labels_sha"),
          d.get("structure_none", {}).get("ms_per_step"))
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
