#!/bin/bash
# Round-4: deferred insertion (per-query LDS queues) against the direct insertion of rounds 1-3, one box; kNN tests.
set -u
TAG="${1:-r04j}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for knob in "" "SCAMD_KNN_QUEUE=0" "" "SCAMD_KNN_QUEUE=0" "SCAMD_KNN_THR_MARGIN=10" "SCAMD_KNN_THR_MARGIN=6" "SCAMD_KNN_ASSIGN_MFMA=0"; do
  env $knob timeout -k 5 200 python tools/knn_only.py 1000000 4 > "$OUT/knob.log" 2>&1 < /dev/null
  echo "knob [$knob]: $(grep 'knn n=' "$OUT/knob.log" | sort -t' ' -k7 -n | head -1 | sed 's/.*select/select/' | cut -c1-75) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
done
for st in weak none; do
  for knob in "" "SCAMD_KNN_QUEUE=0"; do
    env $knob timeout -k 5 300 python tools/knn_only.py 1000000 2 50 15 $st > "$OUT/knob.log" 2>&1 < /dev/null
    echo "$st [$knob]: $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*select/select/' | cut -c1-75) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
  done
done
for knob in "" "SCAMD_KNN_IVF_WPS=3"; do
  env $knob timeout -k 5 300 python tools/knn_only.py 1000000 3 32 > "$OUT/knob.log" 2>&1 < /dev/null
  echo "d=32 float32 engine [$knob]: $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*select/select/' | cut -c1-75) $(grep 'knn n=' "$OUT/knob.log" | tail -1 | sed 's/.*fallback/fallback/')"
done
timeout -k 5 900 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_kernels.py tests/test_gpu_knn_certificate.py tests/test_gpu_sharded_one_device.py -q -p no:faulthandler > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "tests rc=$?"; tail -2 "$OUT/pytest_knn.log" | cut -c1-200
timeout -k 5 300 python bench.py --steps 10 --warmup 3 --cpu-sizes 0 --no-side --h2h-reps 0 > "$OUT/bench_short.json" 2> "$OUT/bench_short.err" < /dev/null
python - "$OUT/bench_short.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("bench", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "labels", d["result"]["labels_sha"],
      "properties", d.get("full_size_properties", {}).get("failed_gates"), "frac", d["roofline"]["frac"], "none", d["structure_none"]["ms_per_step"], d["structure_none"]["stage_ms"], d["structure_none"]["labels_sha"])
PY
