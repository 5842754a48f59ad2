#!/bin/bash
# Round-4 sixth GPU call: scores SpMM A/B (LDS-staged vs wave-per-row) inside the bench, then BASELINE configs[4] sizes
# (10M x 4k on one GPU) with the full-size properties enforced, then the same path once under SCAMD_GUARD.
set -u
TAG="${1:-r04f}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for v in 1 0 1; do
  SCAMD_SPMM_LDS=$v timeout -k 5 300 python bench.py --steps 10 --warmup 3 --cpu-sizes 0 --no-side --h2h-reps 0 --no-noise-variant > "$OUT/bench_spmm$v.json" 2> "$OUT/bench_spmm$v.err" < /dev/null
  python - "$OUT/bench_spmm$v.json" $v <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("SPMM_LDS=" + sys.argv[2], round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "labels", d["result"]["labels_sha"],
      "properties", d.get("full_size_properties", {}).get("failed_gates"), "scores sha", d["result"].get("scores_sha"))
PY
done
timeout -k 5 1200 python bench.py --n-obs 10000000 --n-vars 4000 --steps 2 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" < /dev/null
echo "10M x 4k rc=$?"; tail -3 "$OUT/bench_c5.err" | cut -c1-300
python - "$OUT/bench_c5.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("10M x 4k:", round(d["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, d["result"],
          "properties failed:", d.get("full_size_properties", {}).get("failed_gates"), {k: v for k, v in d.get("full_size_properties", {}).items() if not isinstance(v, (dict, list))})
except Exception as exc:  # noqa: BLE001
    print("no line:", exc)
PY
SCAMD_GUARD=1 timeout -k 5 900 python bench.py --n-obs 10000000 --n-vars 4000 --steps 1 --warmup 0 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side --no-properties > "$OUT/bench_c5_guard.json" 2> "$OUT/bench_c5_guard.err" < /dev/null
echo "10M x 4k under SCAMD_GUARD rc=$?"; tail -2 "$OUT/bench_c5_guard.err" | cut -c1-300
