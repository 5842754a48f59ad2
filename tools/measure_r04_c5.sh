#!/bin/bash
# BASELINE configs[4] sizes (10M x 4k) on one GPU, full-size properties enforced.
set -u
TAG="${1:-r04c5}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout -k 5 1200 python bench.py --n-obs 10000000 --n-vars 4000 --steps 2 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" < /dev/null
echo "10M x 4k rc=$?"; tail -3 "$OUT/bench_c5.err" | cut -c1-300
python - "$OUT/bench_c5.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("10M x 4k:", round(d["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, d["result"],
          "properties failed:", d.get("full_size_properties", {}).get("failed_gates"), "roofline", {k: d["roofline"][k] for k in ("frac", "launch_ms", "pairs_evaluated_fraction")})
except Exception as exc:  # noqa: BLE001
    print("no line:", exc)
PY
