#!/bin/bash
# Round-4, last GPU seconds: the weak-structure line and Leiden alone with the final code.
set -u
TAG="${1:-r04u}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$R"
timeout -k 5 70 python bench.py --structure weak --steps 5 --warmup 2 --cpu-sizes 0 --no-side --h2h-reps 0 --no-noise-variant > "$OUT/bench_weak.json" 2> "$OUT/bench_weak.err" < /dev/null
python - "$OUT/bench_weak.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("weak", round(d["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, d["result"]["n_communities"], d["result"]["modularity"], d["result"]["labels_sha"], "properties", d["full_size_properties"]["failed_gates"])
except Exception as exc:  # noqa: BLE001
    print("no weak line:", exc)
PY
timeout -k 5 40 python tools/leiden_only.py 1000000 planted 5 2>&1 | grep "^leiden"
