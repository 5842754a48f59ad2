#!/bin/bash
# Round-5 GPU call: HEAD with the CPM objective -- GPU suite, smoke, a bench line (labels of the modularity path must not move).
set -u
TAG="${1:-r05m}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
timeout -k 5 1200 python -m pytest tests -m gpu -q -p no:faulthandler > "$OUT/pytest_gpu.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log" | cut -c1-300
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 < /dev/null
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout -k 5 900 python bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
echo "bench rc=$?"; tail -2 "$OUT/bench.err" | cut -c1-300
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
    print("bench:", round(d["value"]), "cells/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "labels", d["result"]["labels_sha"])
    for st in ("none", "weak"):
        sn = d["structure_" + st]
        print(st, round(sn["ms_per_step"], 1), {k: round(v, 1) for k, v in sn["stage_ms"].items()}, sn["labels_sha"], sn["leiden_guarantees"]["failed_gates"], sn["leiden_n_iterations_2"]["ms"])
    print("roofline", round(d["roofline"]["frac"], 4), d["roofline"]["launch_ms"], "failed gates", d["parity"]["failed_gates"], d.get("variant_failed_gates"), d["full_size_properties"]["failed_gates"])
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
