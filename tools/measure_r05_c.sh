#!/bin/bash
# Round-5 third GPU call: the two-rank bench line on one device + the UMAP quality floor (new GPU tests), then BASELINE
# configs[4] sizes (10M x 4k) on one GPU with the APPROXIMATE search in the timed path (--knn-nprobe 8): sampled recall
# against a float64 brute force over all 10M rows, every other full-size property enforced.
set -u
TAG="${1:-r05c}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
timeout -k 5 900 python -m pytest tests/test_gpu_sharded_one_device.py tests/test_gpu_umap.py -m gpu -q -s -p no:faulthandler > "$OUT/pytest_new.log" 2>&1 < /dev/null
echo "new tests rc=$?"; grep -E "graph neighbours|passed|failed|Error" "$OUT/pytest_new.log" | cut -c1-400 | tail -8
timeout -k 5 1500 python bench.py --n-obs 10000000 --n-vars 4000 --steps 2 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side --knn-nprobe 8 > "$OUT/bench_c5_ivf.json" 2> "$OUT/bench_c5_ivf.err" < /dev/null
echo "10M x 4k approximate rc=$?"; tail -3 "$OUT/bench_c5_ivf.err" | cut -c1-300
python - "$OUT/bench_c5_ivf.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("10M x 4k nprobe 8:", round(d["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, d["result"])
    print("knn properties", d["full_size_properties"]["knn"], "failed:", d["full_size_properties"]["failed_gates"])
    print("leiden", {k: v for k, v in d["leiden"].items() if k != "note"}, d["full_size_properties"]["leiden"])
    print("roofline", {k: d["roofline"][k] for k in ("frac", "launch_ms", "pairs_evaluated_fraction")})
except Exception as exc:  # noqa: BLE001
    print("no line:", exc)
PY
