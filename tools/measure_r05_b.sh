#!/bin/bash
# Round-5 second GPU call: rebuild check, the GPU suite, Leiden alone on the three structures (polish / split cost), the
# two-rank bench line on one device (multi_gpu block), bench.
set -u
TAG="${1:-r05b}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
timeout -k 5 900 python -m pytest tests -m gpu -q -p no:faulthandler -x > "$OUT/pytest_gpu.log" 2>&1 < /dev/null
echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log" | cut -c1-300
for ST in planted weak none; do
  timeout -k 5 300 python tools/leiden_only.py 1000000 $ST 3 2>&1 | grep "leiden n=" | tee -a "$OUT/leiden_only.log" | cut -c1-700
done
SCAMD_LEIDEN_POLISH=0 timeout -k 5 300 python tools/leiden_only.py 1000000 weak 3 2>&1 | grep "leiden n=" | sed 's/^/[POLISH=0] /' | tee -a "$OUT/leiden_only.log" | cut -c1-400
SCAMD_BENCH_ONE_DEVICE=1 timeout -k 5 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --n-obs 200000 --steps 2 --warmup 1 > "$OUT/bench_2ranks_one_device.json" 2> "$OUT/bench_2ranks.err" < /dev/null
echo "2-rank bench rc=$?"; python - "$OUT" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1] + "/bench_2ranks_one_device.json") if l.startswith("{")][-1])
    print(json.dumps(d["multi_gpu"])[:1500])
except Exception as exc:  # noqa: BLE001
    print("no 2-rank line:", exc)
PY
timeout -k 5 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
echo "bench rc=$?"; tail -2 "$OUT/bench.err" | cut -c1-400
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
    print("bench:", round(d["value"]), "cells/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()})
    print("leiden", {k: v for k, v in d["leiden"].items() if k != "note"})
    for st in ("none", "weak"):
        sn = d["structure_" + st]
        print(st, round(sn["ms_per_step"], 1), {k: round(v, 1) for k, v in sn["stage_ms"].items()}, sn["n_communities"], sn["labels_sha"], sn["leiden_guarantees"])
        print("   leiden", {k: v for k, v in sn["leiden"].items() if k != "note"})
    print("failed gates", d["parity"]["failed_gates"], "variants", d.get("variant_failed_gates"), "properties", d["full_size_properties"]["failed_gates"])
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
