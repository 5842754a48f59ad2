#!/bin/bash
# Round-5 GPU call: the driver's N = 2 launch of bench.py at FULL size with both ranks on the one GPU of the lease
# (SCAMD_BENCH_ONE_DEVICE=1: gloo collectives, RCCL refuses two ranks on one device).  The times are those of two processes
# sharing a device -- NOT a scaling measurement; what the line records is the N > 1 path itself at 1M x 2k: shards, per-rank
# stage table, bytes of every collective, the rank-0-only share, and that the gates hold on the sharded result.
set -u
TAG="${1:-r05_2ranks}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
SCAMD_BENCH_ONE_DEVICE=1 MASTER_ADDR=127.0.0.1 timeout -k 5 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29547 bench.py --gpus 2 --steps 5 --warmup 2 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_2ranks.json" 2> "$OUT/bench_2ranks.err" < /dev/null
echo "rc=$?"; tail -3 "$OUT/bench_2ranks.err" | cut -c1-300
python - "$OUT/bench_2ranks.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["n_gpus"], d["scaling"], round(d["ms_per_step"], 1), "ms", json.dumps(d["multi_gpu"])[:1500])
    print("failed gates", d["parity"]["failed_gates"], d["full_size_properties"]["failed_gates"], d["result"]["labels_sha"])
except Exception as exc:  # noqa: BLE001
    print("no line:", exc)
PY
