#!/bin/bash
# Round-4: quantiser assignment on the bf16 MFMA vs the float32 kernel (kernel trace of tools/knn_only.py), kNN tests.
set -u
TAG="${1:-r04i}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for mode in 1 0 1 0; do
  SCAMD_KNN_ASSIGN_MFMA=$mode timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kn_${TAG}_$mode -o knn -- python "$R/tools/knn_only.py" 1000000 4 > "$OUT/knn_only_assign$mode.log" 2>&1 < /dev/null
  echo "assign_mfma=$mode rc=$?"; grep "knn n=" "$OUT/knn_only_assign$mode.log" | tail -2 | sed 's/.*select/select/' | cut -c1-200
  find /tmp/kn_${TAG}_$mode -name '*kernel_stats.csv' -exec cp {} "$OUT/knn_only_assign${mode}_kernel_stats.csv" \;
  python - "$OUT/knn_only_assign${mode}_kernel_stats.csv" <<'PY'
import csv, sys
try:
    tot = 0.0
    for r in csv.DictReader(open(sys.argv[1])):
        if any(t in r["Name"] for t in ("knn_", "ivf_")):
            tot += float(r["TotalDurationNs"])
            if float(r["AverageNs"]) > 3e4: print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
    print("   kNN kernels total per call (5 calls incl. warm-up on 8192 rows):", tot / 5e6, "ms")
except Exception as exc:
    print("no stats", exc)
PY
done
cd "$R"
timeout -k 5 600 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_kernels.py tests/test_gpu_knn_certificate.py -q -k "knn" -p no:faulthandler > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "knn tests rc=$?"; tail -2 "$OUT/pytest_knn.log" | cut -c1-200
timeout -k 5 300 python bench.py --steps 10 --warmup 3 --cpu-sizes 0 --no-side --h2h-reps 0 --no-noise-variant > "$OUT/bench_short.json" 2> "$OUT/bench_short.err" < /dev/null
python - "$OUT/bench_short.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("bench", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "labels", d["result"]["labels_sha"],
      "properties", d.get("full_size_properties", {}).get("failed_gates"), "pairs frac", d["roofline"]["pairs_evaluated_fraction"])
PY
