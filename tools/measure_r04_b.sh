#!/bin/bash
# Round-4 second GPU call: the new certificate tests, the kNN tests, A/B of the XCD-aware launch order and the row-wise
# re-rank (kernel trace of tools/knn_only.py), a short bench.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/measure_r04_b.sh r04b'
set -u
TAG="${1:-r04b}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout -k 5 600 python -m pytest tests/test_gpu_knn_certificate.py -q -s -p no:faulthandler > "$OUT/pytest_cert.log" 2>&1 < /dev/null
echo "cert rc=$?"; grep -E "ratio|bound =|cert_scale|differing|passed|failed|Error|error" "$OUT/pytest_cert.log" | cut -c1-220 | tail -30
timeout -k 5 600 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_kernels.py -q -k "knn" -p no:faulthandler > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "knn tests rc=$?"; tail -3 "$OUT/pytest_knn.log" | cut -c1-200
cd /tmp
for mode in 1 0; do
  SCAMD_KNN_XCD_ORDER=$mode timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kn_${TAG}_$mode -o knn -- python "$R/tools/knn_only.py" 1000000 3 > "$OUT/knn_only_xcd$mode.log" 2>&1 < /dev/null
  echo "xcd=$mode rc=$?"; grep "knn n=" "$OUT/knn_only_xcd$mode.log" | tail -2
  find /tmp/kn_${TAG}_$mode -name '*kernel_stats.csv' -exec cp {} "$OUT/knn_only_xcd${mode}_kernel_stats.csv" \;
  python - "$OUT/knn_only_xcd${mode}_kernel_stats.csv" <<'PY'
import csv, sys
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if any(t in r["Name"] for t in ("knn_", "ivf_")):
            print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
except Exception as exc:
    print("no stats", exc)
PY
done
cd "$R"
timeout -k 5 400 python bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
echo "bench rc=$?"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
    print("bench:", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()},
          "failed gates:", d.get("parity", {}).get("failed_gates"), "properties:", d.get("full_size_properties", {}).get("failed_gates"),
          "labels", d.get("labels_sha"), d.get("structure_none", {}).get("labels_sha"))
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
