#!/bin/bash
# Round-4 third GPU call: certificate tests after the float64-scan retry, traffic of the XCD-ordered sweep, kNN knob
# sweep, a per-level Leiden trace, Leiden x5 under SCAMD_GUARD (the unreproduced faults of round 3), the weak-graph probe.
set -u
TAG="${1:-r04c}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout -k 5 900 python -m pytest tests/test_gpu_knn_certificate.py -q -s -p no:faulthandler > "$OUT/pytest_cert.log" 2>&1 < /dev/null
echo "cert rc=$?"; grep -E "bound =|cert_scale|differing|passed|failed|Error" "$OUT/pytest_cert.log" | cut -c1-220 | tail -30
timeout -k 5 600 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_kernels.py -q -k "knn" -p no:faulthandler > "$OUT/pytest_knn.log" 2>&1 < /dev/null
echo "knn tests rc=$?"; tail -2 "$OUT/pytest_knn.log" | cut -c1-200
cd /tmp
for P in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex 'knn_select_reg' --pmc $P -d /tmp/pmc_${TAG}_$P -o knn -- python "$R/tools/knn_only.py" 1000000 1 > "$OUT/pmc_$P.log" 2>&1 < /dev/null
  echo "pmc $P rc=$?"
  find /tmp/pmc_${TAG}_$P -name '*counter_collection.csv' -exec cp {} "$OUT/knn_xcd_pmc_$P.csv" \;
done
python - "$OUT" <<'PY'
import collections, csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/knn_xcd_pmc_*.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "knn_select_reg" in r.get("Kernel_Name", "")]
    if not rows:
        print(f, "no rows"); continue
    gmax = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(float)
    for r in rows:
        if int(r["Grid_Size"]) == gmax:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[-1], gmax, {k: f"{v:.4g}" for k, v in acc.items()})
PY
cd "$R"
for knob in "" "SCAMD_KNN_THR_MARGIN=8" "SCAMD_KNN_THR_MARGIN=6" "SCAMD_KNN_PREPASS_TILES=64" "SCAMD_KNN_PREPASS_TILES=16" "SCAMD_KNN_XCD_ORDER=0"; do
  env $knob timeout -k 5 200 python tools/knn_only.py 1000000 3 > "$OUT/knob.log" 2>&1 < /dev/null
  echo "knob [$knob]: $(grep 'knn n=' "$OUT/knob.log" | tail -1 | cut -c30-130)"
done
SCAMD_LEIDEN_DEBUG=1 timeout -k 5 300 python tools/leiden_only.py 1000000 planted 1 > "$OUT/leiden_trace_planted.log" 2>&1 < /dev/null
echo "leiden trace rc=$?"; grep -E "^\[leiden\] (level|small|iteration)" "$OUT/leiden_trace_planted.log" | tail -40 | cut -c1-170
SCAMD_GUARD=1 timeout -k 5 400 python tools/leiden_only.py 1000000 planted 5 > "$OUT/leiden_guard.log" 2>&1 < /dev/null
echo "leiden x5 under SCAMD_GUARD rc=$?"; tail -2 "$OUT/leiden_guard.log" | cut -c1-200
timeout -k 5 900 python tools/leiden_weak_probe.py 100000 weak > "$OUT/weak_probe.log" 2>&1 < /dev/null
echo "weak probe rc=$?"; grep -E "^(oracle:|gpu )" "$OUT/weak_probe.log" | cut -c1-330
