#!/bin/bash
# First GPU call of round 4 (through gpurun, ~6 GPU-minutes): the four kernel edits of round 3 that were validated on the
# host emulator only -- per-workgroup atomics in fss_max/sum_kernel, four slots per group in fss_recip_rec_kernel,
# two batches in flight in knn_rerank_kernel, one-vertex rounds on tiny Leiden levels -- get their first GPU test run and
# their first timings.  EVERY stage under its own `timeout`; no stage reads a file whose name comes from a glob that may
# be empty (round 3 lost 9 GPU-minutes to `grep ... $empty_variable` waiting on stdin).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/measure_r04_first.sh r04a'
set -u
tag="${1:-r04a}"
out="gpurun_out/${tag}"
mkdir -p "${out}"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"

timeout 600 python -m pytest tests -m gpu -q -p no:faulthandler > "${out}/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -2 "${out}/pytest_gpu.log" | cut -c1-200

timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "${out}/smoke.log" 2>&1
echo "smoke rc=$?"; tail -1 "${out}/smoke.log"

timeout 200 rocprofv3 --kernel-trace -d "${out}/prof_fuzzy" -o fz -- python tools/fuzzy_only.py 1000000 3 planted > "${out}/fuzzy_only.log" 2>&1
echo "fuzzy rc=$?"; grep "fuzzy n=" "${out}/fuzzy_only.log" | tail -3

timeout 300 python bench.py --steps 10 --warmup 3 > "${out}/bench.json" 2> "${out}/bench.err"
echo "bench rc=$?"

# per-kernel table from the rocpd database (this rocprofv3 writes a .db, not a CSV, when -o is given)
timeout 60 python - "${out}" <<'PY'
import glob, json, sqlite3, sys
out = sys.argv[1]
for db in glob.glob(out + "/prof_fuzzy/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), avg(end-start) from kernels where name like '%fss_%' or name like '%knn_rerank%' "
                       "group by name order by 3 desc").fetchall()
    for name, calls, avg in rows:
        print(f"{name[:70]:70s} calls {calls:3d} avg {avg / 1e3:8.1f} us")
try:
    line = [l for l in open(out + "/bench.json") if l.startswith("{")][-1]
    d = json.loads(line)
    print("bench:", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()},
          "failed gates:", d.get("parity", {}).get("failed_gates"), "properties:", d.get("full_size_properties", {}).get("failed_gates"))
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
