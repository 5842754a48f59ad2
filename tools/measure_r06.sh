#!/bin/bash
# Round-6 measurement driver (ONE parameterised script instead of one file per call):
#   bash tools/measure_r06.sh <tag> <step> [<step> ...]
# steps: build | oracle_iters | bench_short | bench_full | bench_weak | bench_none | bench_default | prof | prof_weak |
#        leiden3 | leiden_prof:<structure> | knn_pmc | pytest | pytest:<-k expr> | smoke | c5
set -u
TAG="${1:-r06}"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(round(d["value"]), "cells/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()},
          "roofline", round(d["roofline"]["frac"], 3), d["roofline"].get("launch_ms"), "h2h", d.get("value_h2h", d.get("value_host_to_host")))
    if "leiden" in d: print("leiden", {k: v for k, v in d["leiden"].items() if k not in ("note", "bound", "peak_GBps")})
    for st in ("none", "weak"):
        sn = d.get("structure_" + st)
        if sn: print(st, round(sn["ms_per_step"], 1), {k: round(v, 1) for k, v in sn["stage_ms"].items()}, sn["n_communities"], sn["labels_sha"], sn["leiden_guarantees"].get("failed_gates"), {k: v for k, v in sn["leiden"].items() if k in ("iterations", "launches", "host_round_trips", "ended_by_iteration_cap")})
    print("failed gates", d.get("parity", {}).get("failed_gates"), d.get("variant_failed_gates"), d.get("full_size_properties", {}).get("failed_gates"), "labels", d.get("result", {}).get("labels_sha"))
except Exception as exc:  # noqa: BLE001
    print("no bench line:", exc)
PY
}
stats() { python - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total", round(tot / 1e6, 1), "ms kernel time;", sum(int(r["Calls"]) for r in rows), "launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f} % {int(r['Calls']):6d} calls avg {float(r['AverageNs'])/1e3:8.1f} us max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:70]}")
PY
}
for STEP in "$@"; do
  ARG="${STEP#*:}"; NAME="${STEP%%:*}"
  case "$NAME" in
    build) python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?";;
    oracle_iters)  # background: single-threaded CPU job beside the GPU steps
      ( timeout -k 5 1500 python tools/oracle_iters_probe.py 1000000 "${ARG:-weak}" ${ORACLE_SEEDS:-0} ${GPU_SEEDS:-0} > "$OUT/oracle_iters_${ARG:-weak}.log" 2>&1 < /dev/null ) & ;;
    bench_short) timeout -k 5 600 python bench.py --steps 20 --warmup 5 --cpu-sizes 0 --no-noise-variant --no-side --no-verify > "$OUT/bench_short.json" 2> "$OUT/bench_short.err" < /dev/null; echo "bench_short rc=$?"; line "$OUT/bench_short.json";;
    bench_full) timeout -k 5 1500 python bench.py --steps 20 --warmup 5 ${ARG_FULL:-} > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null; echo "bench rc=$?"; tail -2 "$OUT/bench.err" | cut -c1-300; line "$OUT/bench.json";;
    bench_weak|bench_none) ST="${NAME#bench_}"; timeout -k 5 600 python bench.py --structure $ST --steps 5 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > "$OUT/bench_$ST.json" 2> "$OUT/bench_$ST.err" < /dev/null; echo "bench $ST rc=$?"; line "$OUT/bench_$ST.json";;
    bench_default) ( time timeout -k 5 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" < /dev/null ) 2> "$OUT/bench_default.time"; echo "bench default rc=$? $(grep real "$OUT/bench_default.time")"; line "$OUT/bench_default.json";;
    prof|prof_weak|prof_none) ST=planted; [ "$NAME" != prof ] && ST="${NAME#prof_}"
      ( cd /tmp; timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$ST -o bench -- python "$R/bench.py" --structure $ST --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side --no-verify --no-properties > "$OUT/bench_prof_$ST.log" 2>&1 < /dev/null; echo "prof $ST rc=$?" )
      find /tmp/prof_${TAG}_$ST -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_${ST}_kernel_stats.csv" \;
      test -s "$OUT/bench_${ST}_kernel_stats.csv" && stats "$OUT/bench_${ST}_kernel_stats.csv";;
    leiden3) for ST in ${ARG:-planted weak none}; do [ "$ST" = leiden3 ] && continue; timeout -k 5 300 python tools/leiden_only.py 1000000 $ST 3 2>&1 | grep "leiden n=" ; done | tee "$OUT/leiden3.log";;
    leiden_prof) ( cd /tmp; timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lprof_${TAG}_$ARG -o leiden -- python "$R/tools/leiden_only.py" 1000000 $ARG 1 > "$OUT/leiden_${ARG}_prof.log" 2>&1 < /dev/null )
      find /tmp/lprof_${TAG}_$ARG -name '*kernel_stats.csv' -exec cp {} "$OUT/leiden_${ARG}_kernel_stats.csv" \;
      test -s "$OUT/leiden_${ARG}_kernel_stats.csv" && stats "$OUT/leiden_${ARG}_kernel_stats.csv";;
    knn_pmc) PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; i=0
      for P in "$PMC1" "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1))
        ( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex 'knn_select_reg' --pmc $P -d /tmp/pmc_${TAG}_$i -o knn -- python "$R/tools/knn_only.py" 1000000 1 > "$OUT/pmc$i.log" 2>&1 < /dev/null; echo "pmc$i rc=$?" )
        find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} "$OUT/knn_pmc$i.csv" \;
      done
      test -s "$OUT/knn_pmc2.csv" && test -s "$OUT/knn_pmc3.csv" && python tools/make_traffic_json.py "$OUT/knn_pmc2.csv" "$OUT/knn_pmc3.csv" > /dev/null && cp profiles/knn_select_traffic.json "$OUT/knn_select_traffic.json";;
    pytest) if [ "$ARG" != "$NAME" ]; then timeout -k 5 1500 python -m pytest tests -m gpu -q -p no:faulthandler -k "$ARG" > "$OUT/pytest_gpu.log" 2>&1 < /dev/null; else timeout -k 5 1800 python -m pytest tests -m gpu -q -p no:faulthandler > "$OUT/pytest_gpu.log" 2>&1 < /dev/null; fi
      echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log" | cut -c1-300;;
    smoke) timeout -k 5 180 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 < /dev/null; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log";;
    c5) timeout -k 5 900 python bench.py --n-obs 10000000 --n-vars 4000 --steps 2 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side --no-verify ${ARG_C5:-} > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" < /dev/null; echo "c5 rc=$?"; line "$OUT/bench_c5.json";;
    cmd) timeout -k 5 900 bash -c "$ARG" > "$OUT/cmd.log" 2>&1 < /dev/null; echo "cmd rc=$?"; tail -30 "$OUT/cmd.log" | cut -c1-400;;
    *) echo "unknown step $STEP";;
  esac
done
wait
for f in "$OUT"/oracle_iters_*.log; do test -s "$f" && { echo "== $f"; grep -c "oracle leiden. iteration" "$f"; grep "^gpu\|^oracle\|^ARI" "$f" | cut -c1-260; tail -3 "$f" | cut -c1-300; }; done
exit 0
