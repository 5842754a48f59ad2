#!/bin/bash
# Round-4 evidence run: the weak-structure bench line, counters of the Leiden decision kernels (restricted by name).
set -u
TAG="${1:-r04n}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout -k 5 400 python bench.py --structure weak --steps 5 --warmup 2 --cpu-sizes 0 --no-side --h2h-reps 0 --no-noise-variant > "$OUT/bench_weak.json" 2> "$OUT/bench_weak.err" < /dev/null
python - "$OUT/bench_weak.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("weak", round(d["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, d["result"]["n_communities"], d["result"]["modularity"], d["result"]["labels_sha"], "properties", d["full_size_properties"]["failed_gates"])
PY
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
cd /tmp
i=0
for P in "$PMC1" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex 'ld_move_kernel|ld_refine_propose_kernel|ld_agg_wave' --pmc $P -d /tmp/pmc_${TAG}_$i -o ld -- python "$R/tools/leiden_only.py" 1000000 planted 1 > "$OUT/ld_pmc$i.log" 2>&1 < /dev/null
  echo "leiden pmc$i rc=$?"
  find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} "$OUT/leiden_pmc$i.csv" \;
done
cd "$R"
python - "$OUT" <<'PY'
import collections, csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/leiden_pmc*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void scamd::", "").replace("scamd::", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); calls[name] += 1; dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for name in acc:
        print(f.split("/")[-1], name[:34], "calls", calls[name], f"us {dur[name]:.0f}", {k: f"{v:.3g}" for k, v in acc[name].items()})
PY
