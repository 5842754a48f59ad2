#!/bin/bash
# Round-2 measurement on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes of the select
# kernel on the bench's OWN embedding.  Usage (via gpurun): bash tools/measure_r02.sh <tag>
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 400 $OUT/bench.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > $R/$OUT/bench_prof.log 2>&1
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $R/$OUT/bench_kernel_stats.csv \;
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
i=0
for P in "$PMC1" "$PMC2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $P -d /tmp/pmc_${TAG}_$i -o knn -- python $R/tools/knn_only.py 1000000 1 > $R/$OUT/pmc$i.log 2>&1
  find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} $R/$OUT/knn_pmc$i.csv \;
  grep "knn n=" $R/$OUT/pmc$i.log
done
cd $R
python - <<PY
import csv, glob, collections, json
tot = {}
for f in sorted(glob.glob("$OUT/knn_pmc*.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "knn_select_reg" in r.get("Kernel_Name", "")]
    if not rows:
        continue
    gmax = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(float)
    name = ""
    for r in rows:
        if int(r["Grid_Size"]) == gmax:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            name = r["Kernel_Name"]
    print(f, gmax, dict(acc))
    tot.update(acc)
    tot["kernel"] = name[:160]
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    out = {"kernel": tot["kernel"], "mode": "ivf", "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
           "bytes_per_launch": 2.0 * tot["FETCH_SIZE"] * 1024 + tot["WRITE_SIZE"] * 1024,
           "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/knn_only.py 1000000 on the bench's own "
                   "embedding (cell-pruned sweep, one launch, summed over the dispatch's rows = all XCDs); FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); L2-side fabric requests incl. "
                   "Infinity-Cache hits"}
    json.dump(out, open("$OUT/knn_select_traffic.json", "w"), indent=1)
    print(out)
if "SQ_INSTS_MFMA" in tot:
    print("VALU per MFMA", tot["SQ_INSTS_VALU"] / tot["SQ_INSTS_MFMA"], "SALU per MFMA", tot.get("SQ_INSTS_SALU", 0) / tot["SQ_INSTS_MFMA"])
PY
