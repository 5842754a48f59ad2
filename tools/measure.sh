#!/bin/bash
# Round measurement on the GPU box: tests, bench line, rocprofv3 kernel stats, PMC passes for the kNN select kernel.
# Usage (via gpurun): bash tools/measure.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $R/$OUT/bench_prof.log 2>&1
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $R/$OUT/bench_kernel_stats.csv \;
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
i=0
for P in "$PMC1" "$PMC2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $P -d /tmp/pmc_${TAG}_$i -o knn -- python $R/tools/knn_only.py 1000000 1 > $R/$OUT/pmc$i.log 2>&1
  find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} $R/$OUT/knn_pmc$i.csv \;
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/knn_pmc*.csv")):
    acc = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if "knn_select" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
    print(f, dict(acc))
PY
