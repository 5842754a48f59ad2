#!/bin/bash
# FETCH_SIZE / WRITE_SIZE PMC passes of the kNN select kernel at 1M (HBM-side traffic per launch)
TAG=${1:-traffic}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $P -d /tmp/pmc_${TAG}_$i -o knn -- python $R/tools/knn_only.py 1000000 1 > $R/$OUT/pmc$i.log 2>&1
  find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} $R/$OUT/knn_traffic_pmc$i.csv \;
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/knn_traffic_pmc*.csv")):
    for row in csv.DictReader(open(f)):
        if "knn_select" in row.get("Kernel_Name", "") and int(row["Grid_Size"]) > 100000:
            print(f, row["Counter_Name"], row["Counter_Value"], "dur_ms", (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
PY
