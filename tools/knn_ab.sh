#!/bin/bash
# A/B of the kNN select kernels on the GPU box (register-list vs legacy LDS-list) + PMC of the new one.
TAG=${1:-knn_ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "knn or mfma" > $OUT/pytest_knn.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_knn.log
timeout 300 python tools/knn_only.py 1000000 2 2>&1 | grep knn | tee $OUT/new.log
SCAMD_KNN_LEGACY=1 timeout 300 python tools/knn_only.py 1000000 1 2>&1 | grep knn | tee $OUT/legacy.log
timeout 300 python tools/knn_only.py 125000 2 2>&1 | grep knn | tee -a $OUT/new.log
cd /tmp
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
