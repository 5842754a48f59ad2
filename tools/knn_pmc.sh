#!/bin/bash
# PMC passes for the kNN select kernel under a given SCAMD_KNN_MODE list
TAG=${1:-knn_pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
PMC3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH"
for M in ${MODES:-1}; do
  i=0
  for P in "$PMC1" "$PMC2" "$PMC3"; do
    i=$((i+1))
    SCAMD_KNN_MODE=$M timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $P -d /tmp/pmc_${TAG}_${M}_$i -o knn -- python $R/tools/knn_only.py 1000000 1 > $R/$OUT/pmc_m${M}_$i.log 2>&1
    find /tmp/pmc_${TAG}_${M}_$i -name '*counter_collection.csv' -exec cp {} $R/$OUT/knn_m${M}_pmc$i.csv \;
  done
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/knn_m*_pmc*.csv")):
    acc = collections.defaultdict(float)
    dur = 0
    for row in csv.DictReader(open(f)):
        if "knn_select" in row.get("Kernel_Name", "") and int(row["Grid_Size"]) > 100000:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    print(f, "dur_ms=%.1f" % (dur / 1e6), {k: "%.4g" % v for k, v in acc.items()})
PY
