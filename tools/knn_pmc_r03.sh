TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
PMC3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
i=0
cd /tmp
for P in "$PMC1" "$PMC2" "$PMC3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $P -d /tmp/pmc_${TAG}_$i -o knn -- python $R/tools/knn_only.py "$@" > $R/$OUT/pmc$i.log 2>&1
  find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} $R/$OUT/knn_pmc$i.csv \;
  grep "knn n=" $R/$OUT/pmc$i.log | tail -1
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/knn_pmc*.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "knn_select_reg" in r.get("Kernel_Name", "")]
    if not rows: print(f, "no rows"); continue
    gmax = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(float)
    for r in rows:
        if int(r["Grid_Size"]) == gmax: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f, gmax, {k: f"{v:.4g}" for k, v in acc.items()})
PY
