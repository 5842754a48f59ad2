#!/bin/bash
# two --pmc passes over the Gram kernels (tools/gram_only.py), for the build / environment given: tools/gram_pmc.sh <tag> [ENV=VAL ...]
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  ( cd /tmp; env "$@" timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex 'gram_(packed|tile|quad)_kernel' --pmc $P -d /tmp/gpmc_${TAG}_$i -o gram -- python "$R/tools/gram_only.py" > "$OUT/gram_pmc$i.log" 2>&1 < /dev/null; echo "pmc$i rc=$?" )
  find /tmp/gpmc_${TAG}_$i -name '*counter_collection.csv' -exec cp {} "$OUT/gram_pmc$i.csv" \;
done
python - "$OUT" <<'PY'
import csv, collections, sys
out = sys.argv[1]
for i in (1, 2):
    tot = collections.defaultdict(lambda: collections.Counter()); calls = collections.Counter()
    try:
        for r in csv.DictReader(open(f"{out}/gram_pmc{i}.csv")):
            k = r["Kernel_Name"].split("(")[0][-40:]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    except OSError as e:
        print("no file", e); continue
    for k, c in tot.items():
        print(k, {n: f"{v:.3e}" for n, v in sorted(c.items())})
PY
