#!/bin/bash
# kernel trace (per-dispatch durations, in order) of one bench step, filtered to a kernel-name prefix
TAG=${1:-trace}
PREFIX=${2:-ld_}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/$OUT/trace.log 2>&1
F=$(find /tmp/tr_$TAG -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$F")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open("$R/$OUT/trace_filtered.csv", "w")
out.write("i,name,grid,start_us,dur_us\n")
t0 = int(rows[0]["Start_Timestamp"])
for i, r in enumerate(rows):
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("scamd::", "").replace("void ", "")
    if "$PREFIX" in short or "knn" in short or "spmm" in short or "fuzzy" in short:
        out.write("%d,%s,%s,%.1f,%.1f\n" % (i, short[:50], r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
out.close()
# everything in front of the first kNN kernel = the PCA stage (all kernels, unfiltered)
pca = open("$R/$OUT/trace_pca.csv", "w")
pca.write("i,name,grid,start_us,dur_us\n")
started = False
for i, r in enumerate(rows):
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("scamd::", "").replace("void ", "")
    if "gram_absmax" in short:
        started = True
    if "ivf_" in short or "knn_" in short:
        break
    if started:
        pca.write("%d,%s,%s,%.1f,%.1f\n" % (i, short[:60], r.get("Grid_Size", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
pca.close()
PY
wc -l $R/$OUT/trace_filtered.csv
