#!/bin/bash
# Copy the summaries of a `tools/measure.sh <tag>` run from gpurun_out/<tag>/ into profiles/ (tracked), named per round.
# Usage (in the build container, after the gpurun call returned): bash tools/install_profiles.sh <tag>
set -e
TAG=${1:?usage: install_profiles.sh <tag>}
SRC=gpurun_out/$TAG
test -s $SRC/bench.json || { echo "no $SRC/bench.json"; exit 1; }
cp $SRC/bench.json profiles/${TAG}_bench_1M.json
cp $SRC/bench_kernel_stats.csv profiles/${TAG}_bench_1M_kernel_stats.csv
for i in 1 2 3 4; do
  test -s $SRC/knn_pmc$i.csv && cp $SRC/knn_pmc$i.csv profiles/${TAG}_knn_select_pmc$i.csv
done
test -s $SRC/knn_select_traffic.json && cp $SRC/knn_select_traffic.json profiles/knn_select_traffic.json
python - <<PY
import json
d = json.loads(open("profiles/${TAG}_bench_1M.json").read().strip().splitlines()[-1])
print("installed ${TAG}:", round(d["value"]), "cells/s,", round(d["ms_per_step"], 1), "ms/step, roofline frac", round(d["roofline"]["frac"], 3))
PY
