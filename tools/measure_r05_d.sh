#!/bin/bash
# Round-5 fourth GPU call: the UMAP quality floor with the new spectral solver; A/B of the workgroup counts of the coarse-row
# builder's two workgroup tiers (planted + weak, Leiden alone); kernel stats of Leiden on the weak graph.
set -u
TAG="${1:-r05d}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
timeout -k 5 600 python -m pytest tests/test_gpu_umap.py -m gpu -q -s -p no:faulthandler > "$OUT/pytest_umap.log" 2>&1 < /dev/null
echo "umap tests rc=$?"; grep -E "graph neighbours|passed|failed|Error" "$OUT/pytest_umap.log" | cut -c1-400 | tail -5
for K in "" "SCAMD_LEIDEN_AGG_BIG_GRID=256" "SCAMD_LEIDEN_AGG_BIG_GRID=128" "SCAMD_LEIDEN_AGG_BIG_GRID=64" "SCAMD_LEIDEN_AGG_MID_GRID=384" "SCAMD_LEIDEN_AGG_MID_GRID=256" "SCAMD_LEIDEN_AGG_BIG_GRID=128 SCAMD_LEIDEN_AGG_MID_GRID=256"; do
  echo "[$K] $(env $K timeout -k 5 200 python tools/leiden_only.py 1000000 planted 5 2>&1 | grep 'leiden n=' | tail -1 | cut -c1-120)" | tee -a "$OUT/leiden_agg_grid.log"
done
for K in "" "SCAMD_LEIDEN_AGG_BIG_GRID=128 SCAMD_LEIDEN_AGG_MID_GRID=256"; do
  echo "[$K] $(env $K timeout -k 5 300 python tools/leiden_only.py 1000000 weak 3 2>&1 | grep 'leiden n=' | tail -1 | cut -c1-120)" | tee -a "$OUT/leiden_agg_grid.log"
done
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o leiden -- python "$R/tools/leiden_only.py" 1000000 weak 1 > "$OUT/leiden_weak_prof.log" 2>&1 < /dev/null
echo "leiden weak prof rc=$?"
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/leiden_weak_kernel_stats.csv" \;
head -25 "$OUT/leiden_weak_kernel_stats.csv" | cut -c1-60,200-330 | sed 's/  */ /g' | cut -c1-200
