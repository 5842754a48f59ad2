#!/bin/bash
# A/B of the grid of ivf_pack_image_kernel (SCAMD_KNN_PACK_BLOCKS) by kernel stats of tools/knn_only.py
R="${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for B in ${@:-4096 16384 65536}; do
  ( cd /tmp; SCAMD_KNN_PACK_BLOCKS=$B timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$B -o knn -- python $R/tools/knn_only.py 1000000 2 > /dev/null 2>&1 < /dev/null )
  f=$(find /tmp/kp_$B -name "*kernel_stats.csv" | head -1)
  echo "== blocks $B"
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("ivf_pack_image", "ivf_scatter", "ivf_assign_mfma", "knn_select", "rerank_rows", "fallback")):
        print(f"{float(r['AverageNs']) / 1e3:9.1f} us x {r['Calls']:>3}  {r['Name'][:60]}")
PY
done
