#!/bin/bash
# host-to-host time of the path with / without the overlapped upload of pp.pca, and over its chunk count (one box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" python bench.py --steps 2 --warmup 1 --cpu-sizes 0 --no-noise-variant --no-side --no-verify --no-properties --h2h-reps 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); h=d['host_to_host']; print('$*', 'best', {k: round(x,1) for k,x in h['best'].items()})"; }
for c in 2 3 4 6 8 12; do run SCAMD_PCA_OVERLAP_CHUNKS=$c; done
run SCAMD_PCA_OVERLAP_UPLOAD=0
run SCAMD_PCA_OVERLAP_CHUNKS=4
