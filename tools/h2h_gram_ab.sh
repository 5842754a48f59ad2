cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" python bench.py --steps 2 --warmup 1 --cpu-sizes 0 --no-noise-variant --no-side --no-verify --no-properties --h2h-reps 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); h=d['host_to_host']; print('$*', 'best', {k: round(x,1) for k,x in h['best'].items()})"; }
run SCAMD_PCA_OVERLAP_CHUNKS=6
run SCAMD_GRAM_LEGACY=1
run SCAMD_PCA_OVERLAP_CHUNKS=3
run SCAMD_PCA_OVERLAP_CHUNKS=2
run SCAMD_PCA_OVERLAP_UPLOAD=0
run SCAMD_PCA_OVERLAP_CHUNKS=6
