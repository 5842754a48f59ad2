cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  SCAMD_PCA_OVERLAP_UPLOAD=$v python bench.py --steps 2 --warmup 1 --cpu-sizes 0 --no-noise-variant --no-side --no-verify --no-properties --h2h-reps 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); h=d['host_to_host']; print('overlap=$v', 'best', {k: round(x,1) for k,x in h['best'].items()})"
done
