#!/bin/bash
# First GPU call of the round after this one: everything that was built after round 1's GPU minutes ran out.
#   gpurun --timeout 1500 -- 'bash tools/first_gpu_call.sh r02a'
# Outputs under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# 1. the whole GPU suite (the late file last): host-side rework of pp.neighbors / tl.leiden, out-of-core route, seurat_v3
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
# 2. wall time of the drop-in calls (host in / host out) after the host-side rework
timeout 300 python tools/dropin_wall.py 1000000 > $OUT/dropin_wall.log 2>&1; tail -2 $OUT/dropin_wall.log
# 3. out-of-core PCA: in-memory vs resident vs streamed, zarr and h5ad, pageable vs page-locked staging
timeout 600 python tools/streaming_probe.py 1000000 /tmp/scamd_stream > $OUT/streaming_probe.json 2> $OUT/streaming_probe.err
tail -c 1200 $OUT/streaming_probe.json; echo
# 4. A/B of the pruned kNN sweep's register budget (3 blocks per CU with spills vs 2 without)
for W in 3 2; do
  SCAMD_KNN_IVF_WPS=$W timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_wps$W.json 2> $OUT/bench_wps$W.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_wps$W.json").read().strip().splitlines()[-1])
    print("WPS=$W", d.get("value"), d.get("ms_per_step"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("launch_ms"))
except Exception as e:
    print("WPS=$W: no bench line", e)
PY
done
