#!/bin/bash
# Round-3 measurement on the GPU box: GPU test suite, smoke, bench line, rocprofv3 kernel stats of the same command.
# Usage (via gpurun): bash tools/measure_r03.sh <tag>      (no --pmc passes: see profiles/README.md)
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side > $R/$OUT/bench_prof.log 2>&1
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $R/$OUT/bench_kernel_stats.csv \;
cd $R
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["stage_ms_per_step"])
print("h2h", d.get("value_host_to_host"), d["host_to_host"]["best"])
sn=d["structure_none"]; print("none", sn["ms_per_step"], sn["stage_ms"], sn["n_communities"], sn["modularity"], sn["labels_sha"])
r=d["roofline"]; print({k:r[k] for k in ("engine","achieved","peak","frac","launch_ms","f32_equivalent_tflops","pairs_evaluated_fraction")})
print("failed gates", d["parity"]["failed_gates"], "weak", d["parity"]["weak"]["failed_gates"])
PY
