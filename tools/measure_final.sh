#!/bin/bash
# End-of-round measurement on the GPU box (one gpurun call): full GPU test suite, smoke, the default bench line, and the
# rocprofv3 kernel stats of the same bench command.  Usage (via gpurun): bash tools/measure_final.sh <tag>
TAG=${1:-r02z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 330 python -m pytest tests -m gpu -q < /dev/null > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 90 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 330 python bench.py --steps 20 --warmup 5 < /dev/null > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 600 $OUT/bench.json; echo
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --cpu-sizes 0 --no-noise-variant --h2h-reps 0 --no-side --no-verify > $R/$OUT/bench_prof.log 2>&1 < /dev/null
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $R/$OUT/bench_kernel_stats.csv \;
ls -la $R/$OUT
