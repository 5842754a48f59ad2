#!/bin/bash
# Quality / time of the GPU Leiden on the 1M weak graph under its knobs (one process per setting; oracle not run).
#   bash tools/leiden_knobs_ab.sh <tag> [structure]
TAG="${1:-r06_knobs}"; ST="${2:-weak}"
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
run() { name="$1"; shift; echo "== $name"; env "$@" timeout -k 5 300 python tools/oracle_iters_probe.py 1000000 $ST none 0,1,2 2>&1 | grep "^gpu seed" | cut -c1-140; }
{
run default SCAMD_NOP=1
run lm_classes16 SCAMD_LEIDEN_LM_CLASSES=16
run lm_classes32 SCAMD_LEIDEN_LM_CLASSES=32
run rf_classes32 SCAMD_LEIDEN_RF_CLASSES=32
run lm_stop0 SCAMD_LEIDEN_LM_STOP_PERMILLE=0
run iter_cap64 SCAMD_LEIDEN_ITER_CAP=64
run lm32_rf32 SCAMD_LEIDEN_LM_CLASSES=32 SCAMD_LEIDEN_RF_CLASSES=32
} | tee "$OUT/knobs.log"
