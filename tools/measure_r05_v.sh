#!/bin/bash
# Round-5 GPU call: row chunks per tile pair of the Gram kernel (136 pairs x chunks workgroups, one workgroup per CU):
# does a grid that is a whole number of rounds of 256 workgroups (15 chunks = 2040) beat the default 16 (2176 = 8.5 rounds)?
set -u
TAG="${1:-r05v}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build rc=$?"
for C in 16 15 13 11 17 19 30 32 16; do
  echo "[chunks $C] $(SCAMD_GRAM_CHUNKS=$C timeout -k 5 200 python tools/pca_stage_probe.py 2>&1 | grep -E '^(pca_fit|gram)' | tr '\n' ' ')" | tee -a "$OUT/gram_chunks.log"
done
