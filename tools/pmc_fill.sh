#!/bin/bash
# PMC passes over the fused fill kernel alone (tools/fill_bench.py), each in its own rocprofv3 run.
#   gpurun -- 'bash tools/pmc_fill.sh TAG [ENV=VAL ...]'
set -u
TAG=${1:-fill}
shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
             "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    # shellcheck disable=SC2086
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/pmc$i" -o "$TAG" -- \
        python3 "$REPO/tools/fill_bench.py" > "$OUT/pmc$i.log" 2> "$OUT/pmc$i.err"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" fill | grep -A30 "operand_fill" | grep -v "^__amd" > "$OUT/${TAG}_pmc_summary.txt"
cat "$OUT/${TAG}_pmc_summary.txt"
