#!/bin/bash
# PMC passes over the contraction alone (tools/gemm_bench.py), each in its own rocprofv3 run.
#   gpurun -- 'bash tools/pmc_gemm.sh TAG "--rows 50000 --mode self" [ENV=VAL ...]'
set -u
TAG=${1:-gemm}
ARGS=${2:---rows 50000 --mode self}
shift; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
i=0
for group in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"; do
    i=$((i + 1))
    # shellcheck disable=SC2086
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/pmc$i" -o "$TAG" -- \
        python3 "$REPO/tools/gemm_bench.py" $ARGS --rounds 3 base= > "$OUT/pmc$i.log" 2> "$OUT/pmc$i.err"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" gemm | grep -A${PMC_LINES:-24} "split16" > "$OUT/${TAG}_pmc_summary.txt"
cat "$OUT/${TAG}_pmc_summary.txt"
tail -2 "$OUT"/pmc*.err | grep -i "error\|invalid\|not" | head -5
