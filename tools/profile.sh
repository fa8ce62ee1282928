#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/: run ON THE GPU BOX from the repo root,
#   gpurun -- 'bash tools/profile.sh r3_f16x3 f16x3 [extra bench.py arguments, e.g. --rows 200000]'
# 1. kernel trace + stats of the default bench command (per-kernel average durations);
# 2. PMC passes, each in its own run with --kernel-trace only (never with sys/hip/hsa traces),
#    summarised per kernel by tools/pmc_summary.py.  Every pass runs under `timeout`: a counter
#    set the hardware cannot schedule aborts the program and leaves rocprofv3 waiting forever.
set -u
TAG=${1:-r3_f16x3}
PREC=${2:-f16x3}
shift 2 2>/dev/null || true
EXTRA="$*"
T_TRACE=${T_TRACE:-200}
T_PMC=${T_PMC:-150}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# shellcheck disable=SC2086
timeout -k 10 $T_TRACE rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$TAG" -- \
    python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --precision "$PREC" $EXTRA \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/trace.err"
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
    i=$((i + 1))
    # shellcheck disable=SC2086
    timeout -k 10 $T_PMC rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/pmc$i" -o "$TAG" -- \
        python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --precision "$PREC" $EXTRA \
        > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err"
done
cd "$REPO"
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/${TAG}_kernel_stats.csv" \;
# shellcheck disable=SC2086
python3 tools/pmc_summary.py "$OUT" "$PREC" $EXTRA > "$OUT/${TAG}_pmc_summary.txt"
ls -la "$OUT"
head -12 "$OUT/${TAG}_kernel_stats.csv"
