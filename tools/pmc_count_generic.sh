#!/bin/bash
# PMC passes over the any-alphabet counter alone (tools/count_generic_bench.py), each in its own rocprofv3 run.
#   gpurun -- 'bash tools/pmc_count_generic.sh TAG CASE'   (CASE: index into the tool's list of workloads)
set -u
TAG=${1:-countgen}
CASE=${2:-0}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
    i=$((i + 1))
    # shellcheck disable=SC2086
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/pmc$i" -o "$TAG" -- \
        python3 "$REPO/tools/count_generic_bench.py" $CASE > "$OUT/pmc$i.log" 2> "$OUT/pmc$i.err"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "count_generic" not in row["Kernel_Name"]:
            continue
        acc[(row["Kernel_Name"][:60], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key in sorted(acc):
    print(key)
    for name, vals in sorted(acc[key].items()):
        print("    %-26s %.4g  (x%d)" % (name, sum(vals) / len(vals), len(vals)))
PY
