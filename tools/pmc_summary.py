"""Per-kernel averages of the rocprofv3 --pmc passes written by tools/profile.sh."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (stdlib + numpy only at import time)

out_dir, prec = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]  # the extra bench.py arguments the passes ran with (tools/profile.sh)
is_bench = prec in ("fp32", "bf16x3", "f16x3", "f16f8")  # else: the passes ran another program (tools/pmc_gemm.sh)
b_args = bench.parse(["--precision", prec] + extra) if is_bench else None
rows = (b_args.rows or 50000) if is_bench else 0
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
# PMC_CYCLE=n: a call is n dispatches of one kernel (the k chunks of a k = 7 contraction): keep them apart as name#i
cycle = int(os.environ.get("PMC_CYCLE", "0"))
for path in sorted(glob.glob(os.path.join(out_dir, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path, newline="") as fh:
        seen = defaultdict(int)
        rows_in_order = sorted(csv.DictReader(fh), key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
        for row in rows_in_order:
            name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*$", "", name)  # drop the argument list of demangled names
            if cycle and "gemm" in name:
                key = (name, row["Counter_Name"])
                name = "%s#chunk%d" % (name, seen[key] % cycle)
                seen[key] += 1
            cell = acc[name][row["Counter_Name"]]
            cell[0] += float(row["Counter_Value"])
            cell[1] += 1
if is_bench:
    print("# rocprofv3 --pmc passes (separate runs, --kernel-trace only) of: python3 bench.py --steps 2 --warmup 1 "
          "--no-cpu-baseline --precision {} {}".format(prec, " ".join(extra)).rstrip())
    # what bench.py's pmc_traffic() matches before it quotes a number from this file
    generic = not (len(b_args.alphabet) == 4 and len(set(b_args.alphabet)) == 4)
    print("# workload: " + bench.workload_key(rows, b_args.length, b_args.k, prec, 1) + (" alphabet=" + b_args.alphabet if generic else ""))
else:
    print("# rocprofv3 --pmc passes (separate runs, --kernel-trace only): " + " ".join([prec] + extra))
print("# kernel_symbols_sha256: " + bench.kernel_symbols_sha256(os.path.join(bench.ROOT, "seekr_amd", "libseekr_hip.so")))
print("# per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE counts 1/2 of wide "
      "coalesced reads, see MI355X_MICROARCH.md)")
for name in sorted(acc):
    print(name)
    for counter in sorted(acc[name]):
        total, n = acc[name][counter]
        print("    {:<34s} {:.6g}".format(counter, total / max(n, 1)))
