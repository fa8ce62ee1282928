"""Per-kernel averages of the rocprofv3 --pmc passes written by tools/profile.sh."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out_dir, prec = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sorted(glob.glob(os.path.join(out_dir, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*$", "", name)  # drop the argument list of demangled names
            cell = acc[name][row["Counter_Name"]]
            cell[0] += float(row["Counter_Value"])
            cell[1] += 1
print("# rocprofv3 --pmc passes (separate runs, --kernel-trace only) of: python3 bench.py --steps 2 --warmup 1 "
      "--no-cpu-baseline --precision {}".format(prec))
print("# per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE counts 1/2 of wide "
      "coalesced reads, see MI355X_MICROARCH.md)")
for name in sorted(acc):
    print(name)
    for counter in sorted(acc[name]):
        total, n = acc[name][counter]
        print("    {:<34s} {:.6g}".format(counter, total / max(n, 1)))
