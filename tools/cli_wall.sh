#!/bin/bash
# Wall time of the console commands as separate processes (interpreter start, imports, HIP init included).
set -e
D=$(mktemp -d /tmp/cliwall.XXXX)
python3 - "$D" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from seekr_amd.synthetic import synthetic_ascii
d = sys.argv[1]
blob, off = synthetic_ascii(2, 50000, 2000)
with open(os.path.join(d, "big.fa"), "wb") as fh:
    for i in range(50000):
        fh.write(b">s%d\n" % i); fh.write(bytes(blob[off[i]:off[i + 1]])); fh.write(b"\n")
blob, off = synthetic_ascii(3, 5000, 2000)
with open(os.path.join(d, "small.fa"), "wb") as fh:
    for i in range(5000):
        fh.write(b">t%d\n" % i); fh.write(bytes(blob[off[i]:off[i + 1]])); fh.write(b"\n")
PY
run() { local t0=$(date +%s%N); "$@" > /dev/null 2>&1; local t1=$(date +%s%N); printf "%6d ms  %s\n" "$(( (t1 - t0) / 1000000 ))" "${*//$D\//}"; }
KC="python3 -c 'from seekr_amd.console_scripts import console_kmer_counts as m; m()'"
PE="python3 -c 'from seekr_amd.console_scripts import console_pearson as m; m()'"
run python3 -c "import numpy"
run python3 -c "import seekr_amd.console_scripts"
run bash -c "$KC $D/small.fa -o $D/s.npy -b -rl"
run bash -c "$KC $D/big.fa -o $D/c.npy -b -rl"
run bash -c "$KC $D/small.fa -o $D/s.csv"
run bash -c "$PE $D/s.npy $D/s.npy -o $D/r.npy -bi -bo"
run bash -c "$PE $D/s.csv $D/s.csv -o $D/r2.npy -bo"
rm -rf "$D"
