"""Wall time of the console commands end to end on the GPU box (files in, files out), config-2
shaped input: seekr_kmer_counts with its three output formats, seekr_pearson on the labelled CSV
it wrote (default input format) and on the .npy."""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import console_scripts as cs  # noqa: E402
from seekr_amd.synthetic import synthetic_ascii  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=50000)
ap.add_argument("--pearson-rows", type=int, default=10000)
ap.add_argument("--length", type=int, default=2000)
args = ap.parse_args()
d = tempfile.mkdtemp(dir="/tmp")


def write_fasta(path, n):
    blob, off = synthetic_ascii(2, n, args.length)
    with open(path, "wb") as fh:
        for i in range(n):
            fh.write(b">s%d|synthetic transcript %d\n" % (i, i))
            fh.write(bytes(blob[off[i]:off[i + 1]]))
            fh.write(b"\n")


def timed(label, fn):
    t0 = time.time()
    fn()
    dt = time.time() - t0
    print("%-64s %7.2f s" % (label, dt), flush=True)
    return dt


fa, fa_small = os.path.join(d, "big.fa"), os.path.join(d, "small.fa")
write_fasta(fa, args.rows)
write_fasta(fa_small, args.pearson_rows)
print("FASTA: %d x %d nt = %.0f Mbases" % (args.rows, args.length, args.rows * args.length / 1e6))
cs._run_kmer_counts(fa_small, os.path.join(d, "warm.npy"), 6, True, True, True, "Log2.post", True, None, None, "AGTC")  # warm-up
mb = args.rows * args.length / 1e6
t = timed("seekr_kmer_counts -b -rl          (.npy)", lambda: cs._run_kmer_counts(
    fa, os.path.join(d, "c.npy"), 6, True, True, True, "Log2.post", True, None, None, "AGTC"))
print("    -> %.0f Mbases/s file to file" % (mb / t))
t = timed("seekr_kmer_counts -rl             (plain CSV, %1.6f)", lambda: cs._run_kmer_counts(
    fa, os.path.join(d, "c_plain.csv"), 6, False, True, True, "Log2.post", True, None, None, "AGTC"))
print("    -> %.0f Mbases/s, %.2f GB of text" % (mb / t, os.path.getsize(os.path.join(d, "c_plain.csv")) / 1e9))
t = timed("seekr_kmer_counts                 (labelled CSV, the default)", lambda: cs._run_kmer_counts(
    fa, os.path.join(d, "c_lab.csv"), 6, False, True, True, "Log2.post", False, None, None, "AGTC"))
print("    -> %.0f Mbases/s, %.2f GB of text" % (mb / t, os.path.getsize(os.path.join(d, "c_lab.csv")) / 1e9))
for f in ("c_plain.csv", "c_lab.csv"):
    os.remove(os.path.join(d, f))
small_csv, small_npy = os.path.join(d, "s.csv"), os.path.join(d, "s.npy")
cs._run_kmer_counts(fa_small, small_csv, 6, False, True, True, "Log2.post", False, None, None, "AGTC")
cs._run_kmer_counts(fa_small, small_npy, 6, True, True, True, "Log2.post", True, None, None, "AGTC")
n = args.pearson_rows
t = timed("seekr_pearson s.csv s.csv -bo     (%d rows, labelled CSV in, float64)" % n,
          lambda: cs._run_pearson(small_csv, small_csv, os.path.join(d, "r64.npy"), False, True))
t = timed("seekr_pearson s.npy s.npy -bi -bo (%d rows, float32)" % n,
          lambda: cs._run_pearson(small_npy, small_npy, os.path.join(d, "r32.npy"), True, True))
r64, r32 = np.load(os.path.join(d, "r64.npy"), mmap_mode="r"), np.load(os.path.join(d, "r32.npy"), mmap_mode="r")
print("    float64-from-CSV vs float32 r: max |diff| %.2e" % float(np.abs(np.asarray(r64[:2000]) - np.asarray(r32[:2000])).max()))
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
