"""The counting kernels at MANY sequence lengths (round 5; the companion of tools/width_sweep.py).  The counters choose
their path by the length of a sequence — shorter than k, one LDS chunk (4 096 characters of the any-alphabet counter), one
tile of 8 192 windows of the 2-bit counter, several — and by the width of a row (histogram in the LDS up to 16 384 bins,
in HBM above).  This walks the lengths instead of listing them:

  every length 0 .. 300 (k - 1, on which the reference divides by zero, left out), the neighbours (- k - 3 .. + k + 3) of
  every multiple of 1 024 up to 20 480 and of 32 768 / 65 536 / 131 072, seeded random lengths up to 100 000;

every sequence in ONE ragged set per (alphabet, k), letters outside the alphabet sprinkled in and put on the first / last
position and across chunk boundaries.  Integer counts BIT-EXACT against the C oracle (oracle/c_oracle: kmer_counts.py:140-151),
the per-kb float32 rows bit-exact too.

    python tools/length_sweep.py [--quick]

Exit code 1 if any (alphabet, k) differs.  Needs a real MI355X.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [("AGTC", k) for k in range(1, 10)] + [("ACGTN", k) for k in (1, 2, 3, 5, 6, 7)] + [("AT", 3), ("AT", 12), ("AT", 15),
         ("ARNDCQEGHILKMFPSTWYV", 2), ("ARNDCQEGHILKMFPSTWYV", 3), ("ACDEFGHIKL", 4), ("AGTA", 4), ("T", 3),
         ("ACDEFGH", 5), ("ACG", 9), ("ACGTRYNK", 5), ("ACDEFG", 6)]  # 16 807 / 19 683 / 32 768 bins: the LDS above 16 384; 46 656: HBM


def lengths_for(k, quick, rng):
    out = set(range(0, 301))
    marks = list(range(1024, 20481, 1024)) + [32768, 65536, 131072]
    if quick:
        marks = [1024, 2048, 4096, 8192, 12288, 16384, 65536]
    for m in marks:
        for d in range(-k - 3, k + 4):
            out.add(m + d)
    out.update(int(v) for v in rng.integers(301, 100000, 12 if quick else 60))
    out.discard(k - 1)
    return sorted(out)


def sequences(alphabet, k, quick, seed):
    rng = np.random.default_rng([seed, len(alphabet), k])
    letters = np.frombuffer("".join(dict.fromkeys(alphabet)).encode(), dtype=np.uint8)
    foreign = np.frombuffer(b"Nxz", dtype=np.uint8) if "N" not in alphabet else np.frombuffer(b"Xqz", dtype=np.uint8)
    seqs = []
    for i, n in enumerate(lengths_for(k, quick, rng)):
        s = letters[rng.integers(0, len(letters), size=n)].copy()
        style = i % 5
        if n and style == 1:
            s[rng.random(n) < 0.004] = foreign[0]
        elif n and style == 2:
            s[0] = foreign[1]
            s[-1] = foreign[2]
        elif n > 4200 and style == 3:  # runs of foreign letters across the chunk / tile boundaries
            for edge in (4096, 8192, 8192 + k - 1):
                if edge + 3 < n:
                    s[edge - 2:edge + 3] = foreign[0]
        elif n and style == 4:
            s[:] = letters[0]  # a homopolymer: every window in one bin
        seqs.append(s.tobytes().decode("latin-1"))
    return seqs


def sweep(quick=False, seed=1, verbose=True):
    from oracle import c_oracle as co
    from seekr_amd import _lib as L
    ctx = L.default_context()
    bad = {}
    for alphabet, k in CASES:
        seqs = sequences(alphabet, k, quick, seed)
        blob, offsets = co.seqs_to_blob(seqs)
        want = co.count_u32(blob, offsets, k, alphabet)
        lens = [len(s) for s in seqs]
        want_kb = co.per_kb_f32(want, lens, k)
        problems = []
        if alphabet == "AGTC":
            packed = ctx.pack(seqs, "AGTC")
            got = L.count_u32(ctx, packed, k).to_numpy()
            got_kb = L.count_per_kb(ctx, packed, k).to_numpy()
        else:
            got = L.count_generic(ctx, seqs, alphabet, k, np.uint32).to_numpy()
            got_kb = L.count_generic(ctx, seqs, alphabet, k, np.float32).to_numpy()
        if not np.array_equal(got, want):
            rows = np.nonzero((got != want).any(axis=1))[0]
            problems.append("integer counts differ in %d rows, lengths %s" % (len(rows), [lens[r] for r in rows[:8]]))
        if not np.array_equal(got_kb.view(np.uint32), want_kb.view(np.uint32)):
            rows = np.nonzero((got_kb.view(np.uint32) != want_kb.view(np.uint32)).any(axis=1))[0]
            problems.append("per-kb rows differ in %d rows, lengths %s" % (len(rows), [lens[r] for r in rows[:8]]))
        if problems:
            bad[(alphabet, k)] = problems
            print("%-22s k %2d  FAIL  %s" % (alphabet, k, "; ".join(problems)), file=sys.stderr, flush=True)
        elif verbose:
            print("%-22s k %2d  %4d lengths (0 .. %d), %d columns  ok" % (alphabet, k, len(seqs), max(lens), want.shape[1]), flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    bad = sweep(args.quick, args.seed)
    print("%d (alphabet, k) cases, %d failing" % (len(CASES), len(bad)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
