#!/usr/bin/env python3
"""Golden vectors G9 — WIDE rows from alphabets other than four letters — by RUNNING THE REFERENCE (build container only,
no-op without /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -W ignore /root/repo/tests/golden/make_golden_g9.py

Round 5 found the operand fill wrong at row widths that are multiples of 8 but not of 32 above 8 192 columns (10^4 ...):
widths no golden set, test or fuzzer had.  These pin three such shapes to the reference's own output: 10 letters at k = 4
(10 000 columns), 5 letters at k = 6 (15 625) and 7 letters at k = 5 (16 807: past the 16 384 columns one LDS histogram
used to hold).  Stored per case: the reference's raw count matrix (kmer_counts.py:140-151; compresses well), of the
mean-centred Log2.post matrix (kmer_counts.py:201-209) its float64 sum and 64 seeded cells, and the reference's Pearson
matrices of both (pearson.py:32-44).  Inputs are regenerated from the seed.
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("ten4", "ACDEFGHIKL", 4), ("acgtn6", "ACGTN", 6), ("seven5", "ACDEFGH", 5)]
N_SEQS = 14


def sequences(name, alphabet):
    rng = np.random.default_rng(sum(map(ord, name)))
    letters = list(alphabet) + ["X"]  # a letter outside the alphabet now and then
    p = np.array([1.0] * len(alphabet) + [0.02])
    seqs = ["".join(rng.choice(letters, size=int(rng.integers(3000, 9000)), p=p / p.sum())) for _ in range(N_SEQS)]
    seqs[3] = seqs[2][5:] + seqs[2][:5]  # a near-copy: r close to 1
    return seqs


def sampled_cells(name, shape):
    rng = np.random.default_rng(len(name))
    return rng.integers(0, shape[0], 64), rng.integers(0, shape[1], 64)


def main():
    if not os.path.isdir(os.path.join(REF, "seekr")):
        print("reference not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from seekr.kmer_counts import BasicCounter  # noqa: E402
    from seekr.pearson import pearson  # noqa: E402

    out = {}
    for name, alphabet, k in CASES:
        seqs = sequences(name, alphabet)
        mats = {}
        for tag, kw in (("raw", dict(mean=False, std=False, log2="Log2.none")), ("post", dict(mean=True, std=False, log2="Log2.post"))):
            c = BasicCounter(k=k, alphabet=alphabet, silent=True, **kw)
            c.seqs = list(seqs)
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                c.get_counts()
            mats[tag] = c.counts
            with np.errstate(all="ignore"):
                out["%s_r_%s" % (name, tag)] = pearson(c.counts, c.counts)
        out["%s_raw" % name] = mats["raw"]
        rows, cols = sampled_cells(name, mats["post"].shape)
        out["%s_post_sum" % name] = np.array(mats["post"].astype(np.float64).sum())
        out["%s_post_cells" % name] = mats["post"][rows, cols]
    np.savez_compressed(os.path.join(HERE, "g9_wide_alphabets.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
