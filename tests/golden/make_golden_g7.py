#!/usr/bin/env python3
"""Golden vectors G7 — alphabets other than four distinct letters — by RUNNING THE REFERENCE
(same conventions as make_golden.py; build container only, no-op without /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -W ignore /root/repo/tests/golden/make_golden_g7.py

`BasicCounter(alphabet=...)` accepts any string (kmer_counts.py:120-122).  Stored: the inputs (seeded)
and the reference's raw and normalised count matrices for a 5-letter alphabet, a 2-letter one, a
3-letter one with a repeated letter (the dict {kmer: index} keeps the last index), a 20-letter
protein-like one, and `occurrences` on a float64 row.
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [  # name, alphabet, k, letters the sequences are drawn from
    ("acgtn", "ACGTN", 2, "ACGTN"),
    ("two", "AT", 5, "ACGT"),
    ("repeat", "AAG", 2, "AGT"),
    ("repeat4", "AGTA", 3, "ACGT"),
    ("protein", "ACDEFGHIKLMNPQRSTVWY", 2, "ACDEFGHIKLMNPQRSTVWYX"),
    ("one", "A", 3, "AC"),
]


def sequences(name, letters):
    rng = np.random.default_rng(sum(map(ord, name)))
    return ["".join(rng.choice(list(letters), size=int(rng.integers(8, 120)))) for _ in range(9)]


def main():
    if not os.path.isdir(os.path.join(REF, "seekr")):
        print("reference not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from seekr.kmer_counts import BasicCounter  # noqa: E402

    out = {}
    for name, alphabet, k, letters in CASES:
        seqs = sequences(name, letters)
        for tag, kw in (("raw", dict(mean=False, std=False, log2="Log2.none")),
                        ("pre", dict(mean=True, std=False, log2="Log2.pre")),
                        ("post", dict(mean=True, std=True, log2="Log2.post"))):
            c = BasicCounter(k=k, alphabet=alphabet, silent=True, **kw)
            c.seqs = list(seqs)
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                c.get_counts()
            out["%s_%s" % (name, tag)] = c.counts
        c = BasicCounter(k=k, alphabet=alphabet, silent=True)
        out["%s_occ" % name] = c.occurrences(np.full(len(alphabet) ** k, -1.0), seqs[0])
    np.savez_compressed(os.path.join(HERE, "g7_alphabets.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
