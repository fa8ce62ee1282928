#!/usr/bin/env python3
"""Golden vectors G8 — a FASTA file whose FIRST line is not a header — by RUNNING THE REFERENCE (same
conventions as make_golden.py; build container only, no-op without /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -W ignore /root/repo/tests/golden/make_golden_g8.py

The reference's Reader keeps the entries in encounter order ([seq0, h1, s1, ...], fasta_reader.py:47-63) and
slices that list as it stands (:70-78), so `get_seqs()` returns what stands at the odd positions — here the
header lines — and BasicCounter counts those strings.  Stored: the file text, the three reader lists and the
raw k = 2 counts the reference produces from the file.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
TEXT = "ACGTACGTTTGA\nggcc\n>first header GATTACA\nAAAACCCC\nGGGGTTTT\n>second\nacgtnacgt\n"


def main():
    if not os.path.isdir(os.path.join(REF, "seekr")):
        print("reference not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from seekr.fasta_reader import Reader
    from seekr.kmer_counts import BasicCounter
    path = "/tmp/g8_headerless.fa"
    with open(path, "w") as fh:
        fh.write(TEXT)
    out = {"text": TEXT, "lines": Reader(path).get_lines(), "headers": Reader(path).get_headers(),
           "seqs": Reader(path).get_seqs()}
    with contextlib.redirect_stderr(io.StringIO()):
        c = BasicCounter(path, k=2, mean=False, std=False, log2="Log2.none", silent=True)
        c.get_counts()
    out["counter_seqs"] = list(c.seqs)
    out["raw_k2"] = [[float(v) for v in row] for row in c.counts]
    out["raw_k2_bits"] = [[int(v) for v in row] for row in np.asarray(c.counts, np.float32).view(np.uint32)]
    with open(os.path.join(HERE, "g8_headerless.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote g8_headerless.json:", out["headers"], out["seqs"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
