"""Counting time of long sequences (cut into 8 192-window tiles spread over the chip) next to the per-base time of
2 kb rows: one 5 Mbase sequence, one 50 Mbase sequence, a 5 Mbase homopolymer, and the same total as 2 kb rows.

    python tools/long_seq_bench.py [-k 6]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-k", type=int, default=6)
args = ap.parse_args()
ctx = _lib.default_context()
rng = np.random.default_rng(1)
letters = np.frombuffer(b"ACGT", dtype=np.uint8)


def timed(blob, offsets, rounds=12):
    packed = _lib.PackedSeqs.from_buffer(ctx, blob, offsets, "AGTC")
    out = ctx.empty(len(offsets) - 1, 4 ** args.k)
    ts = []
    for _ in range(rounds):
        ctx.prof_reset()
        ctx.prof_enable(True)
        _lib.count_per_kb(ctx, packed, args.k, out=out)
        ctx.sync()
        ctx.prof_enable(False)
        ts.append(sum(ctx.prof_query(n)[0] for n in ctx.prof_names() if n.startswith("count")))
    return float(np.median(ts[2:]))


def one(name, blob):
    ms = timed(blob, np.array([0, len(blob)], dtype=np.int64))
    print("%-36s %9.1f Mbases  %.4f ms  %.1f Gbases/s  %.2f ps/base" % (name, len(blob) / 1e6, ms, len(blob) / ms / 1e6, ms * 1e9 / len(blob)))
    return ms


n2k = 2500
rows = letters[rng.integers(0, 4, size=(n2k, 2000))].reshape(-1)
ms = timed(rows, np.arange(n2k + 1, dtype=np.int64) * 2000)
print("%-36s %9.1f Mbases  %.4f ms  %.1f Gbases/s  %.2f ps/base (each base also pays 8.2 B of row output)"
      % ("2 500 rows of 2 kb (5 Mbases)", n2k * 2e-3, ms, n2k * 2000 / ms / 1e6, ms * 1e9 / (n2k * 2000)))
big = 50000
rows = letters[rng.integers(0, 4, size=(big, 2000))].reshape(-1)
ms = timed(rows, np.arange(big + 1, dtype=np.int64) * 2000)
print("%-36s %9.1f Mbases  %.4f ms  %.1f Gbases/s  %.2f ps/base" % ("50 000 rows of 2 kb", big * 2e-3, ms, big * 2000 / ms / 1e6, ms * 1e9 / (big * 2000)))
one("one random sequence of 5 Mbases", letters[rng.integers(0, 4, size=5_000_000)])
one("one random sequence of 50 Mbases", letters[rng.integers(0, 4, size=50_000_000)])
one("one homopolymer of 5 Mbases", np.full(5_000_000, ord("T"), np.uint8))
one("one 90 kb sequence (Airn-sized)", letters[rng.integers(0, 4, size=90_000)])
