"""The any-alphabet counter alone: 50 000 x 2 kb sequences over ACGTN, k = 6 (5^6 = 15 625 columns, the histogram in the LDS),
float32 per-kb rows; timed by the library's own HIP events.  bytes = rows x columns x 4 written + the characters read."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402

ctx = _lib.default_context()
cases = [("ACGTN", 6, 50000, 2000), ("ACGTN", 5, 50000, 2000), ("ARNDCQEGHILKMFPSTWYV", 3, 50000, 2000), ("ACGTN", 6, 5000, 20000),
         ("ACDEFGHIKL", 4, 50000, 2000), ("ACDEFGHIKLM", 4, 50000, 2000), ("ACGTN", 6, 50000, 500), ("ACDEFGH", 5, 30000, 5000),
         ("ACGTN", 7, 20000, 2000), ("ACDEFG", 6, 30000, 2000)]  # 78 125 / 46 656 bins: three / two bin ranges a row
if len(sys.argv) > 1:  # only the cases named by index (PMC runs: tools/pmc_count_generic.sh)
    cases = [cases[int(a)] for a in sys.argv[1:]]
rng = np.random.default_rng(0)
for alphabet, k, n, length in cases:
    letters = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    blob = letters[rng.integers(0, len(letters), size=n * length)]
    offsets = np.arange(n + 1, dtype=np.int64) * length
    a = _lib.AsciiSeqs(ctx, blob.tobytes(), offsets)
    cols = len(alphabet) ** k
    out = ctx.empty(n, cols)
    ts = []
    for _ in range(7):
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        _lib.count_generic_dev(ctx, a, alphabet, k, out=out)
        ctx.sync()
        ctx.prof_enable(False)
        ts.append(sum(ctx.prof_query(nm)[0] for nm in ctx.prof_names() if nm.startswith("count_generic")))
    t = float(np.median(ts[2:]))
    gb = (n * cols * 4 + n * length) / 1e9
    print("%-22s k %d  %6d x %6d nt -> %6d columns  %.3f ms  %.2f GB -> %.0f GB/s = %.3f of 8 TB/s" % (
        alphabet, k, n, length, cols, t, gb, gb / t * 1e3, gb / t * 1e3 / 8000), flush=True)
    a.free()
