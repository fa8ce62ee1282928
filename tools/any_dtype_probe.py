"""How long the any-dtype normalisation methods take on a matrix of CSV size (20 000 x 4 096 float64 = 655 MB), next to
numpy doing the same on the host — they are the drop-in surface, not the hot path, but they must not be pathological."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd.kmer_counts import BasicCounter  # noqa: E402

rng = np.random.default_rng(0)
for dtype, shape in ((np.float64, (20000, 4096)), (np.float16, (20000, 4096)), (np.int32, (20000, 4096)), (np.float64, (3000, 256))):
    x = (rng.poisson(0.9, size=shape) * 0.5).astype(dtype)
    c = BasicCounter(k=1, silent=True)
    c.counts = x.copy()
    t0 = time.perf_counter()
    if dtype != np.int32:
        c.center()
        t1 = time.perf_counter()
        c.standardize()
        t2 = time.perf_counter()
    else:
        c.log2_norm()
        t1 = t2 = time.perf_counter()
    y = x.copy()
    t3 = time.perf_counter()
    with np.errstate(all="ignore"):
        if dtype != np.int32:
            y -= np.mean(y, axis=0)
            y /= np.std(y, axis=0)
        else:
            y += 1
            y = np.log2(y)
    t4 = time.perf_counter()
    same = np.array_equal(c.counts, y, equal_nan=True) if dtype != np.int32 else np.allclose(c.counts, y, rtol=1e-12)
    print("%-8s %-14s device %.3f + %.3f s   numpy %.3f s   identical: %s" % (np.dtype(dtype).name, shape, t1 - t0, t2 - t1, t4 - t3, same))
