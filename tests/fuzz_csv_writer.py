"""Differential fuzzing of the native labelled-CSV / plain-CSV / .npy writers (host variants, no GPU) against
pandas.DataFrame.to_csv, numpy.savetxt(fmt="%1.6f") and numpy.save: random values (specials, subnormals,
powers of two and ten, integers) and random labels (commas, quotes, newlines, blanks, unicode)."""
import io
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekr_amd import _lib  # noqa: E402


def values(rng, rows, cols, dtype):
    style = rng.integers(0, 6)
    if style == 0:
        x = rng.normal(0, 3, (rows, cols))
    elif style == 1:
        x = rng.normal(0, 1, (rows, cols)) * 10.0 ** rng.integers(-40, 39, (rows, cols))
    elif style == 2:
        x = rng.integers(-5, 6, (rows, cols)).astype(np.float64) * rng.choice([1.0, 0.5, 0.25, 1000.0 / 1995])
    elif style == 3:
        x = 2.0 ** rng.integers(-140, 120, (rows, cols)) * rng.choice([1.0, -1.0], (rows, cols))
    elif style == 4:
        x = 10.0 ** rng.integers(-40, 38, (rows, cols))
    else:
        x = rng.normal(0, 3, (rows, cols))
        x[rng.integers(0, rows, 3), rng.integers(0, cols, 3)] = rng.choice([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, 3.4e38], 3)
    with np.errstate(over="ignore"):
        x = x.astype(dtype)
    if rng.integers(0, 2):  # land on neighbours of round numbers
        x = np.nextafter(x, rng.choice([-np.inf, np.inf], x.shape).astype(dtype))
    return x


def label(rng, i):
    pool = [">s%d" % i, ">ENST%d.1|a,b|" % i, 'he said "x" %d' % i, " lead%d" % i, "trail%d " % i, "multi\nline%d" % i, "cr\rret%d" % i,
            "é%d" % i, "", "plain%d" % i, "'single%d'" % i, "tab\t%d" % i, "#hash%d" % i, '"%d"' % i, ",%d" % i, str(i), "%d.5" % i]
    return pool[int(rng.integers(0, len(pool)))]


def fuzz(seed, budget_s=30.0, max_cases=10 ** 9):
    rng = np.random.default_rng(seed)
    d = tempfile.mkdtemp()
    path = os.path.join(d, "out")
    t0, n_cases = time.time(), 0
    while time.time() - t0 < budget_s and n_cases < max_cases:
        rows, cols = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        dtype = np.float32 if rng.integers(0, 3) else np.float64
        x = values(rng, rows, cols, dtype)
        idx = [label(rng, i) for i in range(rows)]
        colnames = [label(rng, 50 + j) for j in range(cols)]
        try:
            _lib.save_csv_labelled(path + ".csv", x, idx, colnames)
            want = pd.DataFrame(x, idx, colnames).to_csv()
            got = open(path + ".csv", newline="", encoding="utf-8").read()
            assert got == want, ("labelled csv", got[:300], want[:300])
            finite = np.where(np.isfinite(x), x, 0).astype(dtype)
            _lib.save_csv(path + "p.csv", finite)
            buf = io.StringIO()
            np.savetxt(buf, finite, delimiter=",", fmt="%1.6f")
            assert open(path + "p.csv").read() == buf.getvalue(), "plain csv"
            _lib.save_npy(path + ".npy", x)
            buf = io.BytesIO()
            np.save(buf, x)
            assert open(path + ".npy", "rb").read() == buf.getvalue(), "npy"
        except AssertionError:
            out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(out, exist_ok=True)
            np.savez(os.path.join(out, "fuzz_writer_fail_%d_%d.npz" % (seed, n_cases)), x=x, idx=np.array(idx, dtype=object),
                     cols=np.array(colnames, dtype=object))
            raise
        n_cases += 1
    return n_cases


if __name__ == "__main__":
    n = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
    print("writer fuzz ok: %d matrices" % n)
