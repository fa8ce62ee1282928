"""Differential fuzzing of the native labelled-CSV reader (csv_read.hip, host code only — no GPU) against
pandas.read_csv(path, index_col=0): whenever the native reader accepts a file, values (bit for bit, float64),
index and column labels must be what pandas returns; it may decline (None), never differ."""
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekr_amd import _lib  # noqa: E402


def number_text(rng):
    style = rng.integers(0, 12)
    x = float(np.float32(rng.normal(0, 3)))
    if style == 0:
        return repr(float(np.float32(x)))
    if style == 1:
        return "%.6f" % x
    if style == 2:
        return str(np.float32(x))                       # numpy's shortest float32 repr: what the count files hold
    if style == 3:
        return "%.18e" % x
    if style == 4:
        return str(int(rng.integers(-10 ** 6, 10 ** 6)))
    if style == 5:
        return rng.choice(["", "nan", "NaN", "inf", "-inf", "0", "-0.0", "0.0", "1e5", "1E-5", "+3.5", ".5", "5.", "1e400", "1e-400"])
    if style == 6:
        return "%.*e" % (int(rng.integers(0, 20)), x * 10.0 ** int(rng.integers(-30, 30)))
    if style == 7:
        return "%.*f" % (int(rng.integers(0, 18)), x)
    if style == 8:
        return "".join(rng.choice(list("0123456789"), int(rng.integers(1, 25)))) + "." + "".join(rng.choice(list("0123456789"), int(rng.integers(0, 25))))
    if style == 9:
        return " %s" % repr(x) if rng.integers(0, 2) else "%s " % repr(x)   # stray blanks
    if style == 10:
        return '"%s"' % repr(x)                                              # a quoted number
    return repr(x)


def label_text(rng, i):
    style = rng.integers(0, 8)
    if style == 0:
        return ">ENST%05d.1|gene-%d|" % (i, i)
    if style == 1:
        return '">seq %d, with a comma"' % i
    if style == 2:
        return '">he said ""hi"" %d"' % i
    if style == 3:
        return "s%d" % i
    if style == 4:
        return ">s%d " % i
    if style == 5:
        return str(i)            # numeric labels: pandas converts the index
    if style == 6:
        return ">é%d" % i
    return "row%d" % (i // 2)    # duplicates


def fuzz(seed, budget_s=30.0, max_cases=10 ** 9):
    rng = np.random.default_rng(seed)
    d = tempfile.mkdtemp()
    path = os.path.join(d, "f.csv")
    t0, n_cases, n_native = time.time(), 0, 0
    while time.time() - t0 < budget_s and n_cases < max_cases:
        rows, cols = int(rng.integers(0, 12)), int(rng.integers(1, 9))
        eol = "\r\n" if rng.integers(0, 5) == 0 else "\n"
        clean = rng.integers(0, 3) == 0   # a file like the ones the count command writes: the native reader must take it
        kmer_header = clean or rng.integers(0, 4) != 0
        header = [""] + (["AA", "AC", "AG", "AT", "CA", "CC", "CG", "CT"][:cols] if kmer_header else [label_text(rng, 100 + j) for j in range(cols)])
        lines = [",".join(header)]
        for i in range(rows):
            lab = (">s%d|x" % i) if clean else label_text(rng, i)
            cells = [str(np.float32(rng.normal(0, 3))) if clean else number_text(rng) for _ in range(cols)]
            if not clean and rng.integers(0, 40) == 0:
                cells = cells[:-1]   # a short row
            lines.append(",".join([lab] + cells))
        text = eol.join(lines) + (eol if rng.integers(0, 4) else "")
        with open(path, "w", newline="", encoding="utf-8") as fh:
            fh.write(text)
        try:
            got = _lib.load_csv_labelled(path)
        except Exception as e:  # noqa: BLE001 — an error is acceptable only where pandas fails too
            got = e
        try:
            frame = pd.read_csv(path, index_col=0)
            want = None
        except Exception as e:  # noqa: BLE001
            frame, want = None, e
        try:
            if isinstance(got, Exception):
                assert want is not None, ("native reader raised, pandas did not", repr(got))
            elif got is not None:
                n_native += 1
                assert frame is not None, "native reader accepted a file pandas rejects"
                vals, idx, colnames = got
                if len(frame):  # a header-only file: pandas has no values to type, the native reader returns a 0 x cols matrix
                    # (a column of integers is int64 in pandas; DataFrame.values and pearson() promote it to float64)
                    assert all(dt.kind in "fi" for dt in frame.dtypes), ("pandas did not read numeric columns", frame.dtypes.tolist())
                assert vals.shape == frame.shape, (vals.shape, frame.shape)
                assert np.array_equal(vals.view(np.uint64), np.ascontiguousarray(frame.values, dtype=np.float64).view(np.uint64)), "values differ"
                assert list(idx) == [str(x) for x in frame.index] and all(isinstance(x, str) for x in frame.index), "index differs"
                assert list(colnames) == list(frame.columns), "columns differ"
            elif clean and rows > 0:
                raise AssertionError("the native reader declined a file in the count command's own format")
        except AssertionError:
            out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "fuzz_csv_fail_%d_%d.csv" % (seed, n_cases)), "w", newline="", encoding="utf-8") as fh:
                fh.write(text)
            print(repr(text[:600]))
            raise
        n_cases += 1
    return n_cases, n_native


if __name__ == "__main__":
    n, m = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
    print("csv reader fuzz ok: %d files, %d read natively" % (n, m))
