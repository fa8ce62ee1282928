#!/usr/bin/env python3
"""Golden vectors G12 — center() / standardize() on hand-assigned count matrices that are COLUMN-MAJOR (what
`DataFrame.values` of a read CSV is), a single column, or a strided view of a column-major array — by RUNNING THE
REFERENCE (same conventions as make_golden.py; build container only, no-op without /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -W ignore /root/repo/tests/golden/make_golden_g12.py

numpy reduces such a matrix column by column, pairwise, in pieces of 8 192 elements (kmer_counts.py:168,174 call np.mean /
np.std on whatever layout `self.counts` has): other bits than the row-after-row sums of a C-ordered matrix.  For float32 /
float64 / float16 / int32 / uint8 matrices of 300 x 7 and 9 001 x 3 (inputs: `matrix()` below, seeded — the tests import it)
the fixture holds what the reference leaves behind: SHA-256 of the mean / std vectors and of the centred-then-standardised
matrix (C-order bytes), their dtypes, and for integer matrices numpy's exception with the attribute already replaced.
"""
import contextlib
import hashlib
import io
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = [(300, 7), (9001, 3)]
DTYPES = ["float32", "float64", "float16", "int32", "uint8"]
LAYOUTS = ["fortran", "fortran_strided", "single_column"]


def matrix(dtype, shape, layout, seed=12):
    rng = np.random.default_rng([seed, shape[0], shape[1]])
    n = rng.poisson(0.9, size=(2 * shape[0], shape[1]))
    dt = np.dtype(dtype)
    m = (n * (1000.0 / 1995.0)).astype(dt) if dt.kind == "f" else n.astype(dt)
    if layout == "fortran":
        return np.asfortranarray(m[:shape[0]])
    if layout == "fortran_strided":
        return np.asfortranarray(m)[::2]         # rows 0, 2, 4, ... of a column-major array: still reduced column by column
    return np.ascontiguousarray(m[:shape[0], :1])  # one column: C-contiguous AND reduced as a column


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run(counter_cls, dtype, shape, layout):
    """-> dict: what center() then standardize() leave (or raise) on this input."""
    out = {}
    for first in ("center", "standardize"):
        c = counter_cls(silent=True, k=1)
        c.counts = matrix(dtype, shape, layout)
        held = c.counts
        try:
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                getattr(c, first)()
                if first == "center":
                    c.standardize()
            res = {"counts": digest(c.counts), "dtype": np.asarray(c.counts).dtype.name, "in_place": c.counts is held}
        except Exception as e:  # noqa: BLE001
            res = {"exception": type(e).__name__, "message": str(e)}
        for attr in ("mean", "std"):
            v = getattr(c, attr)
            if isinstance(v, np.ndarray):
                res[attr] = {"sha256": digest(v), "dtype": v.dtype.name}
        out[first] = res
    return out


def main():
    if not os.path.isdir(REF):
        print("no reference checkout: nothing to do")
        return
    sys.path.insert(0, REF)
    from seekr.kmer_counts import BasicCounter
    meta = {"generator": "tests/golden/make_golden_g12.py", "numpy": np.__version__, "cases": {}}
    for dtype in DTYPES:
        for shape in SHAPES:
            for layout in LAYOUTS:
                meta["cases"]["%s_%dx%d_%s" % (dtype, shape[0], shape[1], layout)] = run(BasicCounter, dtype, shape, layout)
    with open(os.path.join(HERE, "g12_column_major.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print("wrote", len(meta["cases"]), "cases")


if __name__ == "__main__":
    main()
