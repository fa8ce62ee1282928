"""Worst-case data for the split contractions: rows with very few distinct values (sparse raw
counts), where the representation residual of a value is the same in thousands of columns and
adds up instead of averaging out.  Prints, per precision, the largest error against float64 on
the diagonal and off it, and the worst ratio to the parity bar 2e-6 + 1e-5|r|."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekr_amd import _lib as L

ctx = L.Context(0)
rng = np.random.default_rng(0)
cases = {}
n, k = 2048, 4096
sparse = np.zeros((n, k), np.float32)
for i in range(n):  # 3..200 nonzero k-mers per row, raw per-kb counts of a short sequence
    nnz = rng.integers(3, 200)
    cols = rng.choice(k, nnz, replace=False)
    sparse[i, cols] = rng.integers(1, 4, nnz) * np.float32(1000.0 / rng.integers(50, 900))
cases["sparse raw counts"] = sparse
cases["binomial raw counts"] = (rng.binomial(50, 0.05, size=(n, k)) * np.float32(2.5)).astype(np.float32)
cases["two-valued rows"] = np.where(rng.random((n, k)) < 0.5, np.float32(1.0), np.float32(0.0)).astype(np.float32)
cases["gaussian"] = rng.standard_normal((n, k)).astype(np.float32)
for name, x in cases.items():
    x64 = x.astype(np.float64)
    xc = x64 - x64.mean(axis=1, keepdims=True)
    z = xc / xc.std(axis=1, keepdims=True)
    truth = z @ z.T / k
    off = ~np.eye(n, dtype=bool)
    dev = ctx.from_numpy(x)
    print(name)
    for prec in ("fp32", "bf16x3", "f16x3"):
        r = L.pearson(ctx, dev, dev, True, L.PRECISIONS[prec]).to_numpy().astype(np.float64)
        err = np.abs(r - truth)
        bar = 2e-6 + 1e-5 * np.abs(truth)
        print("  %-7s diag %.2e  off %.2e  worst err/bar %.2f" % (prec, err[~off].max(), err[off].max(), (err / bar).max()))
ctx.close()
