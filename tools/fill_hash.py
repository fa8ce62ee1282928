"""Fingerprints of what skr_operand_fill produces (normalised counts y, and r of the operand against itself) over shapes,
normalisation modes and precisions, on rows that exercise the special cases: zero-variance and NaN columns, a nearly
one-hot row, a constant row, few-valued rows.  Run before and after a change to the fill kernels that must not change
a bit (NaN payloads are canonicalised):  python tools/fill_hash.py > before.txt ; ... ; diff before.txt after.txt"""
import hashlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from seekr_amd import _lib as L
if "--lib" in sys.argv:  # another build of the library (before / after across source versions)
    L.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
WIDE = "--wide" in sys.argv  # the workgroup-per-row kernel: k = 7 rows with float64 vectors (row parked in the LDS), k = 8 rows
ctx = L.default_context()
rng = np.random.default_rng(5)


def canon(m):
    m = m.to_numpy()
    m[np.isnan(m)] = np.float32(np.nan)
    return m.tobytes()


for cols, rows in (((16384, 301), (65536, 203), (65544, 57)) if WIDE else ((1024, 3001), (4096, 2503), (16384, 1201), (256, 1500), (729, 900))):
    x = (rng.binomial(1995, 1.0 / 4096, size=(rows, cols)) * np.float32(1000.0 / 1995)).astype(np.float32)
    x[5] = 0; x[5, 7] = 3.0                      # nearly one-hot
    x[9, :] = np.float32(0.5)                    # constant row -> NaN after row standardisation
    x[11, ::3] = 7.25
    x[20:20 + rows // 3] = rng.choice([0.0, 0.5, 1.0, 4.0], (rows // 3, cols), p=[0.7, 0.2, 0.09, 0.01]).astype(np.float32)
    mean = x.mean(0).astype(np.float32); std = x.std(0).astype(np.float32); std[3] = 0.0; mean[4] = np.nan
    if WIDE:  # float64 vectors keep k = 7 rows off the register kernel
        mean, std = mean.astype(np.float64), std.astype(np.float64)
    dx = ctx.from_numpy(x)
    for mode in ("plain", "zscore", "post"):
        for prec in (("f16x3", "fp32") if WIDE else ("f16x3", "bf16x3", "fp32")):
            y = ctx.empty(rows, cols) if mode != "plain" else None
            kw = {}
            if mode != "plain":
                kw = dict(center=ctx.from_numpy(mean.reshape(1, -1)), scale=ctx.from_numpy(std.reshape(1, -1)), y=y)
            if mode == "post":
                kw.update(post=True, shift=7.0)
            op, nan = L.operand_fill(ctx, dx, precision=L.PRECISIONS[prec], want_nan=True, **kw)
            r = ctx.empty(rows, rows)
            L.pearson_gemm_op(ctx, op, op, r, symmetric=True)
            h = hashlib.sha256(canon(r))
            if y is not None:
                h.update(canon(y))
            print(cols, mode, prec, op.kind, nan, h.hexdigest()[:16])
