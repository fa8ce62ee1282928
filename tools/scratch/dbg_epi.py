import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seekr_amd import _lib as L
ctx = L.default_context()
rng = np.random.default_rng(1)
rows, cols = 768, 1024
xa = (rng.binomial(40, 0.06, size=(rows, cols)) * np.float32(0.5)).astype(np.float32)
xb = (rng.binomial(40, 0.06, size=(rows, cols)) * np.float32(0.5)).astype(np.float32)
za, _ = L.operand_fill(ctx, ctx.from_numpy(xa), precision=L.PREC_F16X3)
zb, _ = L.operand_fill(ctx, ctx.from_numpy(xb), precision=L.PREC_F16X3)
def run(val, sym):
    os.environ["SEEKR_GEMM_EPILOGUE"] = val
    ctx.reload_knobs()
    r = ctx.zeros(rows, rows)
    L.pearson_gemm_op(ctx, za, za if sym else zb, r, symmetric=sym)
    return r.to_numpy()
for sym in (False, True):
    base = run("0", sym)
    for val in ("1", "2", "3"):
        got = run(val, sym)
        bad = np.argwhere(got.view(np.uint32) != base.view(np.uint32))
        print("sym", sym, "arm", val, "mismatches", len(bad))
        if len(bad):
            print(" first", bad[:6].tolist(), "rows%16", sorted(set((bad[:, 0] % 16).tolist()))[:16], "cols%32", sorted(set((bad[:, 1] % 32).tolist())))
            i, j = bad[0]
            # where does the wrong value come from?
            w = np.argwhere(base.view(np.uint32) == got.view(np.uint32)[i, j])
            print(" value at", (i, j), "is the base's value at", w[:4].tolist())
