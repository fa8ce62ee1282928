"""The operand fill on rows whose width is not a power of four: 50 000 x 15 625 (5^6) and 20 000 x
65 536 (4^8) float32 rows, the pipeline's form (float32 mean / std vectors, Log2.post, normalised counts kept) and the bare
form (rows as they are), timed by the library's own HIP events.  bytes = rows x cols x 4 x (read + operand written [+ y])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402

ctx = _lib.default_context()
shapes = [(50000, 15625), (20000, 65536), (50000, 4096), (30000, 16384), (50000, 2401)]
if len(sys.argv) > 1:
    shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
rng = np.random.default_rng(0)
for rows, cols in shapes:
    block = (rng.poisson(0.13, size=(2000, cols)) * np.float32(0.5013)).astype(np.float32)
    x0 = ctx.empty(rows, cols)
    for r0 in range(0, rows, 2000):
        x0.upload(block[:min(2000, rows - r0)], r0)
    mean = ctx.from_numpy(np.full(cols, 0.065, np.float32))
    std = ctx.from_numpy(np.full(cols, 0.18, np.float32))
    x = ctx.empty(rows, cols)
    op = _lib.Operand(ctx, rows, cols, _lib.PREC_F16X3)
    for name, kw, nbuf in (("bare (no vectors, no y)", dict(), 2.0), ("pipeline (mean, std, Log2.post, y kept)", dict(center=mean, scale=std, post=True, shift=0.4, y=x), 3.0)):
        ts = []
        for _ in range(7):
            ctx.sync()
            ctx.prof_reset()
            ctx.prof_enable(True)
            _lib.operand_fill(ctx, x0, op=op, precision=_lib.PREC_F16X3, **kw)
            ctx.sync()
            ctx.prof_enable(False)
            ts.append(sum(ctx.prof_query(n)[0] for n in ctx.prof_names() if n.startswith("operand_fill")))
        med = float(np.median(ts[2:]))
        gb = rows * cols * 4 * nbuf / 1e9
        print("%6d x %6d  %-42s %.3f ms  %.2f GB -> %.0f GB/s = %.3f of 8 TB/s (kind %d)" % (rows, cols, name, med, gb, gb / med * 1e3, gb / med / 8, op.kind), flush=True)
    for m in (x0, x, op, mean, std):
        m.free()
