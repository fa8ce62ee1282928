"""Throughput of the float64 Pearson contraction (the path CSV / integer inputs take)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seekr_amd import _lib
ctx = _lib.default_context()
for n in (8000, 20000):
    a = np.random.default_rng(0).random((n, 4096))
    d = ctx.from_numpy(a)
    r = ctx.empty(n, n, np.float64)
    _lib.pearson(ctx, d, d, True, _lib.PREC_F64, r); ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.time()
    for _ in range(3):
        _lib.pearson(ctx, d, d, True, _lib.PREC_F64, r)
    ctx.sync(); dt = (time.time() - t0) / 3
    ctx.prof_enable(False)
    print("n=%d: %.1f ms per call -> %.1f TFLOP/s (2*K flop per ordered pair)" % (n, dt * 1e3, 2.0 * 4096 * n * n / dt / 1e12))
    for name in ctx.prof_names():
        ms, cnt = ctx.prof_query(name)
        print("    %-28s %8.2f ms / launch" % (name, ms / cnt))
    want = np.corrcoef(a[:500])
    got = r.to_numpy(0, 500)[:, :500]
    print("    max |err| vs numpy float64:", np.abs(got - want).max())
