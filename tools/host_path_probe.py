"""Where the host-to-host time of pearson() goes (12 000 x 4 096 float32 -> 12 000 x 12 000): upload, kernels, download,
with the result pool on and off (SEEKR_RESULT_POOL_MB=0); and the same for get_counts()'s download."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402
from seekr_amd.pearson import pearson  # noqa: E402

ctx = _lib.default_context()
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
x = np.log2(rng.binomial(1995, 1 / 4096, size=(n, 4096)).astype(np.float32) * np.float32(0.5) + 1)


def t(fn, reps=5):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ctx.sync()
        out.append(time.perf_counter() - t0)
        del r
    return "first %.1f ms, then median %.1f ms" % (out[0] * 1e3, float(np.median(out[1:])) * 1e3)


print("pool cap %d MB" % (_lib.host_pool.cap_bytes() >> 20))
print("pearson(x, x) host to host      :", t(lambda: pearson(x, x)))
d = ctx.from_numpy(x)
print("upload %d MB                    :" % (x.nbytes >> 20), t(lambda: ctx.from_numpy(x)))
r = _lib.pearson(ctx, d, d, precision=_lib.PREC_F16X3)
print("kernels (fill + contraction)    :", t(lambda: _lib.pearson(ctx, d, d, precision=_lib.PREC_F16X3)))
print("download %d MB (pooled)        :" % (r.rows * r.cols * 4 >> 20), t(lambda: r.to_numpy()))
keep = np.empty((r.rows, r.cols), np.float32)
keep[:] = 0
print("download into a touched array   :", t(lambda: r.to_numpy(out=keep)))
print("download into fresh np.empty    :", t(lambda: r.to_numpy(out=np.empty((r.rows, r.cols), np.float32))))
print("np.empty + touch (no copy)      :", t(lambda: np.empty((r.rows, r.cols), np.float32).fill(0)))
print(_lib.host_pool.stats)
