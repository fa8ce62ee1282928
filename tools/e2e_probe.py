"""bench.py's e2e sub-record by itself, every repetition printed (is the median what the pool gives?)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from seekr_amd import _lib  # noqa: E402
from seekr_amd.pearson import pearson  # noqa: E402

ctx = _lib.default_context()
print(bench.end_to_end(ctx, 6, 2000, 50000, "f16x3"))
print(_lib.host_pool.stats)
rng = np.random.default_rng(0)
head = rng.random((12000, 4096), dtype=np.float32)
for i in range(6):
    t0 = time.perf_counter()
    pearson(head, head)
    print("pearson %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), _lib.host_pool.stats)
d = ctx.from_numpy(head)
r = _lib.pearson(ctx, d, d, precision=_lib.PREC_F16X3)
keep = np.zeros((12000, 12000), np.float32)
for name, fn in (("upload", lambda: ctx.from_numpy(head)), ("kernels", lambda: _lib.pearson(ctx, d, d, precision=_lib.PREC_F16X3)),
                 ("download touched", lambda: r.to_numpy(out=keep)), ("download pooled", lambda: r.to_numpy())):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        o = fn()
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
        del o
    print(name, ["%.1f" % v for v in ts])
print("affinity", len(os.sched_getaffinity(0)), "cpus; cpu now", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")


def phases(x):
    t = [time.perf_counter()]
    d1 = ctx.from_numpy(x); ctx.sync(); t.append(time.perf_counter())
    rr = _lib.pearson(ctx, d1, d1, row_standardize=True, precision=_lib.PREC_F16X3); ctx.sync(); t.append(time.perf_counter())
    out = rr.to_numpy(); t.append(time.perf_counter())
    d1.free(); rr.free(); t.append(time.perf_counter())
    del out; t.append(time.perf_counter())
    return ["%.1f" % ((b - a) * 1e3) for a, b in zip(t, t[1:])]


for i in range(4):
    print("alloc+upload | alloc r + kernels | download | device frees | host release:", phases(head))
h5 = np.random.default_rng(1).random((12000, 4096), dtype=np.float32)
for i in range(3):
    print("(a new input array)", phases(h5))

import ctypes as C  # noqa: E402
hip = C.CDLL("libamdhip64.so")
buf = C.c_void_p()
n = 12000 * 12000 * 4
assert hip.hipHostMalloc(C.byref(buf), C.c_size_t(n), 0) == 0
C.memset(buf, 1, n)
ptr = r.device_ptr()
for i in range(3):
    t0 = time.perf_counter()
    assert hip.hipMemcpy(buf, C.c_void_p(ptr), C.c_size_t(n), 2) == 0   # D2H
    print("hipMemcpy D2H into hipHostMalloc memory: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for i in range(3):
    t0 = time.perf_counter()
    assert hip.hipMemcpy(C.c_void_p(ptr), buf, C.c_size_t(n), 1) == 0   # H2D
    print("hipMemcpy H2D from hipHostMalloc memory: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
