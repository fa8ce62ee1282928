"""Throughput of the r-matrix consumers (HBM-bound streaming passes) at 20 000 x 20 000."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from seekr_amd import _lib as L, consumers

ctx = L.default_context()
n = 20000
rng = np.random.default_rng(0)
r = ctx.from_numpy(np.clip(rng.normal(0, 0.1, (n, n)), -1, 1).astype(np.float32))
bg = rng.normal(0, 0.1, 1_000_000).astype(np.float32)
ctx.prof_enable(True)
for _ in range(3):
    flat = consumers.triu_values(r)
    p = consumers.empirical_pvalues(r, bg)
    consumers.threshold_zero_diag(r, 0.05)
ctx.sync()
gb = n * n * 4 / 1e9
for name, traffic in (("triu_flatten", gb * 1.0), ("empirical_pvalues", gb * 2), ("threshold_zero_diag", gb * 2)):
    ms, cnt = ctx.prof_query(name)
    print("%-22s %.3f ms/launch  %.0f GB/s (algorithmic %.2f GB)" % (name, ms / cnt, traffic / (ms / cnt * 1e-3), traffic))
