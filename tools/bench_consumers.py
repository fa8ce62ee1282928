"""Throughput of the r-matrix consumers (HBM-bound streaming passes) at 20 000 x 20 000, and the
striped Pearson -> edge list pipeline (r never materialised) on a config-2-shaped operand."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib as L, consumers  # noqa: E402

ctx = L.default_context()
n = 20000
rng = np.random.default_rng(0)
host = np.clip(rng.normal(0, 0.1, (n, n)), -1, 1).astype(np.float32)
r = ctx.from_numpy(host)
bg = rng.normal(0, 0.1, 1_000_000).astype(np.float32)
ctx.prof_enable(True)
for _ in range(3):
    flat = consumers.triu_values(r)
    p = consumers.empirical_pvalues(r, bg)
    e = consumers.edges(r, 0.3, upper_only=True)
    consumers.threshold_zero_diag(r, 0.05)
ctx.sync()
gb = n * n * 4 / 1e9
for name, traffic in (("triu_flatten", gb * 1.0), ("empirical_pvalues", gb * 2), ("threshold_zero_diag", gb * 2),
                      ("edges_count", gb), ("edges_fill", gb)):
    ms, cnt = ctx.prof_query(name)
    print("%-22s %.3f ms/launch  %.0f GB/s (algorithmic %.2f GB)" % (name, ms / cnt, traffic / (ms / cnt * 1e-3), traffic))
print("edges kept at cutoff 0.3: %d of %d upper cells" % (len(e[0]), n * (n - 1) // 2))
del r, p, flat

# ---- striped self-Pearson -> edges: N x N never exists -------------------------------------
N, K = int(os.environ.get("EDGE_ROWS", "100000")), 4096
x = ctx.empty(N, K)
chunk = 10000
for s in range(0, N, chunk):  # clustered synthetic profiles, generated chunk-wise to bound host memory
    rows = min(chunk, N - s)
    base = rng.binomial(40, 0.05, size=(50, K)).astype(np.float32)
    x.upload((base[rng.integers(0, 50, rows)] + rng.binomial(6, 0.3, size=(rows, K))).astype(np.float32), s)
z, _ = L.operand_fill(ctx, x)
ctx.prof_reset()
ctx.sync()
t0 = time.time()
i, j, v = consumers.pearson_edges(z, 0.5, stripe_rows=8192, upper_only=True)
ctx.sync()
dt = time.time() - t0
pairs = N * (N - 1) / 2
print("pearson_edges: N=%d, stripe 8192: %.2f s -> %.1f G unordered pairs/s, %d edges (r >= 0.5); peak r buffer %.1f GB "
      "instead of %.0f GB" % (N, dt, pairs / dt / 1e9, len(i), 8192 * N * 4 / 1e9, N * N * 4 / 1e9))
for name in ctx.prof_names():
    ms, cnt = ctx.prof_query(name)
    print("   %-24s %8.2f ms in %d launches" % (name, ms, cnt))
