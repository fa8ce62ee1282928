"""pearson(x, x) for a result LARGER THAN THE HBM (`np.inner`, seekr/pearson.py:41, is limited by host RAM only):
N = 300 000 rows of 4 096 columns -> r = 360 GB of float32 on a 288 GB GPU, produced in row stripes by the same loop
`pearson()` switches to by itself (seekr_amd/multi.py: pearson_job), every stripe fetched over PCIe while the next one is
contracted.  The host of a test box cannot hold 360 GB either, so the sink here keeps nothing: each stripe lands in one
reusable host buffer, two of its rows are checked against the oracle (numpy float32: row standardisation + np.inner, the
reference's own arithmetic) and the LAST stripe — the one that only exists if every earlier one went by — is checked whole
against float64 on a sample of columns.

    python tools/big_pearson.py [rows=300000] [cols=4096]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import seekr_oracle as orc  # noqa: E402
from seekr_amd import _lib, multi  # noqa: E402
from seekr_amd.distributed import shard_bounds  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = _lib.default_context()
free, total = ctx.mem_info()
print("GPU memory: %.1f GB free of %.1f; r would be %.1f GB" % (free / 1e9, total / 1e9, n * n * 4 / 1e9), flush=True)
rng = np.random.default_rng(0)
t0 = time.time()
x = np.empty((n, cols), np.float32)
for r0 in range(0, n, 20000):  # count-like rows: Poisson levels jittered like normalised counts
    m = min(20000, n - r0)
    x[r0:r0 + m] = rng.poisson(0.5, (m, cols)).astype(np.float32) * np.float32(0.37) + rng.standard_normal((m, cols)).astype(np.float32) * np.float32(0.01)
print("input %d x %d generated in %.1f s" % (n, cols, time.time() - t0), flush=True)


class CheckSink:
    def __init__(self):
        self.buf, self.rows, self.stripes, self.worst, self.bytes = None, 0, 0, 0.0, 0
        self.t_put = 0.0
        self.last = None

    def put(self, dev, nrows, row0, mark):
        t = time.time()
        if self.buf is None or self.buf.shape[0] < nrows:
            self.buf = np.empty((nrows, n), np.float32)
        out = self.buf[:nrows]
        dev.to_numpy_at(mark, out)
        self.t_put += time.time() - t
        self.rows += nrows
        self.stripes += 1
        self.bytes += out.nbytes
        for i in (0, nrows - 1):  # first and last row of the stripe against the reference's float32 arithmetic
            cols_pick = rng.integers(0, n, 4096)
            want = orc.pearson(x[row0 + i:row0 + i + 1], x[cols_pick])[0].astype(np.float64)
            got = out[i, cols_pick].astype(np.float64)
            self.worst = max(self.worst, float(np.max(np.abs(got - want) / (2e-6 + 1e-5 * np.abs(want)))))
        assert abs(out[0, row0] - 1.0) < 2e-6 and abs(out[nrows - 1, row0 + nrows - 1] - 1.0) < 2e-6  # the diagonal, where it belongs
        self.last = (row0, nrows)


sink = CheckSink()
spec = multi.PearsonSpec(x, None, shard_bounds(n, 1), shard_bounds(n, 1), False, False, _lib.PREC_F16X3, True, sink,
                         multi.forced_stripe_rows(), np.float32)
ctx.prof_reset()
ctx.prof_enable(True)
t0 = time.time()
multi.pearson_job(multi._Solo(ctx), spec)
wall = time.time() - t0
ctx.prof_enable(False)
gemm_ms = sum(ctx.prof_query(k)[0] for k in ctx.prof_names() if k.startswith("pearson_gemm"))
row0, nrows = sink.last
tail_cols = rng.integers(0, n, 2048)
truth = orc.pearson_f64_truth(x[row0:row0 + nrows], x[tail_cols])
err = np.abs(sink.buf[:nrows][:, tail_cols] - truth) / (2e-6 + 1e-5 * np.abs(truth))
assert sink.rows == n and row0 + nrows == n
print("r %d x %d = %.1f GB in %d stripes: %.1f s wall (%.1f s of it fetching stripes: %.1f GB/s over PCIe), contraction %.2f s "
      "on the device = %.1f G pairs/s" % (n, n, sink.bytes / 1e9, sink.stripes, wall, sink.t_put, sink.bytes / 1e9 / sink.t_put,
                                          gemm_ms / 1e3, n * n / gemm_ms / 1e6))
print("two rows of every stripe against the reference's float32 result: worst %.3f of the bar; the last stripe (%d rows) "
      "x 2 048 columns against float64: worst %.3f of the bar" % (sink.worst, nrows, float(err.max())))
assert sink.worst <= 1.0 and err.max() <= 0.6
print("ok")
