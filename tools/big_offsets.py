"""Matrices beyond 4 GiB (config 5's per-GPU shard is 8 GB of counts and 8 GB of operand): a k = 7
count matrix of 72 000 rows (4.7 GB) goes through counting, the chained column statistics, the fused
normalise + standardise + split pass and the contraction of its LAST rows against its FIRST rows, each
checked against the oracle — byte offsets past 2^32 in every kernel of the path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import seekr_oracle as orc  # noqa: E402  (checker only)
from seekr_amd import _lib  # noqa: E402
from seekr_amd.distributed import HipEngine, SingleComm, sharded_normalize_prepare  # noqa: E402
from seekr_amd.synthetic import synthetic_ascii  # noqa: E402

N, L, K_MER = int(os.environ.get("BIG_ROWS", "72000")), 600, 7
ctx = _lib.default_context()
blob, off = synthetic_ascii(9, N, L)
packed = _lib.PackedSeqs.from_buffer(ctx, blob, off, "AGTC")
cols = 4 ** K_MER
x = ctx.empty(N, cols)
print("count matrix %.2f GB" % (N * cols * 4 / 1e9), flush=True)
t0 = time.time()
_lib.count_per_kb(ctx, packed, K_MER, out=x)
ctx.sync()
print("count %.3f s" % (time.time() - t0), flush=True)
tail = 150
seqs_tail = [bytes(blob[off[i]:off[i + 1]]).decode() for i in range(N - tail, N)]
want = orc.raw_counts(seqs_tail, K_MER)
got = x.to_numpy(N - tail, tail)
assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "raw counts of the last rows differ"
raw = x.to_numpy()
engine = HipEngine(ctx, _lib.PREC_F16X3)
t0 = time.time()
mean, std, has_nan, z = sharded_normalize_prepare(engine, SingleComm(), x, N, "Log2.post", True, True, keep_counts=True)
ctx.sync()
print("statistics + fused fill %.3f s (nan=%s)" % (time.time() - t0, has_nan), flush=True)
want_x, want_mean, want_std = orc.normalize(raw, True, True, "Log2.post")
del raw
assert np.array_equal(mean.to_numpy().reshape(-1).view(np.uint32), want_mean.view(np.uint32)), "mean vector differs"
assert np.array_equal(std.to_numpy().reshape(-1).view(np.uint32), want_std.view(np.uint32)), "std vector differs"
got_x = x.to_numpy()
if np.isnan(want_x).any():
    assert np.array_equal(np.isnan(got_x), np.isnan(want_x))
else:
    np.testing.assert_allclose(got_x[::97], want_x[::97], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(got_x[-500:], want_x[-500:], rtol=1e-5, atol=2e-6)
blk = 2500
r = ctx.empty(blk, blk)
_lib.pearson_gemm_op(ctx, z.view(N - blk, blk), z.view(0, blk), r)
truth = orc.pearson_f64_truth(got_x[N - blk:], got_x[:blk])
err = np.abs(r.to_numpy() - truth)
print("Pearson of rows [%d, %d) x [0, %d): max |err| %.2e" % (N - blk, N, blk, np.nanmax(err)))
assert np.nanmax(err) < 2e-6 + 1e-5
print("big offsets ok")
