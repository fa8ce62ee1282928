"""The whole step at a BASELINE size on ONE GPU, with the N x N Pearson matrix checked where it was
never checked before: at the size bench.py times (config 2: 50 000 rows, r = 10 GB) and at config 4's
size (200 000 rows, r = 160 GB, fits the 288 GB of one MI355X).

    python tools/fullsize_check.py --rows 200000 [--length 2000] [-k 6] [--precision f16x3]

count -> column statistics -> fused normalise + standardise + split -> self-Pearson (SELF mode: one
triangle multiplied, the other mirrored), then, against the oracle (pearson.py:35-41 restated) on host
copies of the normalised rows:
  * >= 64 random rows x ALL columns and the last tile row (rows N-256..N-1 x all columns);
  * the same rows as COLUMNS (cells r[j, i] gathered on the device, every j): bit-equal to r[i, j] —
    the mirror stores at byte offsets past 2^32 (past 2^37 at 200 000 rows);
  * the whole diagonal;
  * a 2 048 x 2 048 corner block at the far end against float64.
Prints "fullsize ok rows=N" on success.  The oracle is only the checker (tests/ and tools/ may use it).
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import seekr_oracle as orc  # noqa: E402  (checker only)
from seekr_amd import _lib  # noqa: E402
from seekr_amd.distributed import HipEngine, SingleComm, sharded_normalize_prepare  # noqa: E402
from seekr_amd.synthetic import synthetic_ascii  # noqa: E402

RTOL, ATOL = 1e-5, 2e-6  # the parity bar on r (north_star: 1e-5 relative; SURVEY A.6 for the absolute term)


def gather_cells(ctx, mat, flat_idx):
    idx = np.ascontiguousarray(flat_idx, dtype=np.int64)
    out = np.empty(len(idx), dtype=np.float32)
    _lib.check(_lib.lib().skr_gather_f32(ctx._h, mat._h, idx.ctypes.data_as(C.c_void_p), len(idx),
                                         out.ctypes.data_as(C.c_void_p)))
    return out


def oracle_rows_against_all(x_host, rows, chunk=16384):
    """orc.pearson(x[rows], x) without a second copy of the whole matrix: columns in chunks."""
    out = np.empty((len(rows), x_host.shape[0]), dtype=np.float32)
    a = x_host[rows]
    for c0 in range(0, x_host.shape[0], chunk):
        out[:, c0:c0 + chunk] = orc.pearson(a, x_host[c0:c0 + chunk])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--length", type=int, default=2000)
    ap.add_argument("-k", type=int, default=6)
    ap.add_argument("--seed", type=int, default=0, help="default: 2 at 50 000 rows (config 2), else 4 (config 4)")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16x3", "fp32", "f16f8"])
    ap.add_argument("--sample-rows", type=int, default=64)
    args = ap.parse_args()
    n, length, k = args.rows, args.length, args.k
    seed = args.seed or (2 if n == 50000 else 4)
    cols = 4 ** k
    ctx = _lib.default_context()
    t0 = time.time()
    blob, off = synthetic_ascii(seed, n, length)
    packed = _lib.PackedSeqs.from_buffer(ctx, blob, off, "AGTC")
    del blob
    print("generated + packed %d x %d nt in %.1f s" % (n, length, time.time() - t0), flush=True)
    engine = HipEngine(ctx, _lib.PRECISIONS[args.precision])
    x = ctx.empty(n, cols)
    r = ctx.empty(n, n)
    print("r is %.1f GB" % (4.0 * n * n / 1e9), flush=True)
    ctx.sync()
    t0 = time.time()
    _lib.count_per_kb(ctx, packed, k, out=x)
    mean, std, has_nan, z = sharded_normalize_prepare(engine, SingleComm(), x, n, "Log2.post", True, True, keep_counts=True)
    engine.gemm(z, z, r, 0, symmetric=True)
    ctx.sync()
    t_step = time.time() - t0
    print("step (count + normalise + self-Pearson) %.3f s = %.1f G pairs/s (first call, includes lazy kernel loads)"
          % (t_step, n * float(n) / t_step / 1e9), flush=True)
    assert not has_nan
    x_host = x.to_numpy()

    rng = np.random.default_rng(n)
    tile_edges = np.array([0, 255, 256, n - 257, n - 256, n - 1, (n // 512) * 256, (n // 512) * 256 - 1])
    rows = np.unique(np.concatenate([rng.choice(n, args.sample_rows, replace=False), tile_edges]))
    rows = rows[(rows >= 0) & (rows < n)]
    want = oracle_rows_against_all(x_host, rows)
    worst = 0.0
    for i, row in enumerate(rows):
        got = r.to_numpy(int(row), 1).reshape(-1)
        err = np.abs(got - want[i])
        bar = ATOL + RTOL * np.abs(want[i])
        assert (err <= bar).all(), "row %d: worst error / bar = %.3f at column %d" % (row, (err / bar).max(), (err / bar).argmax())
        worst = max(worst, float((err / bar).max()))
        # the same row as a COLUMN of r: the mirror of every tile it crosses, bit for bit
        col = gather_cells(ctx, r, np.arange(n, dtype=np.int64) * n + int(row))
        assert np.array_equal(col.view(np.uint32), got.view(np.uint32)), "column %d is not the mirror of row %d" % (row, row)
    print("%d sampled rows x %d columns inside the bar (worst error / bar %.3f); the same rows as columns: bit-equal"
          % (len(rows), n, worst), flush=True)

    # last tile row: rows N-256 .. N-1, every column
    last = np.arange(max(0, n - 256), n)
    want = oracle_rows_against_all(x_host, last)
    got = r.to_numpy(int(last[0]), len(last))
    err, bar = np.abs(got - want), ATOL + RTOL * np.abs(want)
    assert (err <= bar).all(), "last tile row: worst error / bar = %.3f" % (err / bar).max()
    # and the last tile COLUMN (cells r[j, N-256 .. N-1] for every j) equals its transpose bit for bit
    for c in (n - 256, n - 129, n - 1):
        col = gather_cells(ctx, r, np.arange(n, dtype=np.int64) * n + c)
        assert np.array_equal(col.view(np.uint32), got[c - int(last[0])].view(np.uint32)), "last tile column %d" % c
    print("last tile row (256 x %d) inside the bar (worst %.3f), last tile column mirrored bit for bit"
          % (n, float((err / bar).max())), flush=True)

    diag = gather_cells(ctx, r, np.arange(n, dtype=np.int64) * (n + 1))
    assert np.abs(diag - 1.0).max() < 2e-6, "diagonal: max |r_ii - 1| = %.2e" % np.abs(diag - 1.0).max()
    far = min(2048, n)
    blk = r.to_numpy(n - far, far)[:, :far]  # rows at the far end x the first columns: below the diagonal = mirrored cells
    truth = orc.pearson_f64_truth(x_host[n - far:], x_host[:far])
    e64 = np.abs(blk - truth)
    assert (e64 <= ATOL + RTOL * np.abs(truth)).all(), "far corner against float64: %.2e" % e64.max()
    print("diagonal |r_ii - 1| <= %.1e; far corner (%d x %d, mirrored cells) max |err| vs float64 %.2e"
          % (np.abs(diag - 1.0).max(), far, far, e64.max()), flush=True)

    # timed repeat of the contraction alone (everything warm)
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(2):
        engine.gemm(z, z, r, 0, symmetric=True)
    ctx.sync()
    ctx.prof_enable(False)
    for name in ctx.prof_names():
        ms, cnt = ctx.prof_query(name)
        if cnt:
            print("kernel %s: %.3f ms per launch -> %.1f G pairs/s" % (name, ms / cnt, n * float(n) / (ms / cnt) / 1e6))
    print("fullsize ok rows=%d" % n)


if __name__ == "__main__":
    main()
