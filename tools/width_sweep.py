"""Every kernel behind normalisation and Pearson, at MANY widths (round 5, after the block fill was found reading a row's
tile padding at 8 200 columns — a width no test, golden set or fuzzer had).  The kernels choose their code path by the
width of a row: multiples of 4 / 8 / 16 / 32, rows that fit a wave / the registers of a workgroup / the LDS / none of
them, whole strips of 16 columns or a ragged last one.  This walks the widths instead of the paths:

  every width 1 .. 130, the neighbours (-9 .. +9) of every power of two and of the kernels' own thresholds up to 70 000,
  multiples of 8 that are not multiples of 32 across the whole range, and seeded random widths;

and at each one checks, against the oracle (numpy float32 restatement of seekr/kmer_counts.py:201-209 and
seekr/pearson.py:35-41):

  normalize   column mean, column std and the normalised matrix of `skr_normalize` (Log2.none) BIT-EXACT; Log2.post within
              1e-5 relative + 2e-6 absolute;
  fused       `skr_operand_fill` with centre / scale / Log2.post writing the normalised counts back == `skr_apply`, bit
              for bit, and the column of the row after the last one untouched (guard rows around the matrix);
  pearson     r of the default precision and of fp32 by tests/parity_rule.py (strict against the reference, float64 as the
              yardstick where the reference itself is a range), self and cross comparison; float64 rows (1e-11 from
              numpy's float64) and `row_standardize=False`.

    python tools/width_sweep.py [--quick] [--widths 8200,10000] [--seed 1]

Exit code 1 and the failing widths on stderr if any check fails.  Needs a real MI355X.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

THRESHOLDS = (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 12288, 16384, 32768, 38400, 49152, 65536)


def widths(quick, seed):
    w = set(range(1, 131))
    for t in THRESHOLDS:
        for d in range(-9, 10):
            w.add(t + d)
    rng = np.random.default_rng(seed)
    w.update(int(v) * 8 for v in rng.integers(17, 8750, 40 if quick else 160))          # multiples of 8
    w.update(int(v) * 8 + 4 for v in rng.integers(17, 8750, 10 if quick else 40))       # multiples of 4 only
    w.update(int(v) for v in rng.integers(131, 70000, 30 if quick else 150))             # anything
    named = {8200, 10000, 10648, 15625, 16807, 19683, 20736, 38416, 40004, 46656, 50625, 59049, 69999}  # alphabet^k widths
    if not quick:
        named |= {78125, 100000, 117649, 160000, 262144}  # 5^7, 10^5, 7^6, 20^4, 4^9: beyond the split-fp16 kernel's widths
    if quick:  # thin the neighbourhoods: every third neighbour of the thresholds above 256
        w = {v for v in w if v <= 300 or v % 3 != 1 or v % 8 == 0}
    return sorted(v for v in w | named if v >= 1)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rows_for(rng, n, cols):
    x = (rng.poisson(0.7, size=(n, cols)) * np.float32(0.5013)).astype(np.float32)
    x += (rng.integers(0, 3, size=(n, 1)) * np.float32(0.125)).astype(np.float32)
    if cols >= 2:
        x[np.arange(n), rng.integers(0, cols, n)] += np.float32(2.5)  # no constant rows
        x[0::2, 0] += np.float32(1.0)  # no zero-variance first column
    if n >= 4 and cols >= 8:
        x[3] = x[2] * np.float32(2.0) + np.float32(0.25)  # r = 1 off the diagonal
    return x


def check_width(L, ctx, orc, parity_rule, pearson, cols, seed):
    problems = []
    rng = np.random.default_rng([seed, cols])
    n = 33 if cols > 20000 else 61
    x = rows_for(rng, n, cols)
    # normalize --------------------------------------------------------------------------------------------------------
    with np.errstate(all="ignore"):
        z_ref, mean_ref, std_ref = orc.normalize(x, log2="Log2.none")
        post_ref = orc.normalize(x, log2="Log2.post")[0]
    clean = not np.isnan(z_ref).any()
    dev = ctx.from_numpy(x)
    mean_out, std_out, has_nan = L.normalize(ctx, dev, "Log2.none", 1, None, 1, None)
    if has_nan == clean:
        problems.append("normalize: has_nan %s, oracle clean %s" % (has_nan, clean))
    if not np.array_equal(bits(mean_out.vector()), bits(mean_ref)):
        problems.append("column mean differs")
    if not np.array_equal(bits(std_out.vector()), bits(std_ref)):
        problems.append("column std differs")
    got = dev.to_numpy()
    if not (np.array_equal(bits(got), bits(z_ref)) or np.all((bits(got) == bits(z_ref)) | (np.isnan(got) & np.isnan(z_ref)))):
        problems.append("normalised matrix differs in %d cells" % int((bits(got) != bits(z_ref)).sum()))
    dev = ctx.from_numpy(x)
    L.normalize(ctx, dev, "Log2.post", 1, None, 1, None)
    got = dev.to_numpy()
    if clean and not np.allclose(got, post_ref, rtol=1e-5, atol=2e-6):
        problems.append("Log2.post beyond the bar: %.3g" % float(np.nanmax(np.abs(got - post_ref))))
    if not clean:
        return problems  # a zero-variance column: everything after is NaN in the reference too
    # fused fill -------------------------------------------------------------------------------------------------------
    dmean, dstd = ctx.from_numpy(mean_ref), ctx.from_numpy(std_ref)
    shift = float(np.abs(L.min_nan(ctx, ctx.from_numpy(x), dmean, dstd)[0]))
    want_y, _ = L.apply(ctx, ctx.from_numpy(x), center=dmean, scale=dstd, post=True, shift=shift)
    want_y = want_y.to_numpy()
    guard = np.float32(-77.0)
    framed = np.full((n + 2, cols), guard, dtype=np.float32)  # a row in front and one behind the matrix that nobody may touch
    framed[1:-1] = x
    for prec in ("f16x3", "fp32"):
        whole = ctx.from_numpy(framed)
        inner = whole.view(1, n)
        op, nan2 = L.operand_fill(ctx, inner, precision=L.PRECISIONS[prec], center=dmean, scale=dstd, post=True, shift=shift,
                                  y=inner, want_nan=True)
        back = whole.to_numpy()
        if nan2:
            problems.append("%s fused: NaN reported" % prec)
        if not np.array_equal(bits(back[1:-1]), bits(want_y)):
            problems.append("%s fused counts differ from skr_apply in %d cells" % (prec, int((bits(back[1:-1]) != bits(want_y)).sum())))
        if not ((back[0] == guard).all() and (back[-1] == guard).all()):
            problems.append("%s fused fill wrote outside its rows" % prec)
        r = ctx.zeros(n, n)
        L.pearson_gemm_op(ctx, op, op, r, symmetric=True)
        with np.errstate(all="ignore"):
            ref, truth = orc.pearson(want_y, want_y).astype(np.float64), orc.pearson_f64_truth(want_y, want_y)
        ok = np.isfinite(ref) & np.isfinite(truth)
        v = parity_rule.judge(r.to_numpy(), ref, truth, ok, want_y, want_y)
        if v["failures"]:
            problems.append("%s fused r: %s" % (prec, v["failures"][0]))
    # pearson() on the raw rows, self and cross ----------------------------------------------------------------------------
    y = rows_for(rng, 19, cols)
    for prec in ("f16x3", "fp32"):
        os.environ["SEEKR_PRECISION"] = prec
        try:
            for name, a, b in (("self", x, x), ("cross", x, y)):
                with np.errstate(all="ignore"):
                    ref, truth = orc.pearson(a, b).astype(np.float64), orc.pearson_f64_truth(a, b)
                    got = pearson(a, b).astype(np.float64)
                ok = np.isfinite(ref) & np.isfinite(truth)
                if not np.array_equal(np.isnan(got), np.isnan(ref)):
                    problems.append("%s %s: NaN pattern differs" % (prec, name))
                v = parity_rule.judge(got, ref, truth, ok, a, b)
                if v["failures"]:
                    problems.append("%s %s r: %s" % (prec, name, v["failures"][0]))
        finally:
            os.environ.pop("SEEKR_PRECISION", None)
    # the other kernels behind the same call: float64 rows (the float64 contraction) and rows taken as they are
    a64, b64 = x[:23].astype(np.float64), y.astype(np.float64)
    with np.errstate(all="ignore"):
        for name, a, b in (("self", a64, a64), ("cross", a64, b64)):
            got, truth = pearson(a, b), orc.pearson_f64_truth(a, b)
            if got.dtype != np.float64 or not np.allclose(got, truth, rtol=1e-11, atol=1e-12, equal_nan=True):
                problems.append("float64 %s: %.3g from numpy's float64" % (name, float(np.nanmax(np.abs(got - truth)))))
        a, b = x[:23], y
        got = pearson(a, b, row_standardize=False).astype(np.float64)
        ref, truth = orc.pearson(a, b, False).astype(np.float64), orc.pearson_f64_truth(a, b, False)
    unit = float(max(np.abs(truth).max(), 1.0))
    v = parity_rule.judge(got, ref, truth, np.isfinite(ref) & np.isfinite(truth), a, b, row_standardize=False, unit=unit)
    if v["failures"]:
        problems.append("rows as they are: %s" % (v["failures"][0],))
    return problems


def sweep(ws, seed=1, verbose=True):
    from oracle import seekr_oracle as orc
    from seekr_amd import _lib as L
    from seekr_amd.pearson import pearson
    import parity_rule
    ctx = L.default_context()
    bad = {}
    for i, cols in enumerate(ws):
        p = check_width(L, ctx, orc, parity_rule, pearson, cols, seed)
        if p:
            bad[cols] = p
            print("width %6d  FAIL  %s" % (cols, "; ".join(p)), file=sys.stderr, flush=True)
        elif verbose and i % 50 == 0:
            print("width %6d  ok  (%d of %d)" % (cols, i + 1, len(ws)), flush=True)
    return bad


ROW_COUNTS = (1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 384, 385, 511, 512, 513,
              767, 768, 769, 1023, 1024, 1025, 1279, 1281)


def rows_sweep(seed=1, verbose=True):
    """The other axis: row counts around the contraction's tile heights (16 / 32 / 64 / 128 / 256 rows) for the self
    comparison (the symmetric kernel: blocks on the diagonal, mirrored ones), a cross comparison with different row counts
    on the two sides, and row stripes of the self comparison (bit-identical to the one-block call), at four widths."""
    from oracle import seekr_oracle as orc
    from seekr_amd import _lib as L
    from seekr_amd.pearson import pearson
    import parity_rule
    ctx = L.default_context()
    bad = {}
    for cols in (24, 1000, 4096, 8200):
        for n in ROW_COUNTS:
            rng = np.random.default_rng([seed, cols, n])
            x = rows_for(rng, n, cols)
            m = ROW_COUNTS[(ROW_COUNTS.index(n) * 7 + 3) % len(ROW_COUNTS)]
            y = rows_for(rng, m, cols)
            problems = []
            for name, a, b in (("self", x, x), ("cross", x, y)):
                with np.errstate(all="ignore"):
                    ref, truth = orc.pearson(a, b).astype(np.float64), orc.pearson_f64_truth(a, b)
                    got = pearson(a, b)
                if not np.array_equal(np.isnan(got), np.isnan(ref)):
                    problems.append("%s: NaN pattern differs" % name)
                ok = np.isfinite(ref) & np.isfinite(truth)
                v = parity_rule.judge(got.astype(np.float64), ref, truth, ok, a, b)
                if v["failures"]:
                    problems.append("%s r: %s" % (name, v["failures"][0]))
                if name == "self":
                    if not np.array_equal(bits(got), bits(got.T.copy())):
                        problems.append("self comparison not symmetric")
                    if n >= 3:  # three ragged stripes carry the one-block bits
                        op, _ = L.operand_fill(ctx, ctx.from_numpy(x))
                        whole = ctx.zeros(n, n)
                        L.pearson_gemm_op(ctx, op, op, whole, symmetric=True)
                        want = whole.to_numpy()
                        cut = sorted({0, n // 3, n // 3 + (n + 1) // 2, n})
                        for s0, s1 in zip(cut[:-1], cut[1:]):
                            buf = ctx.zeros(s1 - s0, n)
                            L.pearson_gemm_op_rows(ctx, op.view(s0, s1 - s0), op, s0, buf)
                            if not np.array_equal(bits(buf.to_numpy()), bits(want[s0:s1])):
                                problems.append("stripe [%d, %d) differs from the one-block call" % (s0, s1))
            if problems:
                bad[(cols, n)] = problems
                print("cols %5d rows %5d x %5d  FAIL  %s" % (cols, n, m, "; ".join(problems)), file=sys.stderr, flush=True)
        if verbose:
            print("cols %5d: %d row counts" % (cols, len(ROW_COUNTS)), flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--widths", default=None)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    ws = [int(w) for w in args.widths.split(",")] if args.widths else widths(args.quick, args.seed)
    bad = sweep(ws, args.seed)
    print("%d widths (%d .. %d), %d failing%s" % (len(ws), ws[0], ws[-1], len(bad), ": " + ",".join(map(str, sorted(bad))) if bad else ""))
    bad_rows = {} if args.widths else rows_sweep(args.seed)
    print("%d row counts x 4 widths, %d failing" % (len(ROW_COUNTS), len(bad_rows)))
    sys.exit(1 if bad or bad_rows else 0)


if __name__ == "__main__":
    main()
