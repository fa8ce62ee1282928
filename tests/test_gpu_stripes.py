"""Row stripes of a self-comparison carry the bits of the one-block call (skr_pearson_gemm_op_rows, symmetric = 2 of
skr_pearson_gemm_op, skr_pearson_gemm_f64): what lets `pearson()` produce a result larger than the HBM stripe by stripe,
and several GPUs produce it row block by row block, without the answer depending on how it was cut
(reference: one np.inner, seekr/pearson.py:41).  Needs a real MI355X: run with `-m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b)) or bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def rows_like_counts(n, cols, seed, structured=False):
    rng = np.random.default_rng(seed)
    x = rng.poisson(1.3, size=(n, cols)).astype(np.float32) * np.float32(0.5013)
    x += rng.standard_normal((n, cols)).astype(np.float32) * np.float32(0.05)
    if structured:  # rows that are mostly one repeated value: the contraction restarts its accumulators more often
        x[:] = np.float32(0.25)
        x[np.arange(n), rng.integers(0, cols, n)] = np.float32(3.0)
        x[:, :7] += rng.standard_normal((n, 7)).astype(np.float32)
    return x


@pytest.fixture(scope="module")
def L():
    from seekr_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.default_context()


CASES = [("f16x3", 4096, 1333, 500, False), ("f16x3", 16384, 700, 300, False), ("f16x3", 256, 900, 257, False),
         ("f16x3", 4096, 900, 256, True), ("bf16x3", 4096, 700, 333, False),
         ("f16f8", 4096, 800, 300, False), ("fp32", 4096, 500, 200, False), ("fp32", 100, 300, 77, False),
         ("f16x3", 1000, 520, 130, False)]


@pytest.mark.parametrize("precision,cols,n,stripe,structured", CASES)
def test_stripes_of_a_self_comparison_equal_the_one_block_call(L, ctx, precision, cols, n, stripe, structured):
    x = ctx.from_numpy(rows_like_counts(n, cols, 5, structured))
    op, _ = L.operand_fill(ctx, x, precision=L.PRECISIONS[precision])
    whole = ctx.zeros(n, n)
    L.pearson_gemm_op(ctx, op, op, whole, symmetric=True)
    want = whole.to_numpy()
    assert np.array_equal(bits(want), bits(want.T.copy()))
    # (a) the C entry point, stripes that are views of the operand
    got = np.empty_like(want)
    buf = ctx.zeros(stripe, n)
    for s0 in range(0, n, stripe):
        m = min(stripe, n - s0)
        L.pearson_gemm_op_rows(ctx, op.view(s0, m), op, s0, buf)
        got[s0:s0 + m] = buf.to_numpy(0, m)
    assert same_bits(got, want), "rows: %d cells differ" % int((bits(got) != bits(want)).sum())
    # (b) the pieces by hand, the stripe a COPY of its rows (what a rank's own shard is to the all-gathered operand)
    s0, m = stripe, min(stripe, n - stripe)
    shard_x = ctx.from_numpy(x.to_numpy(s0, m))
    shard, _ = L.operand_fill(ctx, shard_x, precision=L.PRECISIONS[precision])
    if shard.kind != op.kind:
        pytest.skip("the shard alone chose another layout (kind %d vs %d)" % (shard.kind, op.kind))
    shard.coherent = op.coherent
    if precision == "f16f8":
        shard.x8_stats = op.x8_stats
    L.pearson_gemm_op_rows(ctx, shard, op, s0, buf)
    assert same_bits(buf.to_numpy(0, m), want[s0:s0 + m])
    # (c) without the swapped form the block left of the diagonal is NOT the mirror's bits for the split precisions
    if precision in ("f16x3", "bf16x3") and not structured and cols >= 1024:
        plain = ctx.zeros(m, s0)
        L.pearson_gemm_op(ctx, op.view(s0, m), op.view(0, s0), plain)
        assert not same_bits(plain.to_numpy(), want[s0:s0 + m, :s0]), "the swapped form would be unnecessary"


@pytest.mark.parametrize("cols,n,stripe", [(4096, 700, 300), (1000, 400, 130), (64, 300, 70)])
def test_float64_stripes(L, ctx, cols, n, stripe):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, cols))
    x[5] = 1.0  # a constant row: NaN row and column
    dx = ctx.from_numpy(x)
    want = L.pearson(ctx, dx, dx, precision=L.PREC_F64).to_numpy()
    kp = (cols + 15) // 16 * 16
    z = ctx.zeros(n, kp, np.float64)
    L.row_standardize(ctx, dx, z)
    got = np.empty_like(want)
    buf = ctx.zeros(stripe, n, np.float64)
    for s0 in range(0, n, stripe):
        m = min(stripe, n - s0)
        a = z.view(s0, m)
        if s0:
            L.pearson_gemm_f64(ctx, a, z.view(0, s0), buf, cols)
        L.pearson_gemm_f64(ctx, a, a, buf, cols, symmetric=True, col0=s0)
        if s0 + m < n:
            L.pearson_gemm_f64(ctx, a, z.view(s0 + m, n - s0 - m), buf, cols, col0=s0 + m)
        got[s0:s0 + m] = buf.to_numpy(0, m)
    assert same_bits(got, want)


@pytest.mark.parametrize("cols", [15625, 8197, 9999, 16383, 12346, 16807, 19683, 8200, 10000, 10648, 38416, 40004, 46656])
def test_rows_of_odd_width_through_the_register_row_kernel(L, ctx, cols):
    """Round 5: widths of 8 193 .. 16 384 columns that are not a multiple of 8 (5^6 = 15 625, 7^5 = 16 807 is beyond) take
    operand_fill_rowreg_kernel — the row in the registers of a sixteen-wave workgroup, rows on 4-byte boundaries read in
    16-byte pieces.  The normalised counts it writes back are the elementwise reference values bit for bit, r is inside
    the bar of the reference (strict) and of float64, the flags behave (a mostly-constant set is 'coherent'), and the
    pipeline form (float32 mean / std, Log2.post) agrees with the separate elementwise pass (skr_apply).  The widths
    above 16 384 (7^5, 3^9: the block kernel with the row staged through the LDS in 16-byte pieces; 6^6 and 40 004: rows too
    wide for the LDS, sixteen waves a row) and the multiples of 8 that are not multiples of 32 (8 200, 10^4, 22^3, 14^4: the
    block kernel's vector path, whose padding groups read past the row until round 5 — r was 5 bars off) run the same
    checks on the neighbouring kernels."""
    from oracle import seekr_oracle as orc
    rng = np.random.default_rng(cols)
    n = 70 if cols < 20000 else 40
    x = (rng.poisson(0.4, size=(n, cols)) * np.float32(0.5013)).astype(np.float32)
    x[3] = x[2] * np.float32(2.0) + np.float32(0.25)       # r = 1 off the diagonal
    x[5, 7:] = x[4, :-7]                                  # a shifted near-copy
    dev = ctx.from_numpy(x)
    mean = ctx.from_numpy(x.mean(axis=0).astype(np.float32))
    std = ctx.from_numpy((x.std(axis=0) + np.float32(0.05)).astype(np.float32))
    # bare form: rows as they are
    op, _ = L.operand_fill(ctx, dev, precision=L.PREC_F16X3)
    assert op.kind == 2
    r = ctx.zeros(n, n)
    L.pearson_gemm_op(ctx, op, op, r, symmetric=True)
    got = r.to_numpy().astype(np.float64)
    with np.errstate(all="ignore"):
        ref, truth = orc.pearson(x, x).astype(np.float64), orc.pearson_f64_truth(x, x)
    assert (np.abs(got - ref) <= 2e-6 + 1e-5 * np.abs(ref)).all(), float((np.abs(got - ref) / (2e-6 + 1e-5 * np.abs(ref))).max())
    assert (np.abs(got - truth) <= 0.6 * (2e-6 + 1e-5 * np.abs(truth))).all()
    # pipeline form: centre, scale, Log2.post, counts kept — against the separate elementwise kernel, bit for bit
    y = ctx.empty(n, cols)
    shift = 3.0
    op2, has_nan = L.operand_fill(ctx, dev, precision=L.PREC_F16X3, center=mean, scale=std, post=True, shift=shift, y=y, want_nan=True)
    want_y, _ = L.apply(ctx, dev, y=ctx.empty(n, cols), center=mean, scale=std, post=True, shift=shift)
    assert not has_nan and np.array_equal(bits(y.to_numpy()), bits(want_y.to_numpy()))
    r2 = ctx.zeros(n, n)
    L.pearson_gemm_op(ctx, op2, op2, r2, symmetric=True)
    yy = y.to_numpy()
    with np.errstate(all="ignore"):
        ref2 = orc.pearson(yy, yy).astype(np.float64)
    assert (np.abs(r2.to_numpy() - ref2) <= 2e-6 + 1e-5 * np.abs(ref2)).all()
    # fp32 layout (the float instantiation) and a set that must raise the coherent flag
    opf, _ = L.operand_fill(ctx, dev, precision=L.PREC_FP32)
    rf = ctx.zeros(n, n)
    L.pearson_gemm_op(ctx, opf, opf, rf, symmetric=True)
    assert (np.abs(rf.to_numpy() - ref) <= 2e-6 + 1e-5 * np.abs(ref)).all()
    flat = np.full((8, cols), np.float32(0.25))
    flat[np.arange(8), rng.integers(0, cols, 8)] = 3.0
    flat[:, :5] += rng.standard_normal((8, 5)).astype(np.float32)
    opc, _ = L.operand_fill(ctx, ctx.from_numpy(flat), precision=L.PREC_F16X3)
    assert opc.coherent and not op.coherent
