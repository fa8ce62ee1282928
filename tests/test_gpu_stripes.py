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
         ("f16x3", 4096, 900, 256, True), ("bf16x3", 4096, 700, 333, False), ("bf16x4", 4096, 600, 256, False),
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
