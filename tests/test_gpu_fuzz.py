"""Differential fuzzing of BasicCounter / pearson against the oracle (tests/fuzz_differential.py):
fixed seeds, a bounded number of cases per seed."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_cases(seed):
    from fuzz_differential import fuzz
    assert fuzz(seed, budget_s=25.0, max_cases=60) >= 10


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_pearson_api(seed):
    from fuzz_pearson import fuzz
    assert fuzz(seed, budget_s=8.0, max_cases=4000) >= 200


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_fasta_reader(seed):
    from fuzz_fasta import fuzz
    assert fuzz(seed, budget_s=6.0, max_cases=3000) >= 200


@pytest.mark.parametrize("seed", [31, 32])
def test_fuzz_consumers(seed):
    from fuzz_consumers import fuzz
    assert fuzz(seed, budget_s=6.0, max_cases=1500) >= 100


@pytest.mark.parametrize("seed", [41, 42])
def test_fuzz_normalisation_methods_on_any_dtype_and_layout(seed):
    """center / standardize / log2_norm on hand-assigned matrices of twelve dtypes, C-like / strided / column-major layouts and
    eight kinds of user vector against the three numpy statements of kmer_counts.py:165-192 on the same input
    (tests/fuzz_any_dtype.py): bytes of the result and of the replaced mean / std, in-place writes into the caller's memory
    (gaps untouched), dtypes, the warning, numpy's exceptions with their text."""
    from fuzz_any_dtype import fuzz
    assert fuzz(seed, budget_s=10.0, max_cases=8000) >= 500


@pytest.mark.parametrize("dtype,rows", [("float32", 50_000), ("float64", 30_011), ("float32", 1)])
def test_column_major_matrices_get_numpys_pairwise_column_sums(dtype, rows):
    """`DataFrame.values` of a CSV is column-major: numpy then reduces np.mean / np.std(axis=0) column by column in the
    pairwise order of its float loops — 1e-5 relative away from the row-after-row order at 50 000 float32 rows, the size
    at which the reference's C-ordered matrix needed the row order reproduced (G5).  skr_host_colstat_colmajor: mean, std and
    the centred / scaled matrix bit for bit with numpy, also for the single-column case (reduced pairwise in any order)."""
    import numpy as np
    from seekr_amd.kmer_counts import BasicCounter
    rng = np.random.default_rng(rows)
    cols = 24 if rows > 1 else 1
    n = rows if rows > 1 else 5000
    base = np.asfortranarray((rng.poisson(0.5, size=(n, cols)) * (1000.0 / 1995.0)).astype(dtype))
    assert base.flags.f_contiguous
    mine, ref = base.copy(order="F"), base.copy(order="F")
    c = BasicCounter(k=1, silent=True)
    c.counts = mine
    c.center()
    m = np.mean(ref, axis=0)
    ref -= m
    assert c.mean.dtype == m.dtype and c.mean.tobytes() == m.tobytes() and c.counts is mine and np.array_equal(mine, ref)
    c.standardize()
    with np.errstate(all="ignore"):
        s = np.std(ref, axis=0)
        ref /= s
    assert c.std.tobytes() == s.tobytes() and np.array_equal(mine, ref, equal_nan=True)
    if rows > 1 and dtype == "float32":  # the row-after-row order would NOT have been the same bits here
        seq = np.zeros(cols, np.float32)
        for row in base:
            seq = seq + row
        assert not np.array_equal(seq / np.float32(n), m)


@pytest.mark.parametrize("dtype", ["float16", "int32", "uint8", "bool"])
def test_column_major_half_and_integer_matrices_follow_numpys_pieces_too(dtype):
    """The same layout in the other dtypes, past the 8 192-element buffer piece: float16 columns are added pairwise in float32
    accumulators and rounded to half once per piece (numpy's HALF_add), integer and bool columns as the float64 values numpy
    casts them to piece by piece; np.mean / np.std bit for bit, and for integers the reference's UFuncTypeError with the
    attribute already replaced (kmer_counts.py:168-169, 174-175)."""
    import numpy as np
    from seekr_amd.kmer_counts import BasicCounter
    rng = np.random.default_rng(7)
    n, cols = 30_011, 6
    raw = rng.poisson(0.4, size=(n, cols))
    base = np.asfortranarray((raw * 0.5).astype(dtype) if dtype == "float16" else raw.astype(dtype))
    mine, ref = base.copy(order="F"), base.copy(order="F")
    c = BasicCounter(k=1, silent=True)
    c.counts = mine
    with np.errstate(all="ignore"):
        m, s = np.mean(ref, axis=0), None
        if dtype == "float16":
            c.center()
            ref -= m
            assert c.mean.dtype == m.dtype and c.mean.tobytes() == m.tobytes() and mine.tobytes("A") == ref.tobytes("A")
            c.standardize()
            s = np.std(ref, axis=0)
            ref /= s
            assert c.std.tobytes() == s.tobytes() and np.array_equal(mine, ref, equal_nan=True)
        else:
            with pytest.raises(TypeError, match="Cannot cast ufunc"):
                c.center()
            assert c.mean.dtype == m.dtype and c.mean.tobytes() == m.tobytes() and np.array_equal(mine, base)
            with pytest.raises(TypeError, match="Cannot cast ufunc"):
                c.standardize()
            s = np.std(ref, axis=0)
            assert c.std.dtype == s.dtype and c.std.tobytes() == s.tobytes()

def test_regression_rows_that_are_mostly_one_repeated_value(golden_dir):
    """Found by the fuzzer after 21 000 cases: with a 7-letter alphabet 90 % of the 4-mer columns are structurally
    zero, every zero column gets the same (hi, lo) pair under round-to-nearest, and the coherent hi*lo products
    were truncated one-sidedly by the MFMA accumulate: r = 0.956 came out 2.2e-5 off (bar 1.2e-5).  The fill
    kernel now alternates the rounding direction of the hi half by column."""
    import json
    import os
    import numpy as np
    from oracle import seekr_oracle as orc
    from seekr_amd.pearson import pearson
    case = json.load(open(os.path.join(golden_dir, "regress_sparse_rows.json")))
    raw = orc.raw_counts(case["seqs"], case["k"], case["alphabet"])
    ref = orc.normalize(raw, False, False, case["log2"])[0]
    truth = orc.pearson_f64_truth(ref, ref)
    r = pearson(ref, ref)
    err = np.abs(r - truth)
    assert (err <= 2e-6 + 1e-5 * np.abs(truth)).all(), float(err.max())
    assert abs(truth[1, 6] - 0.9564) < 1e-3 and err[1, 6] < 1.1e-5


def test_constructed_worst_case_stays_at_the_bar():
    """Near-copies of a profile whose columns are 97 % one repeated value (tools/margin_probe.py): every product is
    positive and the truncating accumulate works one-sidedly.  At K = 4 096 the error must stay inside the bar."""
    import numpy as np
    from oracle import seekr_oracle as orc
    from seekr_amd.pearson import pearson
    rng = np.random.default_rng(0)
    K, n = 4096, 48
    proto = np.zeros(K, np.float32)
    hot = rng.random(K) > 0.97
    proto[hot] = rng.integers(1, 9, int(hot.sum()))
    x = np.tile(proto, (n, 1))
    for i in range(1, n):
        idx = rng.integers(0, K, int(rng.choice([0, 1, 3, 10, 40])))
        x[i, idx] = rng.integers(0, 9, len(idx))
    truth = orc.pearson_f64_truth(x, x)
    err = np.abs(pearson(x, x) - truth)
    assert (err <= 2e-6 + 1e-5 * np.abs(truth)).all(), float((err / (2e-6 + 1e-5 * np.abs(truth))).max())


def test_constructed_worst_case_k16384_through_the_block_fill_kernel():
    """The same construction at K = 16 384 through operand_fill_block_kernel (the k = 7 kernel whenever the normalised
    counts are kept without device float32 vectors): it must raise the "mostly one repeated value" flag like the
    register kernels do, and the contraction then stays inside the bar (1.04 x the bar without the flag)."""
    import numpy as np
    from coherent_case import coherent_matrix
    from oracle import seekr_oracle as orc
    from seekr_amd import _lib
    ctx = _lib.default_context()
    x = coherent_matrix(96, 16384, 1)[48:]  # the degenerate half
    d, keep = ctx.from_numpy(x), ctx.empty(48, 16384)
    op, _ = _lib.operand_fill(ctx, d, precision=_lib.PREC_F16X3, y=keep)   # y given, no vectors -> block kernel
    assert op.kind == 2 and op.coherent
    r = ctx.empty(48, 48)
    _lib.pearson_gemm_op(ctx, op, op, r, symmetric=True)
    truth = orc.pearson_f64_truth(x, x)
    err = np.abs(r.to_numpy() - truth)
    assert (err <= 2e-6 + 1e-5 * np.abs(truth)).all(), float((err / (2e-6 + 1e-5 * np.abs(truth))).max())


def test_regression_rows_that_repeat_a_value_other_than_their_minimum(golden_dir):
    """Found 24 598 cases into round 4's soak under the strict rule: k = 4 over ACGTRYN (2 401 columns, 89 % of them
    structurally empty), Log2.pre, centred only — every row is 89 % ZEROS, but zero is not its minimum, so the fill's
    "mostly one repeated value" flag (share of the row MINIMUM) stayed down, the contraction kept 128-tile accumulator
    chunks and r = -0.4623 came out 1.07 bars from float64 and from the reference (which was 0.01 from float64).  The
    flag now also rises when 70 % of the neighbouring cells are equal, whatever the value."""
    import contextlib
    import io
    import os
    import numpy as np
    from oracle import seekr_oracle as orc
    from seekr_amd.kmer_counts import BasicCounter
    from seekr_amd.pearson import pearson
    d = np.load(os.path.join(golden_dir, "regress_r4_soak_k4_acgtryn.npz"), allow_pickle=True)
    seqs, tag = list(d["seqs"]), eval(str(d["tag"][0]))
    c = BasicCounter(silent=True, k=tag["k"], alphabet=tag["alphabet"], mean=tag["mean"], std=tag["std"], log2=tag["log2"])
    c.seqs = seqs
    with contextlib.redirect_stdout(io.StringIO()):
        c.get_counts()
    x = np.array(c.counts, dtype=np.float32)
    assert x.shape == (12, 2401) and (x[2] == 0).mean() > 0.85 and x[2].min() < 0
    with np.errstate(all="ignore"):
        ref, truth = orc.pearson(x, x).astype(np.float64), orc.pearson_f64_truth(x, x)
    got = pearson(x, x).astype(np.float64)
    ok = np.isfinite(ref)
    assert (np.abs(got - ref)[ok] <= (2e-6 + 1e-5 * np.abs(ref))[ok]).all()
    assert (np.abs(got - truth)[ok] <= 0.5 * (2e-6 + 1e-5 * np.abs(truth))[ok]).all()


def test_regression_near_copies_of_a_two_level_row(golden_dir):
    """Found by the structured-row class of tests/fuzz_pearson.py (round 4): 45 rows that are scaled, 6e-6-jittered copies of
    ONE pattern on two levels (2 062 columns at -0.993, 2 034 at +1.007 after standardisation).  Every product of two such
    rows is nearly 1, the k-tile sums that reach the MFMA's truncating accumulate are as regular as a clock, the truncations
    all go one way: r = 1.0000122, 1.01 bars from float64 (the reference: 0.04).  A standardised row on two levels meets
    kurtosis = skewness^2 + 1; the fills flag it 'coherent' and the contraction restarts its accumulators every 32 k-tiles."""
    import os
    import numpy as np
    from oracle import seekr_oracle as orc
    from seekr_amd import _lib as L
    from seekr_amd.pearson import pearson
    x = np.ascontiguousarray(np.load(os.path.join(golden_dir, "regress_r4_two_level_rows.npz"))["a"])
    assert x.shape == (16, 4096)
    with np.errstate(all="ignore"):
        ref, truth = orc.pearson(x, x).astype(np.float64), orc.pearson_f64_truth(x, x)
    got = pearson(x, x).astype(np.float64)
    assert (np.abs(got - ref) <= 2e-6 + 1e-5 * np.abs(ref)).all()
    assert (np.abs(got - truth) <= 0.5 * (2e-6 + 1e-5 * np.abs(truth))).all()
    ctx = L.default_context()
    rng = np.random.default_rng(1)
    for K in (729, 4096, 16384, 65536):            # the generic, both register-resident and the workgroup-per-row fill
        for share in (0.5, 0.3):
            base = np.where(rng.random((1, K)) < share, 2.5, -0.75)
            y = np.ascontiguousarray((base * rng.uniform(0.5, 2.0, (24, 1)) * (1 + 1e-6 * rng.standard_normal((24, K)))).astype(np.float32))
            op, _ = L.operand_fill(ctx, ctx.from_numpy(y), precision=L.PREC_F16X3, row_standardize=True)
            assert op.kind == 2 and op.coherent, (K, share)
            r = ctx.empty(24, 24)
            L.pearson_gemm_op(ctx, op, op, r, symmetric=True)
            t = orc.pearson_f64_truth(y, y)
            assert (np.abs(r.to_numpy() - t) <= 0.5 * (2e-6 + 1e-5 * np.abs(t))).all(), (K, share)
        z = rng.standard_normal((24, K)).astype(np.float32)   # three levels and continuous rows stay as they were
        op, _ = L.operand_fill(ctx, ctx.from_numpy(z), precision=L.PREC_F16X3, row_standardize=True)
        assert not op.coherent
        z3 = np.array([-1.0, 0.25, 2.0], np.float32)[rng.integers(0, 3, (24, K))]
        op, _ = L.operand_fill(ctx, ctx.from_numpy(z3), precision=L.PREC_F16X3, row_standardize=True)
        assert not op.coherent

