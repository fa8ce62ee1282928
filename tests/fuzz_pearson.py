"""Differential fuzzing of seekr_amd.pearson.pearson(a, b, row_standardize) against the oracle:
random shapes (K not a multiple of anything, single rows, empty-ish), dtypes (float32, float64,
integers -> float64 path), same-object and different operands, NaN / constant rows."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import seekr_oracle as orc  # noqa: E402
from seekr_amd.pearson import pearson  # noqa: E402


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from strict_tally import StrictTally  # noqa: E402
import parity_rule  # noqa: E402

TALLY = StrictTally()

def gen_case(rng):
    """One random case of the pearson() API: (a, b, row_standardize, tag)."""
    K = int(rng.choice([1, 2, 3, 4, 5, 16, 31, 32, 33, 64, 100, 255, 256, 257, 625, 729, 1000, 1023, 1024, 1025, 1500, 2048, 3125,
                        4096, 4100, 16384]))  # (this list is frozen: tests/golden/refspread.json stores states of this stream)
    M, N = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    dt = rng.choice(["f32", "f32", "f32", "f64", "i64"])
    same = bool(rng.integers(0, 3) == 0)
    rs = bool(rng.integers(0, 4) != 0)
    kind = rng.integers(0, 6)
    def make(rows):
        if kind == 0:
            x = rng.standard_normal((rows, K)) * rng.uniform(0.1, 50)
        elif kind == 1:
            x = rng.binomial(20, 0.2, size=(rows, K)).astype(np.float64) * 1.5
        elif kind == 2:
            x = np.abs(rng.standard_normal((rows, K))) ** 3
        elif kind == 3:
            x = rng.standard_normal((rows, K)); x[rng.integers(0, rows)] = 3.25  # a constant row
        elif kind == 4:
            # count-like and sparse: one-hot rows (homopolymers), a few hot columns, and rows repeated
            # exactly or scaled — pairs with r = 1 whose products are dominated by one column
            x = np.zeros((rows, K))
            for i in range(rows):
                hot = rng.integers(0, K, size=int(rng.choice([1, 1, 2, 3, 8])))
                x[i, hot] = rng.uniform(1, 12, size=len(hot))
                if rng.integers(0, 3) == 0:
                    x[i] += rng.binomial(3, 0.1, K) * 0.5
            for i in range(1, rows):
                if rng.integers(0, 4) == 0:
                    x[i] = x[rng.integers(0, i)] * rng.choice([1.0, 2.0, 0.37])
        else:
            base = rng.standard_normal((1, K)) * 3
            x = base + rng.standard_normal((rows, K)) * rng.choice([1e-3, 1e-2, 0.3])  # near-duplicates: r ~ 1
        if dt == "i64":
            return np.rint(x * 4).astype(np.int64)
        return x.astype(np.float32 if dt == "f32" else np.float64)
    a = make(M)
    b = a if same else make(N)
    tag = dict(K=K, M=a.shape[0], N=b.shape[0], dt=str(dt), same=same, rs=rs, kind=int(kind))
    return a, b, rs, tag


def gen_case_wide(rng):
    """k = 8 rows (65 536 columns: sixteen accumulator chunks of the split contraction since round 4), float32, row-standardised
    — and, since round 5, the widths above 8 192 columns that the block / register-row fills serve (8 200: a multiple of 8
    that is not a multiple of 32, where the block fill read past the row until round 5; 10^4, 5^6, 14^4).
    A generator of its own — gen_case's random stream is what tests/golden/refspread.json stores states of."""
    K = int(rng.choice([65536, 65536, 8200, 10000, 15625, 38416]))
    M, N = int(rng.integers(1, 24)), int(rng.integers(1, 24))
    same = bool(rng.integers(0, 2))
    kind = int(rng.choice([0, 1, 5]))
    def make(rows):
        if kind == 0:
            x = rng.standard_normal((rows, K)) * rng.uniform(0.1, 50)
        elif kind == 1:
            x = rng.binomial(20, 0.2, size=(rows, K)).astype(np.float64) * 1.5
        else:
            x = rng.standard_normal((1, K)) * 3 + rng.standard_normal((rows, K)) * rng.choice([1e-3, 1e-2, 0.3])
        return x.astype(np.float32)
    a = make(M)
    b = a if same else make(N)
    return a, b, True, dict(K=K, M=a.shape[0], N=b.shape[0], dt="f32", same=same, rs=True, kind=kind)


LAYOUT_WIDTHS = [4096, 4096, 16384]
if os.environ.get("SEEKR_FUZZ_LAYOUT_WIDTHS"):  # e.g. 64,256,1024,4096,16384,65536: every width the split-fp16 kernel serves
    LAYOUT_WIDTHS = [int(t) for t in os.environ["SEEKR_FUZZ_LAYOUT_WIDTHS"].split(",")]
LAYOUT_CROSS_STRUCTURE = os.environ.get("SEEKR_FUZZ_LAYOUT_CROSS", "0") == "1"  # b drawn from ANOTHER structure class than a


def gen_case_layout(rng):
    """Rows with STRUCTURE (float32, row-standardised): short periods, a few jittered levels, few distinct values, levels
    aligned between rows, quantised values, and mixtures of those with plain gaussian rows — the inputs on which the fp8
    roundings of the opt-in f16f8 layout do not average out and its fill has to route the operand back to the three-product
    split (tools/f8_cross_study.py), and on which, under the default precision, the coherent-row flag and the accumulator
    chunks earn their keep.  By default at the two widths that can take the f16f8 layout (4 096 / 16 384 columns: the
    stream tests/golden and the round-4 soaks were drawn from); SEEKR_FUZZ_LAYOUT_WIDTHS names any others (round 5: every
    width the split-fp16 kernel serves, 64 ... 65 536), SEEKR_FUZZ_LAYOUT_CROSS=1 draws the second operand of a cross
    comparison from a different structure class than the first."""
    K = int(rng.choice(LAYOUT_WIDTHS))
    cap = 48 if K <= 16384 else 20
    M, N = int(rng.integers(2, cap)), int(rng.integers(2, cap))
    same = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 7))
    jit = float(10.0 ** rng.uniform(-6.5, -1.0))
    kinds = [kind]
    def make(rows):
        kind = kinds[0]
        if kind == 0:    # periodic rows, shifted copies of one sequence
            P = int(rng.integers(2, 6000))
            vals = rng.standard_normal(P)
            x = vals[(np.arange(K)[None, :] + rng.integers(0, P, (rows, 1))) % P] * (1.0 + jit * rng.standard_normal((rows, K)) * (rng.integers(0, 2)))
        elif kind == 1:  # a few levels, jittered
            L = int(rng.integers(2, 33))
            x = (rng.standard_normal(L) * 2)[rng.integers(0, L, (rows, K))] * (1.0 + jit * rng.standard_normal((rows, K)))
        elif kind == 2:  # D distinct values at random
            D = int(2 ** rng.uniform(1, 13))
            x = rng.standard_normal(D)[rng.integers(0, D, (rows, K))]
        elif kind == 3:  # the SAME level in a column for every row, rows scaled and jittered
            L = int(rng.integers(2, 17))
            x = (rng.standard_normal(L) * 2)[rng.integers(0, L, (1, K))] * rng.uniform(0.5, 2.0, (rows, 1)) * (1.0 + jit * rng.standard_normal((rows, K)))
        elif kind == 4:  # quantised gaussian
            q = 2.0 ** int(rng.integers(-8, 3))
            x = np.rint(rng.standard_normal((rows, K)) * 3 / q) * q
        elif kind == 5:  # count levels jittered by column statistics (what the pipeline hands over), tighter than real
            c = rng.binomial(int(rng.integers(200, 4000)), 1.0 / K * rng.uniform(0.5, 8), (rows, K)).astype(np.float64)
            m = 0.3 * (1.0 + jit * rng.standard_normal((1, K)))
            x = (np.log2(c + 1.0) - m) / (0.45 * (1.0 + jit * rng.standard_normal((1, K))))
        else:            # a smooth ramp or wave plus noise
            t = np.arange(K)[None, :] / K
            x = np.sin(2 * np.pi * t * rng.uniform(0.5, 40, (rows, 1)) + rng.uniform(0, 6, (rows, 1))) + jit * rng.standard_normal((rows, K))
        x = np.array(x, dtype=np.float64)
        if rng.integers(0, 3) == 0:  # some rows plain gaussian: ONE structured row has to be enough
            plain = rng.integers(0, 2, rows).astype(bool)
            x[plain] = rng.standard_normal((int(plain.sum()), K))
        return x.astype(np.float32)
    a = make(M)
    if not same and LAYOUT_CROSS_STRUCTURE:
        kinds[0] = int(rng.integers(0, 7))
    b = a if same else make(N)
    return a, b, True, dict(K=K, M=a.shape[0], N=b.shape[0], dt="f32", same=same, rs=True, kind=100 + kind, kind_b=100 + kinds[0],
                            jitter=jit)


LAYOUT_EVERY = 3 if os.environ.get("SEEKR_PRECISION", "") == "f16f8" else 40
if os.environ.get("SEEKR_FUZZ_LAYOUT_EVERY"):  # a soak of the structured class alone: SEEKR_FUZZ_LAYOUT_EVERY=2
    LAYOUT_EVERY = max(2, int(os.environ["SEEKR_FUZZ_LAYOUT_EVERY"]))


def fuzz(seed, budget_s=60.0, max_cases=10 ** 9):
    rng = np.random.default_rng(seed)
    t0, n_cases = time.time(), 0
    while time.time() - t0 < budget_s and n_cases < max_cases:
        pick = int(rng.integers(0, 200 * LAYOUT_EVERY))
        a, b, rs, tag = gen_case_wide(rng) if pick == 0 else gen_case_layout(rng) if pick % LAYOUT_EVERY == 1 else gen_case(rng)
        same = tag["same"]
        try:
            with np.errstate(all="ignore"):
                want = orc.pearson(a, b, rs)
                truth = orc.pearson_f64_truth(a, b, rs)
            got = pearson(a, b, row_standardize=rs)
            assert got.dtype == want.dtype and got.shape == want.shape, ("dtype/shape", tag, got.dtype, want.dtype)
            if n_cases % 8 == 0 and a.size <= 1 << 20:
                # the same VALUES in another memory layout (column-major: DataFrame.values of a CSV; a strided view): the
                # result does not depend on how the caller's matrix lies in memory (a generator of its own: gen_case's stream is frozen)
                lay = np.random.default_rng([seed, n_cases])
                def relaid(x):
                    return np.asfortranarray(x) if lay.integers(0, 2) else np.repeat(x, 2, axis=1)[:, ::2]
                a2 = relaid(a)
                b2 = a2 if b is a else relaid(b)
                again = pearson(a2, b2, row_standardize=rs)
                assert again.dtype == got.dtype and again.tobytes() == got.tobytes(), ("layout of the input changed r", tag)
            with np.errstate(all="ignore"):
                g = np.where(np.isinf(got), np.nan, got).astype(np.float64)
                w = np.where(np.isinf(want), np.nan, want).astype(np.float64)
            # rows that are constant (or nearly: std below the rounding noise of the mean) are garbage in,
            # NaN / inf / anything out, for numpy as for the device: judge the others
            def bad_rows(x):
                x = np.asarray(x, dtype=np.float64)
                return x.std(axis=1) <= 1e-6 * np.maximum(np.abs(x).max(axis=1), 1e-300) if rs else np.zeros(len(x), bool)
            live = ~bad_rows(a)[:, None] & ~bad_rows(b)[None, :]
            assert not np.isnan(g[live]).any() or np.isnan(w[live]).any(), ("unexpected NaN", tag)
            ok = live & ~np.isnan(w) & ~np.isnan(truth)
            e_ours = np.where(ok, np.abs(g - truth), 0.0)
            scale = np.maximum(np.abs(np.where(ok, truth, 0.0)), 1e-300)
            if got.dtype == np.float64:
                # float64 inputs: oracle.pearson IS the float64 evaluation; only the summation order differs
                limit = 1e-12 * np.maximum(scale, 1.0)
                bad = e_ours > limit
                assert not bad.any(), ("value", tag, float(e_ours[bad].max()))
            else:
                # float32: strict against the reference's float32 result; a cell may leave it only where the float32
                # inner product is order-sensitive (an input property, tests/parity_rule.py), and must then be within
                # the bar of float64.  Without row standardisation r scales with the operands: so does the bar.
                unit = 1.0 if rs else float(np.maximum(scale.max(), 1.0))
                verdict = parity_rule.judge(g, w, truth, ok, a, b, row_standardize=rs, unit=unit)
                if rs:
                    TALLY.add(verdict, tag)
                assert not verdict["failures"], ("value", tag, verdict["failures"][:3])
        except AssertionError:
            d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(d, exist_ok=True)
            np.savez(os.path.join(d, "fuzz_pearson_fail_%d_%d.npz" % (seed, n_cases)), a=a, b=b, rs=rs, same=same)
            raise
        n_cases += 1
    return n_cases


if __name__ == "__main__":
    n = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("pearson fuzz ok: %d cases" % n)
