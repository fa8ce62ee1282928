"""The routing rules of the opt-in f16f8 operand layout, checked WITHOUT a GPU on the numpy emulation of that layout
(tools/f8_cross_study.py: fp16 hi / lo, OCP e4m3 copies, sums in float64): every structured input whose emulated error
exceeds 0.6 of the bar is routed back to the three-product split by one of the statistics the fill kernel computes
(equal neighbours, distinct values, row means of the rounding residues, the row's own cross-term error), and unstructured
rows keep the layout.  The kernel's own decisions on the same classes: tests/test_gpu_parity.py (-m gpu)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import f8_cross_study as study  # noqa: E402


def test_e4m3_emulation_known_values():
    e = study.e4m3
    assert e(0.3) == 0.3125 and e(-0.3) == -0.3125          # 3 mantissa bits: step 2^-5 in [0.25, 0.5)
    assert e(448.0) == 448.0 and e(500.0) == 448.0 and e(-1e9) == -448.0   # saturating, no inf (OCP e4m3fn)
    assert e(2.0 ** -9) == 2.0 ** -9 and e(2.0 ** -11) == 0.0               # subnormal step 2^-9, ties to even
    assert e(17.0) == 16.0 and e(19.0) == 20.0 and e(18.0) == 18.0 and e(21.0) == 20.0 and e(23.0) == 24.0   # ties to even in [16, 32): step 2
    x = np.linspace(-400, 400, 10001)
    assert np.all(np.abs(e(x) - x) <= np.maximum(np.abs(x), 2.0 ** -6) * 2.0 ** -4 + 1e-12)


def routed(x, K):
    amax, amin = study.adjacent_equal(x)
    occ = study.bitmap_share(x, K)
    means, own = study.emulate(x, K, want_means=True)
    return amin >= 1.0 / 256.0 or occ * K < 2048 or means > 0.6 or own > 1.0


def test_every_emulated_hazard_is_routed_and_plain_rows_are_kept():
    K, n = 4096, 48
    rng = np.random.default_rng(11)
    cases = [("gaussian", rng.standard_normal((n, K)))]
    for D in (3, 16, 64):
        cases.append(("%d values" % D, rng.standard_normal(D)[rng.integers(0, D, (n, K))]))
    for P in (3, 64, 513, 1025):
        cases.append(("period %d" % P, rng.standard_normal(P)[(np.arange(K)[None, :] + rng.integers(0, P, (n, 1))) % P]))
    for L, jit in ((2, 1e-4), (4, 1e-5), (16, 1e-5), (4, 1e-2), (16, 1e-3)):
        cases.append(("%d levels jitter %g" % (L, jit), (rng.standard_normal(L) * 2)[rng.integers(0, L, (n, K))] * (1 + jit * rng.standard_normal((n, K)))))
    for L, jit in ((2, 3e-5), (4, 3e-5)):
        cases.append(("%d aligned levels jitter %g" % (L, jit), (rng.standard_normal(L) * 2)[rng.integers(0, L, (1, K))] * rng.uniform(0.5, 2, (n, 1)) * (1 + jit * rng.standard_normal((n, K)))))
    cases.append(("Poisson(2)", rng.poisson(2.0, (n, K))))
    kept_errors = []
    for name, x in cases:
        x = np.asarray(x, dtype=np.float32)
        err8, err3 = study.emulate(x, K)
        assert err3 <= 0.1, (name, err3)                       # the three-product operands never come close to the bar
        if err8 > 0.6:
            assert routed(x, K), (name, err8)
        if not routed(x, K):
            kept_errors.append((name, err8))
    assert ("gaussian" in [k for k, _ in kept_errors]) and max(e for _, e in kept_errors) <= 0.6, kept_errors


def test_two_level_statistic_separates_two_point_rows():
    """The flag the fill kernels raise for rows on two tight levels (operand.hip: row_on_two_levels): for a standardised
    row, kurtosis - skewness^2 - 1 >= 0 with equality exactly for a two-point distribution.  The numpy mirror of the
    device arithmetic (float32 z, sums of z^2, z^3, z^4) on the classes the GPU test uses: two levels at any share with a
    jitter up to 1 % fall below the 5e-3 threshold, three or more levels and continuous rows stay far above it."""
    rng = np.random.default_rng(4)

    def stat(x):
        x = np.asarray(x, np.float32)
        z = ((x - x.mean()) / x.std()).astype(np.float32).astype(np.float64)
        var = (z ** 2).mean()
        return (z ** 4).mean() / var ** 2 - ((z ** 3).mean() / var ** 1.5) ** 2 - 1.0

    for K in (729, 4096, 16384, 65536):
        for share in (0.5, 0.3, 0.1):
            for jitter in (0.0, 1e-6, 1e-3, 1e-2):
                base = np.where(rng.random(K) < share, 2.5, -0.75) * (1 + jitter * rng.standard_normal(K))
                assert stat(base) < 5e-3, (K, share, jitter, stat(base))
        assert stat(np.array([-1.0, 0.25, 2.0])[rng.integers(0, 3, K)]) > 0.1
        assert stat(rng.standard_normal(K)) > 1.0
        assert stat(rng.poisson(0.5, K)) > 0.5
        assert stat(np.where(rng.random(K) < 0.5, 1.0, -1.0) * (1 + 0.1 * rng.standard_normal(K))) > 5e-3
