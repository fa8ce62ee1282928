"""numpy emulation of the opt-in f16f8 operand (operand.hip, X8 lines) — which INPUTS keep its error inside the bar.

hi = fp16(z s) rounded to nearest, lo = fp16(z s - hi), h8 = e4m3(hi / 128), l8 = e4m3(16 lo);
r = (hi.hi + 8 (h8.l8 + l8.h8)) / (K s^2) — the sums in float64 here, so what is measured is the operands' own rounding
(the lost lo.lo term and the fp8 roundings of the cross terms), not the accumulation order.  Rows with D distinct values
placed at random, rows of a short period, rows whose values sit on a few tightly jittered levels: the fp8 rounding of a value
is the same wherever it stands, so the error of a cell is a sum over value PAIRS, not over columns — the cases the fill's
three routing statistics (equal neighbours, distinct values, row means of the rounding residues) have to find.

    python tools/f8_cross_study.py [--cols 4096] [--rows 192]"""
import argparse
import numpy as np


def e4m3(x):
    """OCP e4m3 (gfx950's fp8), round to nearest even, saturating at 448, subnormal step 2^-9."""
    x = np.asarray(x, np.float64)
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    q = np.round(a / step) * step          # np.round is half-to-even
    q = np.minimum(q, 448.0)
    return np.sign(x) * q


def emulate(x, cols, want_means=False):
    x = np.asarray(x, np.float32)
    z = ((x.T - x.mean(1)).T)
    z = ((z.T / z.std(1)).T).astype(np.float32)
    s = 2.0 ** np.floor(np.log2(32768.0 / np.sqrt(cols)))
    zs = z.astype(np.float64) * s
    hi = zs.astype(np.float16).astype(np.float64)
    lo = (zs - hi).astype(np.float16).astype(np.float64)
    h8, l8 = e4m3(hi / 128.0), e4m3(lo * 16.0)
    if want_means:
        # the fill's third statistic: the largest row means of what the fp8 copies hold and lose -> a bound on the error
        # of a cell from the means alone, in units of the bar at r = 0 (skr_operand_fill routes above 0.6)
        dh, dl = hi - 128.0 * h8, lo - l8 / 16.0
        m = [np.abs(v.mean(1)).max() for v in (dh, lo, dl)]
        means = 2.0 * (m[0] * (m[1] + m[2]) + (m[0] + m[1]) * m[2]) / (s * s) / 2e-6
        # the fourth: the error of a row's cross term WITH ITSELF (rows whose levels are aligned meet in the same pairs), as a
        # multiple of its threshold 3e-6
        own = 2.0 * np.abs((hi * lo - (hi - dh) * (lo - dl)).mean(1)).max() / (s * s) / 3e-6
        return means, own
    r8 = (hi @ hi.T + 8.0 * (h8 @ l8.T + l8 @ h8.T)) / (cols * s * s)
    r3 = (hi @ hi.T + hi @ lo.T + lo @ hi.T) / (cols * s * s)
    truth = (z.astype(np.float64) @ z.astype(np.float64).T) / cols
    bar = 2e-6 + 1e-5 * np.abs(truth)
    off = ~np.eye(len(x), dtype=bool)
    return float((np.abs(r8 - truth) / bar)[off].max()), float((np.abs(r3 - truth) / bar)[off].max())


def bitmap_share(x, cols):
    """the fill's second statistic (round 4): the row's standardised values hashed into 2 K bits; occupied bits / K, worst row"""
    x = np.asarray(x, np.float32)
    z = ((x.T - x.mean(1)).T)
    z = np.ascontiguousarray((z.T / z.std(1)).T, dtype=np.float32)
    logb = int(np.log2(2 * cols))
    h = ((z.view(np.uint32).astype(np.uint64) * 0x9E3779B1) & 0xFFFFFFFF) >> (32 - logb)
    return min(len(np.unique(row)) for row in h) / cols


def adjacent_equal(x):
    """the fill's own statistic: equal neighbours inside groups of four cells, worst row (as a share of K)"""
    g = x.reshape(len(x), -1, 4)
    eq = (g[:, :, 0] == g[:, :, 1]).sum(1) + (g[:, :, 1] == g[:, :, 2]).sum(1) + (g[:, :, 2] == g[:, :, 3]).sum(1)
    return float(eq.max()) / x.shape[1], float(eq.min()) / x.shape[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--rows", type=int, default=192)
    args = ap.parse_args()
    K, n = args.cols, args.rows
    rng = np.random.default_rng(1)
    print("K = %d, %d rows; worst off-diagonal cell in bars vs float64: f16f8 | f16x3 operands; adjacent-equal share (max, min row); "
          "occupied bits of the 2 K-bit value bitmap / K (worst row); the means bound in bars; the row's own cross-term error / 3e-6; routed by any rule (adjacent-equal >= K/256, fewer than 2 048 occupied bits, means bound > 0.6, own error > 1)" % (K, n))
    cases = []
    cases.append(("gaussian (all distinct)", rng.standard_normal((n, K)).astype(np.float32)))
    for D in (8192, 4096, 2048, 1024, 512, 256, 128, 64, 16, 3):
        vals = rng.standard_normal(D).astype(np.float32)
        cases.append(("%5d values drawn at random" % D, vals[rng.integers(0, D, (n, K))]))
    for D in (1000, 300, 100):
        cases.append(("integers 0..%d" % (D - 1), rng.integers(0, D, (n, K)).astype(np.float32)))
    for lam in (200.0, 50.0, 10.0, 2.0):
        cases.append(("Poisson(%g) counts" % lam, rng.poisson(lam, (n, K)).astype(np.float32)))
    for P in (3, 7, 64, 513, 1025, 2049, 2900):
        vals = rng.standard_normal(P).astype(np.float32)
        idx = (np.arange(K)[None, :] + rng.integers(0, P, (n, 1))) % P
        cases.append(("period %d, no equal neighbours" % P, vals[idx]))
    for nc in (2, 4, 16):
        centres = (rng.standard_normal(nc) * 2).astype(np.float32)
        for jit in (1e-2, 1e-3, 3e-4, 1e-4, 1e-5):
            x = centres[rng.integers(0, nc, (n, K))] * (1.0 + jit * rng.standard_normal((n, K)))
            cases.append(("%2d levels, jitter %.0e (all distinct)" % (nc, jit), x.astype(np.float32)))
    for nc in (2, 4, 16):
        centres = (rng.standard_normal(nc) * 2).astype(np.float32)
        for jit in (1e-3, 3e-5):
            x = centres[rng.integers(0, nc, (1, K))] * rng.uniform(0.5, 2.0, (n, 1)) * (1.0 + jit * rng.standard_normal((n, K)))
            cases.append(("%2d ALIGNED levels, jitter %.0e" % (nc, jit), x.astype(np.float32)))
    for name, x in cases:
        e8, e3 = emulate(x, K)
        mb, own = emulate(x, K, want_means=True)
        amax, amin = adjacent_equal(x)
        occ = bitmap_share(x, K)
        routed = amin >= 1.0 / 256.0 or occ * K < 2048 or mb > 0.6 or own > 1.0
        print("%-36s %7.3f | %6.3f   adj-eq %.4f / %.4f  bitmap %.3f  means %6.3f  own %5.2f  %s%s" % (name, e8, e3, amax, amin, occ, mb, own, "routed" if routed else "KEPT",
                                                                   "  <-- over the bar, not routed" if (e8 > 1.0 and not routed) else ""))


if __name__ == "__main__":
    main()
