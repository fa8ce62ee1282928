"""Differential fuzzing of the device consumers of r (seekr_amd.consumers) against their numpy definitions:
random shapes, cutoffs, block offsets, ties, NaN, +-0, +-inf, constant rows."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekr_amd import _lib, consumers  # noqa: E402


def random_block(rng, n, m):
    r = np.clip(rng.normal(0, 0.3, (n, m)), -1, 1).astype(np.float32)
    style = rng.integers(0, 5)
    if style == 0:
        r = np.round(r, 1)                                   # many ties
    elif style == 1:
        r[rng.integers(0, n, 5), rng.integers(0, m, 5)] = np.nan
    elif style == 2:
        r[rng.integers(0, n)] = rng.choice([0.0, 0.25, -0.0])  # a constant row
        r[rng.integers(0, n, 3), rng.integers(0, m, 3)] = rng.choice([np.inf, -np.inf, 0.0, -0.0], 3)
    elif style == 3:
        r[:, rng.integers(0, m)] = r[:, rng.integers(0, m)]
    return r


def eq(a, b):
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


def fuzz(seed, budget_s=30.0, max_cases=10 ** 9):
    rng = np.random.default_rng(seed)
    ctx = _lib.default_context()
    t0, n_cases = time.time(), 0
    while time.time() - t0 < budget_s and n_cases < max_cases:
        n, m = int(rng.integers(1, 200)), int(rng.integers(1, 700))
        r = random_block(rng, n, m)
        d = ctx.from_numpy(r)
        cutoff = float(rng.choice([-2.0, -0.1, 0.0, 0.05, 0.3, 0.99, 5.0, rng.uniform(-1, 1)]))
        row0 = int(rng.integers(0, 50)) if rng.integers(0, 2) else 0  # global id of local row 0 (diagonal placement)
        c0 = int(rng.integers(0, m))
        c1 = int(rng.integers(c0, m + 1))
        upper = bool(rng.integers(0, 2))
        tag = dict(n=n, m=m, cutoff=cutoff, row0=row0, c0=c0, c1=c1, upper=upper)
        # ---- edges of the block r[:, c0:c1] placed at global rows row0.., global columns 0..
        full = np.zeros((row0 + n, m), np.float32)
        full[row0:, c0:c1] = r[:, c0:c1]
        with np.errstate(invalid="ignore"):
            full[full < cutoff] = 0
        for g in range(min(full.shape)):
            full[g, g] = 0
        if upper:
            full = np.triu(full, 1)
        wi, wj = np.nonzero(full)
        i, j, v = consumers.edges(d, cutoff, col_begin=c0, col_end=c1, row_global0=row0, upper_only=upper)
        assert np.array_equal(i, wi.astype(np.uint32)) and np.array_equal(j, wj.astype(np.uint32)) and eq(v, full[wi, wj]), ("edges", tag)
        # ---- per-row top-k, diagonal cell excluded
        k = int(rng.integers(1, 40))
        idx, val = consumers.topk_rows(d, k, row_global0=row0)
        for a in range(n):
            cand = np.array([c for c in range(m) if c != row0 + a], dtype=np.int64)
            order = cand[np.argsort(-r[a][cand], kind="stable")][:k] if len(cand) else cand
            want = np.full(k, 0xFFFFFFFF, np.uint32)
            want[:len(order)] = order
            assert np.array_equal(idx[a], want), ("topk idx", tag, a, idx[a][:6], want[:6])
            assert eq(val[a][:len(order)], r[a][order]) and np.isnan(val[a][len(order):]).all(), ("topk val", tag, a)
        # ---- threshold + zero diagonal in place (square placement at column offset)
        if rng.integers(0, 3) == 0:
            dc0 = int(rng.integers(0, max(1, m - n + 1)))
            want = r.copy()
            with np.errstate(invalid="ignore"):
                want[want < cutoff] = 0
            for a in range(n):
                if a + dc0 < m:
                    want[a, a + dc0] = 0
            got = consumers.threshold_zero_diag(ctx.from_numpy(r), cutoff, dc0).to_numpy()
            assert eq(got, want), ("threshold", tag, dc0)
        # ---- empirical p-values
        if rng.integers(0, 3) == 0:
            bg = rng.normal(0, 0.3, int(rng.integers(1, 3000))).astype(np.float32)
            if rng.integers(0, 2):
                bg[: min(len(bg), m)] = r[0][: min(len(bg), m)]  # ties with values of r
            if rng.integers(0, 3) == 0:
                bg[rng.integers(0, len(bg))] = np.nan
            with np.errstate(invalid="ignore"):
                want = (bg[None, None, :] > r[:, :, None]).sum(axis=2).astype(np.float32) / np.float32(len(bg)) if n * m * len(bg) < 3e7 else None
            if want is not None:
                got = consumers.empirical_pvalues(d, bg).to_numpy()
                assert np.array_equal(got, (((bg[None, None, :] > r[:, :, None]).sum(axis=2)) / len(bg)).astype(np.float32)), ("pvalues", tag)
        # ---- upper triangle flatten (square part)
        if n == min(n, m) and rng.integers(0, 3) == 0:
            sq = ctx.from_numpy(np.ascontiguousarray(r[:, :n]))
            kk = int(rng.integers(0, 3))
            got = consumers.triu_values(sq, kk).to_numpy().reshape(-1)
            assert eq(got, r[:, :n][np.triu_indices(n, k=kk)]), ("triu", tag, kk)
        n_cases += 1
    return n_cases


if __name__ == "__main__":
    n = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
    print("consumer fuzz ok: %d cases" % n)
