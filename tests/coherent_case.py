"""The split contraction's constructed worst case (tools/margin_probe.py) as a row-sharded matrix: the rows of the
LAST rank are near-copies of one profile in which 97 % of the columns hold one repeated value (r ~ 1, every product
positive, the truncating MFMA accumulate loses up to an ulp per add, all in one direction); every other rank holds
ordinary count-like rows."""
import numpy as np


def coherent_matrix(n_total, cols, size):
    rng = np.random.default_rng(12)
    x = (rng.binomial(40, 0.05, size=(n_total, cols)) * np.float32(0.5)).astype(np.float32)
    first = (n_total // size) * (size - 1) if size > 1 else n_total // 2
    proto = np.zeros(cols, np.float32)
    hot = rng.random(cols) > 0.97
    proto[hot] = rng.integers(1, 9, int(hot.sum()))
    x[first:] = proto
    for i in range(first + 1, n_total):  # near copies: a few cells changed
        idx = rng.integers(0, cols, int(rng.choice([0, 1, 3, 10, 40])))
        x[i, idx] = rng.integers(0, 9, len(idx))
    return x
