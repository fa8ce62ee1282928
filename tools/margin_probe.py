"""How much of the parity bar |dr| <= 2e-6 + 1e-5 |r| the default fp16 x 3 contraction uses on its constructed worst
case: near-copies of one profile (r ~ 1, so every product is positive) in which 50-97 % of the columns hold one
repeated value — the truncating MFMA accumulate then loses up to an ulp of the running sum on each of the 128 adds of
a 4 096-column chunk, all in the same direction.  Prints max error / bar per shape, next to numpy float32's own."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import seekr_oracle as orc
from seekr_amd.pearson import pearson
rng = np.random.default_rng(0)
worst = 0
for K in (1024, 2401, 4096, 16384):
    for frac in (0.5, 0.8, 0.9, 0.97):
        for base_val in (0.0, 1.0):
            n = 48
            proto = np.full(K, base_val, np.float32)
            hot = rng.random(K) > frac
            proto[hot] = rng.integers(1, 9, int(hot.sum()))
            x = np.tile(proto, (n, 1))
            for i in range(1, n):   # near copies: a few cells changed
                idx = rng.integers(0, K, int(rng.choice([0, 1, 3, 10, 40])))
                x[i, idx] = rng.integers(0, 9, len(idx))
            truth = orc.pearson_f64_truth(x, x)
            r = pearson(x, x)
            e = np.abs(r - truth); bar = 2e-6 + 1e-5 * np.abs(truth)
            m = float((e / bar).max()); worst = max(worst, m)
            print("K=%5d identical fraction %.2f base %.0f: max err %.2e  err/bar %.2f  (numpy f32: %.2f)" % (
                K, frac, base_val, e.max(), m, float((np.abs(orc.pearson(x, x) - truth) / bar).max())))
print("worst err/bar", worst)
