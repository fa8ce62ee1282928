#!/usr/bin/env python3
"""Golden vectors for the parametric p-values (find_pval.py:118-133): p = 1 - scipy.stats.<dist>(*params).cdf(sim)
written into a float32 matrix, for every distribution of find_dist's `common10` list (find_dist.py:96-98), with
parameters of the kind a fit to Pearson similarities returns and a few harder ones (large shapes, heavy tails, values
outside the support).  scipy is not the reference repo, but it is the library the reference calls; run anywhere scipy is
installed:

    python3 tests/golden/make_golden_pvals.py
"""
import os

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [
    ("norm", (0.01, 0.08)), ("norm", (-0.3, 1.7)),
    ("cauchy", (0.005, 0.03)),
    ("expon", (-0.4, 0.25)),
    ("uniform", (-0.5, 1.2)),
    ("rayleigh", (-0.35, 0.3)),
    ("pareto", (3.1, -1.2, 1.0)), ("pareto", (12.5, -2.0, 1.9)),
    ("lognorm", (0.35, -0.6, 0.55)), ("lognorm", (1.4, -0.2, 0.1)),
    ("gamma", (2.3, -0.4, 0.12)), ("gamma", (45.0, -1.5, 0.033)), ("gamma", (0.6, -0.31, 0.2)),
    ("chi2", (3.7, -0.45, 0.11)), ("chi2", (180.0, -3.0, 0.0165)),
    ("exponpow", (1.8, -0.42, 0.6)), ("exponpow", (0.7, -0.3, 0.4)),
]


def main():
    rng = np.random.default_rng(7)
    sim = np.concatenate([rng.normal(0.0, 0.12, 3000), np.linspace(-1, 1, 801), [np.nan, -1.0, 1.0, 0.0]]).astype(np.float32)
    sim = np.clip(sim, -1, 1).reshape(-1, 5)[:760]
    out = {"sim": sim}
    for i, (name, params) in enumerate(CASES):
        dist = getattr(stats, name)(*params)
        p = np.zeros_like(sim)
        flat_p, flat_s = p.reshape(-1), sim.reshape(-1)
        for j in range(flat_s.size):  # the reference's loop: scalar float32 in, float64 arithmetic, float32 store
            flat_p[j] = 1 - dist.cdf(flat_s[j])
        out["p%d" % i] = p
    out["names"] = np.array([c[0] for c in CASES])
    out["params"] = np.array([",".join(repr(float(v)) for v in c[1]) for c in CASES])
    np.savez_compressed(os.path.join(HERE, "pvals_common10.npz"), **out)
    print("wrote pvals_common10.npz:", len(CASES), "cases,", sim.size, "cells each")


if __name__ == "__main__":
    main()
