"""How far the order-sensitivity clause of tests/parity_rule.py reaches (no GPU needed): the share
of cells of a Pearson matrix on which at least one of the nine float32 summation orders lies TAU bars or more from float64
— i.e. on which a strict failure WOULD be excused — for TAU = 0.1 / 0.2 / 0.5 / 1.0, on config-2 data (synthetic 2 kb
transcripts, k = 6): the normalised counts of the default pipeline, and the raw per-kb counts (`-uc -us -l Log2.none`).
The clause is only ever evaluated on cells that already failed strict parity; this measures the ground it could cover.

    python tools/clause_reach.py [rows=160] [cols_sampled=384]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import seekr_oracle as orc  # noqa: E402
import parity_rule  # noqa: E402

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 160
n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 384
seqs = orc.codes_to_seqs(orc.synthetic_codes(2, max(n_rows, n_cols), 2000))
raw = orc.raw_counts(seqs, 6)
norm = orc.normalize(raw)[0]
for name, x in (("normalised counts (Log2.post, centred, standardised)", norm), ("raw per-kb counts (-uc -us -l Log2.none)", raw)):
    a, b = x[:n_rows], x[:n_cols]
    truth = orc.pearson_f64_truth(a, b)
    ref = orc.pearson(a, b).astype(np.float64)
    cells = np.argwhere(np.ones_like(truth, dtype=bool))
    sens = parity_rule.order_sensitivity(a, b, cells, truth).reshape(truth.shape)
    off = ~np.eye(*truth.shape, dtype=bool)
    ref_err = np.abs(ref - truth) / parity_rule.bar_of(truth)
    print("%s: %d x %d cells" % (name, n_rows, n_cols))
    print("   largest order-sensitivity %.3f bars (off the diagonal %.3f); the reference itself is at most %.3f bars from float64"
          % (sens.max(), sens[off].max(), ref_err.max()))
    for tau in (0.1, 0.2, 0.5, 1.0):
        print("   TAU = %.1f: %6.2f %% of the cells would qualify (%6.2f %% off the diagonal)"
              % (tau, 100.0 * (sens >= tau).mean(), 100.0 * (sens[off] >= tau).mean()))
