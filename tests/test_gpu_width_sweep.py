"""Normalisation and Pearson at several hundred widths (tools/width_sweep.py, the thinned list): every width 1 .. 130, the
neighbours of every power of two and of the kernels' own thresholds up to 70 000 columns, multiples of 8 that are not
multiples of 32, seeded random widths.  Column mean / std / normalised matrix bit-exact against the oracle
(kmer_counts.py:201-209), the fused fill == the elementwise kernel bit for bit and nothing written outside its rows, r by
the parity rule (pearson.py:35-41).  Round 5: the class of bug that lives at a width nobody listed (the block fill at
8 200 columns).  Needs a real MI355X: run with `-m gpu`."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_every_kernel_behind_normalisation_and_pearson_across_widths():
    import width_sweep
    ws = width_sweep.widths(quick=True, seed=1)
    assert len(ws) > 300 and {8200, 10000, 15625, 38416, 65536} <= set(ws)
    bad = width_sweep.sweep(ws, seed=1, verbose=False)
    assert not bad, bad


def test_row_counts_around_the_tile_heights():
    import width_sweep
    bad = width_sweep.rows_sweep(seed=1, verbose=False)
    assert not bad, bad


def test_sequence_lengths_through_every_counter():
    """tools/length_sweep.py: every length 0 .. 300, the neighbours of the counters' chunk and tile boundaries, random
    lengths up to 100 000 — 27 (alphabet, k) pairs from 1 to 262 144 columns, integer counts and per-kb float32 rows
    bit-exact against the C oracle (kmer_counts.py:140-151)."""
    import length_sweep
    bad = length_sweep.sweep(quick=False, seed=1, verbose=False)
    assert not bad, bad
