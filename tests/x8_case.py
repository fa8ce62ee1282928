"""Input of the f16f8-across-ranks tests (tests/test_gpu_multirank_mock.py, tests/mock_rank_worker.py): gaussian rows
(every cell a different value: the H / X layout is kept), and — `route` — the first `n_few` rows few-valued (integers 0..3:
neighbouring cells repeat each other, the fill routes them back to the three-product split)."""
import numpy as np


def x8_matrix(n_total, cols, route, n_few):
    rng = np.random.default_rng(77)
    x = rng.standard_normal((n_total, cols)).astype(np.float32)
    if route:
        x[:n_few] = rng.integers(0, 4, size=(n_few, cols)).astype(np.float32)
    return x
