"""Consumers of the Pearson matrix against their numpy definitions in the
reference: kmer_leiden.py:94-96, find_dist.py:163,169, find_pval.py:158-164.  Bit-exact: they
are copies, comparisons and one correctly rounded division."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from seekr_amd import _lib
    return _lib.default_context()


def rand_r(n, m=None, seed=0):
    rng = np.random.default_rng(seed)
    r = np.clip(rng.normal(0.0, 0.12, size=(n, m or n)), -1, 1).astype(np.float32)
    return r


def test_threshold_zero_diag(ctx):
    from seekr_amd import consumers
    r = rand_r(301, seed=1)
    r[5, 7] = np.nan
    want = r.copy()
    with np.errstate(invalid="ignore"):
        want[want < 0.1] = 0       # kmer_leiden.py:94
    np.fill_diagonal(want, 0)      # :96
    d = ctx.from_numpy(r)
    consumers.threshold_zero_diag(d, 0.1)
    got = d.to_numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(want))
    # a row block [100:180, :] of the same matrix: the diagonal sits at column row + 100
    blk = ctx.from_numpy(r[100:180])
    consumers.threshold_zero_diag(blk, 0.1, diag_col0=100)
    assert np.array_equal(np.nan_to_num(blk.to_numpy()), np.nan_to_num(want[100:180]))


@pytest.mark.parametrize("n,k", [(1, 1), (2, 1), (7, 1), (300, 1), (1025, 1), (64, 0), (64, 3)])
def test_triu_values_and_subsample(n, k, ctx):
    from seekr_amd import consumers
    r = rand_r(n, seed=n)
    want = r[np.triu_indices(n, k=k)]  # find_dist.py:163
    d = ctx.from_numpy(r)
    flat = consumers.triu_values(d, k=k)
    assert flat.cols == len(want)
    if len(want):
        assert np.array_equal(flat.to_numpy().reshape(-1), want)
    if len(want) > 10:
        np.random.seed(42)
        ref = np.random.choice(want, size=10, replace=False)  # find_dist.py:169
        np.random.seed(42)
        got = consumers.subsample(flat, 10)
        assert np.array_equal(got, ref)


def test_empirical_pvalues(ctx):
    from seekr_amd import consumers
    sim = rand_r(40, 55, seed=3)
    sim[0, 0] = np.nan
    fitres = rand_r(1, 5000, seed=4).reshape(-1)
    fitres[:55] = sim[1]  # ties with sim values
    fitres[13] = np.nan
    want = np.zeros_like(sim)  # find_pval.py:154-164
    with np.errstate(invalid="ignore"):
        for i in range(sim.shape[0]):
            for j in range(sim.shape[1]):
                want[i, j] = np.sum(fitres > sim[i, j]) / len(fitres)
    got = consumers.empirical_pvalues_host(sim, fitres)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    assert np.array_equal(consumers.empirical_pvalues_host(sim, np.array([np.nan, np.nan])), np.zeros_like(sim))
