"""Device-side versions of what the reference does with the Pearson matrix right after computing
it: keeping these reductions on the GPU avoids shipping an N x N matrix over PCIe.

    threshold_zero_diag   kmer_leiden.py:94-96   ld_sim[ld_sim < cutoff] = 0; fill_diagonal(0)
    triu_values           find_dist.py:163       sim[np.triu_indices(N, k=1)]
    subsample             find_dist.py:169       np.random.choice(values, size, replace=False)
    empirical_pvalues     find_pval.py:158-164   p[i,j] = np.sum(fitres > sim[i,j]) / len(fitres)

Every function takes and returns device matrices (`seekr_amd._lib.Matrix`); `*_host` helpers
wrap host arrays for drop-in use.
"""
import ctypes as C

import numpy as np

from seekr_amd import _lib


def threshold_zero_diag(r, cutoff, diag_col0=0):
    """In place on the device matrix `r`; returns `r`."""
    _lib.check(_lib.lib().skr_threshold_zero_diag(r.ctx._h, r._h, C.c_float(cutoff), int(diag_col0)))
    return r


def triu_values(r, k=1):
    """1 x (n-k)(n-k+1)/2 device vector in np.triu_indices order."""
    n = r.rows
    m = max(0, n - k)
    out = r.ctx.empty(1, m * (m + 1) // 2)
    _lib.check(_lib.lib().skr_triu_flatten(r.ctx._h, r._h, int(k), out._h))
    return out


def subsample(values, size, rng=None):
    """np.random.choice(values, size, replace=False) with the draw made on the host by numpy's
    generator (legacy global state by default, as the reference uses) and the gather on device."""
    n = values.rows * values.cols
    perm = (np.random if rng is None else rng).permutation(n)[:size]
    idx = np.ascontiguousarray(perm, dtype=np.int64)
    out = np.empty(len(idx), dtype=np.float32)
    _lib.check(_lib.lib().skr_gather_f32(values.ctx._h, values._h, idx.ctypes.data_as(C.c_void_p), len(idx),
                                         out.ctypes.data_as(C.c_void_p)))
    return out


def empirical_pvalues(r, fitres):
    """Device matrix of p-values for the device matrix `r` against the 1-D background `fitres`."""
    fitres = np.asarray(fitres, dtype=np.float32).reshape(-1)
    bg = np.sort(fitres[~np.isnan(fitres)])
    ctx = r.ctx
    if len(bg) == 0:  # nothing compares greater: every count is 0
        return ctx.zeros(r.rows, r.cols)
    p = ctx.empty(r.rows, r.cols)
    d_bg = ctx.from_numpy(bg)
    _lib.check(_lib.lib().skr_empirical_pvalues(ctx._h, r._h, d_bg._h, int(len(fitres)), p._h))
    return p


def empirical_pvalues_host(sim, fitres):
    """find_pval's numpy-array branch for a host float32 `sim`; returns a host float32 matrix."""
    sim = np.ascontiguousarray(sim, dtype=np.float32)
    ctx = _lib.default_context()
    return empirical_pvalues(ctx.from_numpy(sim), fitres).to_numpy()
