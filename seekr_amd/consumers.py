"""Device-side versions of what the reference does with the Pearson matrix right after computing
it: keeping these reductions on the GPU avoids shipping an N x N matrix over PCIe.

    threshold_zero_diag   kmer_leiden.py:94-96   ld_sim[ld_sim < cutoff] = 0; fill_diagonal(0)
    triu_values           find_dist.py:163       sim[np.triu_indices(N, k=1)]
    subsample             find_dist.py:169       np.random.choice(values, size, replace=False)
    empirical_pvalues     find_pval.py:158-164   p[i,j] = np.sum(fitres > sim[i,j]) / len(fitres)
    edges / pearson_edges kmer_leiden.py:91-96   the thresholded matrix as an edge list, r produced and consumed
                                                 one row stripe at a time (N x N never exists)

Every function takes and returns device matrices (`seekr_amd._lib.Matrix`); `*_host` helpers
wrap host arrays for drop-in use.
"""
import ctypes as C
import os

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


def parametric_pvalues(r, dist_name, params):
    """Device matrix p = float32(1 - scipy.stats.<dist_name>(*params).cdf(r)) — find_pval's branch for a fitted
    distribution (find_pval.py:118-133); `params` is the tuple find_dist returns (shapes..., loc, scale).
    NotImplementedError for distributions outside find_dist's common10 list."""
    ctx = r.ctx
    vals = [float(v) for v in params]
    arr = (C.c_double * max(1, len(vals)))(*vals)
    p = ctx.empty(r.rows, r.cols)
    _lib.check(_lib.lib().skr_parametric_pvalues(ctx._h, r._h, dist_name.encode(), arr, len(vals), p._h))
    return p


def pvalues_host(sim, fitres, bestfit=1):
    """find_pval's p-value step for a host float32 `sim`: `fitres` is find_dist's output — a list of
    (distribution name, deviance, parameters) (the `bestfit`-th entry is used, find_pval.py:116-121) or a 1-D array of
    background similarities (:158-164).  Returns a host float32 matrix."""
    sim = np.ascontiguousarray(sim, dtype=np.float32)
    if isinstance(fitres, np.ndarray):
        return empirical_pvalues_host(sim, fitres)
    name, _, params = fitres[bestfit - 1]
    ctx = _lib.default_context()
    return parametric_pvalues(ctx.from_numpy(sim), name, params).to_numpy()


def empirical_pvalues_host(sim, fitres):
    """find_pval's numpy-array branch for a host float32 `sim`; returns a host float32 matrix."""
    sim = np.ascontiguousarray(sim, dtype=np.float32)
    ctx = _lib.default_context()
    return empirical_pvalues(ctx.from_numpy(sim), fitres).to_numpy()


def edges(r, cutoff, nrows=None, col_begin=0, col_end=None, row_global0=0, col_global0=0, upper_only=False):
    """(rows, cols, vals) numpy arrays (uint32, uint32, float32): the cells of the device block
    r[0:nrows, col_begin:col_end] that kmer_leiden.py:94-96 leaves non-zero, in np.nonzero order."""
    ctx = r.ctx
    nrows = r.rows if nrows is None else nrows
    col_end = r.cols if col_end is None else col_end
    count = C.c_int64(0)
    args = (ctx._h, r._h, int(nrows), int(col_begin), int(col_end), int(row_global0), int(col_global0),
            C.c_float(cutoff), 1 if upper_only else 0)
    _lib.check(_lib.lib().skr_edges(*args, None, None, None, C.byref(count)))
    n = count.value
    if n == 0:
        return np.empty(0, np.uint32), np.empty(0, np.uint32), np.empty(0, np.float32)
    out_r, out_c, out_v = ctx.empty(1, n, np.uint32), ctx.empty(1, n, np.uint32), ctx.empty(1, n, np.float32)
    _lib.check(_lib.lib().skr_edges(*args, out_r._h, out_c._h, out_v._h, C.byref(count)))
    res = out_r.to_numpy().reshape(-1), out_c.to_numpy().reshape(-1), out_v.to_numpy().reshape(-1)
    for m in (out_r, out_c, out_v):
        m.free()
    return res


class FusedEdges:
    """Edge lists straight from the contraction (skr_pearson_gemm_edges): r is never written.  Keeps the output buffers
    and, for rows of more than 4 096 columns, the scratch block between calls; when a block holds more edges than the
    buffers the call is repeated with larger ones."""

    def __init__(self, ctx, capacity=1 << 20):
        self.ctx, self.cap = ctx, int(capacity)
        self._out = None
        self._scratch = None

    def _buffers(self):
        if self._out is None or self._out[0].rows < self.cap:
            if self._out is not None:
                for m in self._out:
                    m.free()
            # cap x 1: a prefix of the list is then a row range, which is what skr_mat_download copies
            self._out = (self.ctx.empty(self.cap, 1, np.uint32), self.ctx.empty(self.cap, 1, np.uint32),
                         self.ctx.empty(self.cap, 1, np.float32))
        return self._out

    def block(self, a, b, cutoff, row_global0=0, col_global0=0, upper_only=False, scratch=None, retry=True):
        """(rows, cols, vals) of the block a b^T / K, global indices, np.nonzero order.  `scratch`: a float32 matrix of at
        least [a.rows, b.rows] for rows of more than 4 096 columns (allocated and kept here when not given).  The list
        buffers start at 2e-3 entries per cell; a block with more edges is run again with room (retry=True) or reported
        as None (retry=False: the caller has a cheaper way for lists that dense)."""
        ctx = self.ctx
        # more than one accumulator restart per row?  the library's own rule, knob included (not re-derived here)
        needs = C.c_int(0)
        _lib.check(_lib.lib().skr_pearson_gemm_edges_needs_scratch(ctx._h, a._h, b._h, C.byref(needs)))
        needs = bool(needs.value)
        if needs and scratch is None:
            if self._scratch is None or self._scratch.rows < a.rows or self._scratch.cols < b.rows:
                if self._scratch is not None:
                    self._scratch.free()
                self._scratch = ctx.empty(a.rows, b.rows)
            scratch = self._scratch
        self.cap = max(self.cap, min(int(2e-3 * a.rows * b.rows) + 1024, 1 << 30))
        while True:
            o_r, o_c, o_v = self._buffers()
            count = C.c_int64(0)
            _lib.check(_lib.lib().skr_pearson_gemm_edges(ctx._h, a._h, b._h, _lib._h(scratch) if needs else None,
                                                         int(row_global0), int(col_global0), C.c_float(cutoff),
                                                         1 if upper_only else 0, o_r._h, o_c._h, o_v._h, C.byref(count)))
            n = count.value
            if n <= self.cap:
                break
            if not retry:
                return None
            self.cap = int(n * 1.25) + 1024  # more edges than the buffers hold: once more with room
        if n == 0:
            return np.empty(0, np.uint32), np.empty(0, np.uint32), np.empty(0, np.float32)
        return tuple(m.to_numpy(0, n).reshape(-1) for m in (o_r, o_c, o_v))  # only the filled prefix crosses PCIe

    def free(self):
        for m in (self._out or ()):
            m.free()
        if self._scratch is not None:
            self._scratch.free()
        self._out = self._scratch = None


def topk_rows(r, k, nrows=None, col_begin=0, col_end=None, row_global0=0, col_global0=0):
    """(idx uint32 [nrows, k], val float32 [nrows, k]): the k largest cells of each row of the device
    block, descending, diagonal cell excluded — np.argsort(-row, kind="stable")[:k] per row."""
    ctx = r.ctx
    nrows = r.rows if nrows is None else nrows
    col_end = r.cols if col_end is None else col_end
    idx, val = ctx.empty(nrows, k, np.uint32), ctx.empty(nrows, k, np.float32)
    _lib.check(_lib.lib().skr_topk_rows(ctx._h, r._h, int(nrows), int(col_begin), int(col_end), int(row_global0),
                                        int(col_global0), int(k), idx._h, val._h))
    out = idx.to_numpy(), val.to_numpy()
    idx.free()
    val.free()
    return out


FUSE_MAX_DENSITY = 1e-3  # edges per cell above which the fused epilogue costs more than writing the stripe (measured: 100 000 rows, k = 6)


def pearson_edges(z, cutoff, stripe_rows=8192, upper_only=True, engine_gemm=None, fuse="auto"):
    """Edge list of the self-comparison of the prepared operand `z` (seekr_amd._lib.Operand):
    r is produced one stripe of `stripe_rows` rows at a time into one reusable buffer (columns at or
    right of the stripe when `upper_only`) and reduced to edges before the next stripe overwrites
    it, so memory is stripe_rows x N floats instead of N x N (config 5: 32 GB instead of 4 TB).
    Returns (rows, cols, vals) in row-major order: what np.nonzero / indexing of the thresholded,
    zero-diagonal matrix (kmer_leiden.py:94-96) gives — its upper triangle when `upper_only`.
    `fuse`: True — the threshold runs inside the contraction's epilogue (FusedEdges) and no stripe of r is ever
    written; False — the two-step path (contraction into a stripe buffer, then skr_edges); "auto" (default) — fused as
    long as the stripes seen so far hold fewer than FUSE_MAX_DENSITY edges per cell (a sparse list: 15 % faster at
    1e-5 edges per cell; a dense one — the reference's default cutoff 0 keeps half the cells — is cheaper through the
    stripe buffer).  The result is the same, bit for bit, whichever path a stripe takes."""
    ctx = z.ctx
    n = z.rows
    stripe_rows = max(1, min(int(stripe_rows), n))
    fused = FusedEdges(ctx) if z.kind != 0 and fuse else None  # float32-layout operands: two-step path
    buf = None
    out = ([], [], [])
    seen_cells = seen_edges = 0
    for s0 in range(0, n, stripe_rows):
        rows = min(stripe_rows, n - s0)
        c0 = s0 if upper_only else 0
        a = z.view(s0, rows)
        b = z.view(c0, n - c0) if c0 else z
        use_fused = fused is not None and (fuse is True or seen_edges <= FUSE_MAX_DENSITY * max(seen_cells, 1))
        part = None
        if use_fused:
            if z.cols > 1024 and buf is None:
                buf = ctx.empty(stripe_rows, n)  # k = 7: the earlier k chunks need a block to leave their sums in
            part = fused.block(a, b, cutoff, row_global0=s0, col_global0=c0, upper_only=upper_only, scratch=buf,
                               retry=fuse is True)
            if part is None:  # denser than the list buffers: through the stripe buffer from here on
                fused.free()
                fused = None
        if part is None:
            if buf is None:
                buf = ctx.empty(stripe_rows, n)
            _lib.pearson_gemm_op(ctx, a, b, buf, symmetric=False, row0=0, col0=c0)
            part = edges(buf, cutoff, nrows=rows, col_begin=c0, col_end=n, row_global0=s0, col_global0=0,
                         upper_only=upper_only)
        seen_cells += rows * (n - c0)
        seen_edges += len(part[2])
        for acc, p in zip(out, part):
            acc.append(p)
    if fused:
        fused.free()
    if buf is not None:
        buf.free()
    return tuple(np.concatenate(p) if p else np.empty(0) for p in out)
