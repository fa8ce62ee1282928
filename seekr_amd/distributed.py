"""Row-sharded pipeline over the GPUs of one node (one process per GPU, RCCL over xGMI).

Transcripts are split into contiguous row ranges, one per rank, so input order is preserved.
Counting and the elementwise steps are embarrassingly parallel.  Three steps exchange data:

  * column statistics (kmer_counts.py:168,174): numpy adds rows in index order in float32, so
    the running sums are *passed along the ranks* (rank g continues where g-1 stopped) and the
    finished vector is sent back to everyone — bit-identical to the single-GPU result;
  * the global minimum of Log2.post (kmer_counts.py:208): NaN-propagating all-reduce(min);
  * Pearson (pearson.py:41): every rank needs every other rank's standardised rows.  They are
    exchanged as *prepared operands* (already split into 16-bit halves) in P-1 pairwise shifts
    (xGMI is point-to-point: each shift uses one link per direction); the GEMM on the shard
    received in shift s overlaps shift s+1.

The orchestration is written against two small interfaces so the same code runs on the HIP
engine with RCCL (production) and on numpy with gloo (CPU tests of the sharding logic):

  engine: colsum / finish / min_nan / apply / prepare / view / gemm
  comm  : rank, size, send_vec, recv_vec, allreduce(values, op), shift(...) -> ticket, wait
"""
import numpy as np

from seekr_amd import _lib


def shard_bounds(n_rows, size):
    """Contiguous, near-equal row ranges; rank g owns [b[g], b[g+1])."""
    base, extra = divmod(n_rows, size)
    bounds = [0]
    for g in range(size):
        bounds.append(bounds[-1] + base + (1 if g < extra else 0))
    return bounds


# ------------------------------------------------------------------------------ engines ----
class HipEngine:
    """The production engine: every op is a kernel launch through libseekr_hip."""

    def __init__(self, ctx, precision=_lib.PREC_BF16X3, use_symmetry=True):
        self.ctx = ctx
        self.precision = precision
        self.use_symmetry = use_symmetry  # self-blocks compute one triangle and mirror it

    def zeros_vec(self, n):
        return self.ctx.zeros(1, n)

    def cols(self, x):
        return x.cols

    def rows(self, x):
        return x.rows

    def view(self, x, row0, nrows):
        return x.view(row0, nrows)

    def colsum(self, x, acc, center=None, center2=None, square=False):
        _lib.colsum_seq(self.ctx, x, acc, center, center2, square)

    def finish(self, v, n, take_sqrt=False):
        _lib.vec_finish(self.ctx, v, n, take_sqrt)

    def min_nan(self, x, center, scale):
        return _lib.min_nan(self.ctx, x, center, scale)

    def apply(self, x, center, scale, post, shift):
        return _lib.apply(self.ctx, x, center=center, scale=scale, post=post, shift=shift, want_nan=scale is not None)[1]

    def empty_operand(self, rows, cols):
        return _lib.Operand(self.ctx, rows, cols, self.precision)

    def prepare(self, x, center=None, scale=None, post=False, shift=0.0, keep_counts=True, op=None):
        """Normalisation tail + row standardisation + operand layout in one pass over `x`;
        the normalised counts overwrite `x` when keep_counts.  Returns (operand, has_nan)."""
        return _lib.operand_fill(self.ctx, x, op=op, precision=self.precision, center=center, scale=scale, post=post,
                                 shift=shift, y=x if keep_counts else None, row_standardize=True,
                                 want_nan=scale is not None)

    def gemm(self, a, b, r, col0, symmetric=False):
        _lib.pearson_gemm_op(self.ctx, a, b, r, symmetric and self.use_symmetry, 0, col0)


class RcclComm:
    def __init__(self, ctx, rank, size):
        self.ctx, self.rank, self.size = ctx, rank, size

    def send_vec(self, v, dst):
        _lib.comm_sendrecv(self.ctx, v, 0, 1, dst, None, 0, 0, -1)

    def recv_vec(self, v, src):
        t = _lib.comm_sendrecv(self.ctx, None, 0, 0, -1, v, 0, 1, src)
        _lib.comm_wait(self.ctx, t)

    def allreduce(self, values, op):
        return _lib.comm_allreduce(self.ctx, list(values), op)

    def shift(self, send, dst, recv, recv_rows, src):
        """send: an Operand (all rows go to dst); recv: an Operand buffer (rows [0, recv_rows) filled from src)."""
        return _lib.comm_sendrecv(self.ctx, send.as_matrix(), 0, send.rows, dst, recv.as_matrix(), 0, recv_rows, src)

    def wait(self, ticket):
        _lib.comm_wait(self.ctx, ticket)

    def barrier(self):
        _lib.comm_barrier(self.ctx)


class SingleComm:
    """Degenerate communicator for one GPU."""
    rank, size = 0, 1

    def allreduce(self, values, op):
        return list(values)

    def barrier(self):
        pass


# ------------------------------------------------------------------------------ steps -------
def _chain_colsum(engine, comm, x, n_cols, center=None, center2=None, square=False):
    """Sequential float32 column sums over ALL ranks' rows in global row order; the finished
    sums end on every rank."""
    acc = engine.zeros_vec(n_cols)
    if comm.size > 1 and comm.rank > 0:
        comm.recv_vec(acc, comm.rank - 1)
    engine.colsum(x, acc, center, center2, square)
    if comm.size > 1:
        last = comm.size - 1
        if comm.rank < last:
            comm.send_vec(acc, comm.rank + 1)
            comm.recv_vec(acc, last)
        else:
            for g in range(last):
                comm.send_vec(acc, g)
    return acc


def sharded_stats(engine, comm, x, n_total, log2="Log2.post", mean=True, std=True):
    """Everything of kmer_counts.py:203-209 that needs all rows: column mean (:168), the std of
    the centred matrix (:174) and the Log2.post shift |min| (:208).  `x` (this rank's raw shard)
    is only read.  `mean` / `std`: True = compute over all ranks, False = skip, else an engine
    vector to use.  Returns (center, scale, post, shift) for the elementwise tail."""
    n_cols = engine.cols(x)
    center = None
    if mean is True:
        center = _chain_colsum(engine, comm, x, n_cols)
        engine.finish(center, n_total)
    elif mean is not False:
        center = mean
    scale = None
    if std is True:
        mprime = _chain_colsum(engine, comm, x, n_cols, center=center)
        engine.finish(mprime, n_total)
        scale = _chain_colsum(engine, comm, x, n_cols, center=center, center2=mprime, square=True)
        engine.finish(scale, n_total, take_sqrt=True)
    elif std is not False:
        scale = std
    shift = 0.0
    post = log2 == "Log2.post"
    if post:
        local_min, local_nan = engine.min_nan(x, center, scale)
        if comm.size > 1:
            flag = comm.allreduce([1.0 if local_nan else 0.0], "max")[0]
            gmin = comm.allreduce([float(local_min) if not local_nan else 0.0], "min")[0]
            local_min = np.float32(np.nan) if flag else np.float32(gmin)
        shift = float(np.abs(local_min))  # NaN stays NaN (np.abs(np.min(...)), :208)
    return center, scale, post, shift


def _any_rank(comm, flag):
    if comm.size > 1:
        return bool(comm.allreduce([1.0 if flag else 0.0], "max")[0])
    return bool(flag)


def sharded_normalize(engine, comm, x, n_total, log2="Log2.post", mean=True, std=True):
    """kmer_counts.py:203-209 on a row shard `x` of a matrix with `n_total` rows, in place.
    (Log2.pre is fused into counting and therefore not handled here.)
    Returns (mean_vec, std_vec, has_nan)."""
    center, scale, post, shift = sharded_stats(engine, comm, x, n_total, log2, mean, std)
    has_nan = False
    if center is not None or scale is not None or post:
        has_nan = engine.apply(x, center, scale, post, shift)
    return center, scale, _any_rank(comm, has_nan)


def sharded_normalize_prepare(engine, comm, x, n_total, log2="Log2.post", mean=True, std=True, keep_counts=True,
                              op=None):
    """sharded_normalize fused with the Pearson preparation: one pass writes the normalised
    counts (into `x`, when keep_counts) and the row-standardised operand.
    Returns (mean_vec, std_vec, has_nan, operand)."""
    center, scale, post, shift = sharded_stats(engine, comm, x, n_total, log2, mean, std)
    operand, has_nan = engine.prepare(x, center, scale, post, shift, keep_counts=keep_counts, op=op)
    return center, scale, _any_rank(comm, has_nan), operand


def sharded_pearson_rowblock(engine, comm, z, bounds, r, recv_bufs):
    """Row block r[n_g, N] = z_g . Z^T / K of the self-comparison, Z = all ranks' rows.

    `z`: this rank's prepared operand; `bounds`: shard_bounds(N, size); `recv_bufs`: two operand
    buffers with at least max-shard rows (double buffer).  Shift s sends our shard to rank+s and
    receives rank-s's; its GEMM overlaps shift s+1."""
    size, rank = comm.size, comm.rank
    tickets = {}
    if size > 1:
        src = (rank - 1) % size
        tickets[1] = comm.shift(z, (rank + 1) % size, recv_bufs[1 % 2], bounds[src + 1] - bounds[src], src)
    engine.gemm(z, z, r, bounds[rank], symmetric=True)
    for s in range(1, size):
        src = (rank - s) % size
        comm.wait(tickets.pop(s))
        if s + 1 < size:
            nsrc = (rank - s - 1) % size
            tickets[s + 1] = comm.shift(z, (rank + s + 1) % size, recv_bufs[(s + 1) % 2],
                                        bounds[nsrc + 1] - bounds[nsrc], nsrc)
        engine.gemm(z, engine.view(recv_bufs[s % 2], 0, bounds[src + 1] - bounds[src]), r, bounds[src])
    return r
