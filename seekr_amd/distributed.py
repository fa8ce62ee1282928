"""Row-sharded pipeline over the GPUs of one node (one process per GPU, RCCL over xGMI).

Transcripts are split into contiguous row ranges, one per rank, so input order is preserved.
Counting and the elementwise steps are embarrassingly parallel.  Three steps exchange data:

  * column statistics (kmer_counts.py:168,174): numpy adds rows in index order in float32, so
    the running sums are *passed along the ranks* (rank g continues where g-1 stopped) and the
    finished vector is sent back to everyone — bit-identical to the single-GPU result;
  * the global minimum of Log2.post (kmer_counts.py:208): NaN-propagating all-reduce(min);
  * Pearson (pearson.py:41): standardised rows are exchanged as *prepared operands* (already
    split into 16-bit halves) in pairwise shifts (xGMI is point-to-point: each shift uses one
    link per direction); the GEMM on the shard received in shift s overlaps shift s+1.
    Two layouts of the result:
      - sharded_pearson_symmetric (default): r(h,g) = r(g,h)^T, so every unordered pair of
        shards is multiplied ONCE, by one of its two ranks, which stores the block and its
        transpose.  floor(P/2) shifts; each rank multiplies P/2 shard-blocks (1/2 for its own
        triangle) and ends up holding n_g*N of the N^2 ordered pairs, each pair on exactly one
        rank: its diagonal block, the blocks (g,h) of its forward half-ring and their mirrors
        (h,g).  Work per delivered pair is the same as on one GPU.
      - sharded_pearson_rowblock: rank g holds rows bounds[g]:bounds[g+1] of r, all columns —
        the layout np.save expects — at the price of multiplying every off-diagonal block
        twice across the node (P-1 shifts, P-1/2 shard-blocks per rank).

The orchestration is written against two small interfaces so the same code runs on the HIP
engine with RCCL (production) and on numpy with gloo (CPU tests of the sharding logic):

  engine: colsum / finish / min_nan / apply / prepare / view / gemm / gemm_mirror
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

    def __init__(self, ctx, precision=_lib.PREC_F16X3, use_symmetry=True, row_standardize=True):
        self.ctx = ctx
        self.precision = precision
        self.use_symmetry = use_symmetry  # self-blocks compute one triangle and mirror it
        self.row_standardize = row_standardize  # pearson(..., row_standardize=False): the operands hold the rows as they are

    def zeros_vec(self, n):
        return self.ctx.zeros(1, n)

    def cols(self, x):
        return x.cols

    def rows(self, x):
        return x.rows

    def view(self, x, row0, nrows):
        return x.view(row0, nrows)

    def colsum(self, x, acc, center=None, center2=None, square=False, colmin=None):
        if colmin is not None:
            _lib.colsum_seq_colmin(self.ctx, x, acc, colmin)
        else:
            _lib.colsum_seq(self.ctx, x, acc, center, center2, square)

    def colsum_chain(self, chain, x, acc, center=None, center2=None, square=False, colmin=None):
        """One pass of the column sums as a link of the peer-mailbox chain (skr_colsum_seq_chain): on return `acc` holds
        the sums over all ranks' rows."""
        chain.colsum(x, acc, center, center2, square, colmin)

    def colmin_buffer(self, x):
        """A [4, cols] buffer for the raw column minima the first column-sum pass can produce on the way (None for a
        shard without rows: then the Log2.post minimum is scanned from the — empty — matrix).  Any width since round 5."""
        return self.ctx.empty(4, x.cols) if x.rows > 0 else None

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
                                 shift=shift, y=x if keep_counts else None, row_standardize=self.row_standardize,
                                 want_nan=scale is not None)

    def layout(self, op):
        """Storage kind of a prepared operand (0 = float32 fallback, see _lib.Operand.kind)."""
        return op.kind

    def adopt_layout(self, buf, like):
        if buf is not None:
            buf.adopt_layout(like)

    def prepare_f32(self, x, op=None):
        """The rows of `x` (already normalised) as a float32-layout operand: what every rank switches
        to when any rank's rows need the fp32 kernel's dynamic range."""
        op = _lib.Operand(self.ctx, x.rows, x.cols, _lib.PREC_FP32) if op is None or op.kind != 0 else op
        return _lib.operand_fill(self.ctx, x, op=op, precision=_lib.PREC_FP32, row_standardize=self.row_standardize)[0]

    def prepare_f16x3(self, x, op=None):
        """The rows of `x` (already normalised) as a split-fp16 (three-product) operand in the storage of `op`: what every
        rank switches to when any rank's opt-in f16f8 operand routed itself back."""
        like = _lib.Operand(self.ctx, 1, x.cols, _lib.PREC_F16X3)
        op = _lib.Operand(self.ctx, x.rows, x.cols, _lib.PREC_F16X3) if op is None else op.adopt_layout(like)
        like.free()
        return _lib.operand_fill(self.ctx, x, op=op, precision=_lib.PREC_F16X3, row_standardize=self.row_standardize)[0]

    def gemm(self, a, b, r, col0, symmetric=False, lower=False):
        """lower: `b` holds rows that come BEFORE a's in the matrix both belong to — the block below the diagonal of a
        self-comparison, asked for with the bits its mirror would have (skr_pearson_gemm_op, symmetric = 2)."""
        _lib.pearson_gemm_op(self.ctx, a, b, r, symmetric and self.use_symmetry, 0, col0, lower=lower)

    def gemm_rows(self, a, full, a_row0, r):
        """r[i, :] = rows a_row0 + i of the self-comparison of `full`, with the one-block call's bits (skr_pearson_gemm_op_rows)."""
        _lib.pearson_gemm_op_rows(self.ctx, a, full, a_row0, r)

    def gemm_mirror(self, a, b, r, row0, col0, rt, trow0, tcol0):
        """r[row0+i, col0+j] = <a_i, b_j>/K and rt[trow0+j, tcol0+i] = the same value."""
        _lib.pearson_gemm_op_mirror(self.ctx, a, b, r, row0, col0, rt, trow0, tcol0)

    def empty_block(self, rows, cols):
        return self.ctx.empty(rows, cols)

    def fused_edges(self, operand):
        """A consumers.FusedEdges for split-precision operands (threshold inside the contraction's epilogue: no stripe of
        r is written), None for float32-layout ones (two-step path)."""
        from seekr_amd import consumers
        return consumers.FusedEdges(self.ctx) if operand.kind != 0 else None

    def edges(self, r, cutoff, nrows, col_begin, col_end, row_global0, upper_only):
        from seekr_amd import consumers
        return consumers.edges(r, cutoff, nrows=nrows, col_begin=col_begin, col_end=col_end, row_global0=row_global0,
                               upper_only=upper_only)


class RcclComm:
    SEND_TICKET_OPTIONAL = True  # send_vec takes want_ticket (a communicator without the attribute is called without it)

    def __init__(self, ctx, rank, size):
        self.ctx, self.rank, self.size = ctx, rank, size
        self._chain = None        # None: not tried yet; False: unavailable (the vectors travel by send/recv); else _lib.Chain
        self._chain_note = ""

    def chain_for(self, engine, n_cols):
        """The peer-mailbox chain for the column sums (skr_chain), OPT-IN (SEEKR_CHAIN=mailbox; the default is the
        send/recv chain, `rccl`, until the in-kernel wait has run between two real GPUs: ADVICE r3).  Set up
        collectively on first use: every rank exports its mailbox, the 64-byte IPC handles are all-gathered, every rank
        opens the others', and one small chained pass is checked against the send/recv chain.  Any failure on any rank —
        IPC refused, a store that never becomes visible (the waits are bounded) — and ALL ranks stay with send/recv.
        Every collective of the set-up (all-gather, the send/recv reference pass, the two verdict all-reduces) is entered
        by every rank whatever happened to it before: errors are recorded locally and compared at the end."""
        import os
        if self._chain is not None and (self._chain is False or self._chain.cols_cap >= n_cols):
            return self._chain or None
        want = os.environ.get("SEEKR_CHAIN", "rccl")
        if self.size < 2 or self.size > 16 or want != "mailbox":
            self._chain, self._chain_note = False, "not requested"
            return None
        ctx = self.ctx
        if self._chain:
            self._chain.free()
        chain, why = None, ""
        try:
            chain = _lib.Chain(ctx, max(int(n_cols), 16384))
            mine = np.frombuffer(chain.export(), dtype=np.float32).reshape(1, 16)
        except Exception as e:  # noqa: BLE001
            why, mine = "export: {}".format(e), np.zeros((1, 16), np.float32)
        # the all-gather is collective: every rank takes part, whatever happened to it so far
        shard, full = ctx.from_numpy(mine), ctx.zeros(self.size, 16)
        self.wait(_lib.comm_allgather_rows(ctx, shard, full, list(range(self.size + 1))))
        ctx.sync()
        handles = [full.to_numpy(g, 1).tobytes() for g in range(self.size)]
        if not why:
            try:
                chain.connect(self.rank, handles)
            except Exception as e:  # noqa: BLE001
                why = "connect: {}".format(e)
        bad = self.allreduce([1.0 if why else 0.0], "max")[0]
        if not bad:
            # self-test: a small chained pass must equal the same pass over send/recv, bit for bit, on every rank.  The
            # send/recv reference runs FIRST and outside the try: it is a collective, and a rank whose chained pass
            # fails must not leave the others alone in it (that was a hang, not a fallback).
            rows = 5 + self.rank
            x = ctx.from_numpy((np.arange(rows * 48, dtype=np.float32).reshape(rows, 48) % 7 + self.rank) * np.float32(0.37))
            self._chain = False  # the reference pass takes the send/recv path
            want_vec = _chain_colsum(engine, self, x, 48)
            ctx.sync()
            self._chain = None
            try:
                got = engine.zeros_vec(48)
                engine.colsum_chain(chain, x, got)
                if chain.timed_out():
                    raise _lib.SeekrHipError("a peer's store never became visible")
                if not np.array_equal(got.vector().view(np.uint32), want_vec.vector().view(np.uint32)):
                    raise _lib.SeekrHipError("the mailbox chain and the send/recv chain disagree")
            except Exception as e:  # noqa: BLE001
                why = "self-test: {}".format(e)
            bad = self.allreduce([1.0 if why else 0.0], "max")[0]
        if bad:
            if chain is not None:
                chain.free()
            self._chain, self._chain_note = False, why or "another rank could not set it up"
            return None
        self._chain, self._chain_note = chain, "peer mailboxes over HIP IPC"
        return chain

    def chain_gave_up(self):
        """True when a link of the mailbox chain on THIS rank gave up waiting since the last call (its sums are garbage).
        Synchronises the stream; _verdicts() all-reduces it so that every rank raises together."""
        return bool(self._chain) and self._chain.timed_out()

    def send_vec(self, v, dst, want_ticket=True):
        """Returns a ticket: the compute stream must wait on it before it overwrites `v`.  want_ticket=False: fire
        and forget (no event kept; later exchanges on the communication stream are still ordered behind it)."""
        return _lib.comm_sendrecv(self.ctx, v, 0, 1, dst, None, 0, 0, -1, want_ticket=want_ticket)

    def recv_vec(self, v, src):
        t = _lib.comm_sendrecv(self.ctx, None, 0, 0, -1, v, 0, 1, src)
        _lib.comm_wait(self.ctx, t)

    def allreduce(self, values, op):
        return _lib.comm_allreduce(self.ctx, list(values), op)

    def shift(self, send, dst, recv, recv_rows, src):
        """send: an Operand (all rows go to dst); recv: an Operand buffer (rows [0, recv_rows) filled from src)."""
        return _lib.comm_sendrecv(self.ctx, send.as_matrix(), 0, send.rows, dst, recv.as_matrix(), 0, recv_rows, src)

    def shift_part(self, send, srow0, snrows, dst, recv, drow0, dnrows, src):
        """A row range of a shift: rows [srow0, srow0+snrows) of `send` go to dst, rows [drow0, drow0+dnrows) of `recv`
        come from src (the sender splits ITS shard the way the receiver expects: both sides use the sender's row count)."""
        return _lib.comm_sendrecv(self.ctx, send.as_matrix(), srow0, snrows, dst, recv.as_matrix(), drow0, dnrows, src)

    def shift_all(self, shifts):
        """shifts: [(send, dst, recv, recv_rows, src)] as ONE grouped exchange: every peer's link carries traffic at
        once.  Returns one ticket for all of them."""
        return _lib.comm_exchange(self.ctx, [(s.as_matrix(), 0, s.rows, dst, r.as_matrix(), 0, rows, src)
                                             for s, dst, r, rows, src in shifts])

    def allgather_rows(self, shard, full, bounds):
        """shard: this rank's Operand (or Matrix); full: an Operand (Matrix) of bounds[-1] rows; returns a ticket."""
        as_mat = lambda m: m.as_matrix() if hasattr(m, "as_matrix") else m  # noqa: E731
        return _lib.comm_allgather_rows(self.ctx, as_mat(shard), as_mat(full), bounds)

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

    def wait(self, ticket):
        pass


# ------------------------------------------------------------------------------ steps -------
def _send_vec(comm, v, dst, want_ticket):
    # decided from a class attribute, not by catching TypeError around the call: a TypeError raised INSIDE a real send
    # must surface, not post the send a second time (an unmatched send hangs the chain)
    if getattr(comm, "SEND_TICKET_OPTIONAL", False):
        return comm.send_vec(v, dst, want_ticket=want_ticket)
    return comm.send_vec(v, dst)  # e.g. the gloo communicator of the CPU tests


def _chain_colsum(engine, comm, x, n_cols, center=None, center2=None, square=False, colmin=None):
    """Sequential float32 column sums over ALL ranks' rows in global row order; the finished
    sums end on every rank.  `colmin` (first pass only): also filled with this rank's raw column minima."""
    acc = engine.zeros_vec(n_cols)
    chain = comm.chain_for(engine, n_cols) if comm.size > 1 and hasattr(comm, "chain_for") and hasattr(engine, "colsum_chain") else None
    if chain is not None:
        # no transfer between two kernels: the kernel itself waits for the previous rank's sums and stores its own into
        # the next rank's mailbox; the finished sums come back through the result box (skr_colsum_seq_chain)
        engine.colsum_chain(chain, x, acc, center, center2, square, colmin)
        return acc
    if comm.size > 1 and comm.rank > 0:
        comm.recv_vec(acc, comm.rank - 1)
    if colmin is not None:
        engine.colsum(x, acc, colmin=colmin)
    else:
        engine.colsum(x, acc, center, center2, square)
    if comm.size > 1:
        last = comm.size - 1
        if comm.rank < last:
            # no ticket: the receive below is ordered behind this send on the communication stream
            _send_vec(comm, acc, comm.rank + 1, want_ticket=False)
            comm.recv_vec(acc, last)
        else:
            ticket = None
            for g in range(last):  # one stream, in order: the last send's ticket covers them all
                ticket = _send_vec(comm, acc, g, want_ticket=g == last - 1)
            # the caller goes on to rescale `acc` in place (finish) on the compute stream, while these
            # sends sit on the communication stream: without this wait the peers could read the
            # rescaled vector (caught by the asynchronous mode of tests/mock_rccl)
            if ticket is not None:
                comm.wait(ticket)
    return acc


def sharded_stats(engine, comm, x, n_total, log2="Log2.post", mean=True, std=True):
    """Everything of kmer_counts.py:203-209 that needs all rows: column mean (:168), the std of
    the centred matrix (:174) and the Log2.post shift |min| (:208).  `x` (this rank's raw shard)
    is only read.  `mean` / `std`: True = compute over all ranks, False = skip, else an engine
    vector to use.  Returns (center, scale, post, shift) for the elementwise tail."""
    n_cols = engine.cols(x)
    center = None
    # Log2.post needs min z over the whole matrix.  When the mean is computed here anyway, the first column-sum pass
    # brings the raw column minima along, and (rounding being monotone, the scale a computed std >= 0 or absent) the
    # minimum of z is found among the normalised column minima: 4 x cols cells instead of a pass over the matrix.
    colmin = None
    if (log2 == "Log2.post" and mean is True and (std is True or std is False or std is None)
            and hasattr(engine, "colmin_buffer")):
        colmin = engine.colmin_buffer(x)
    if mean is True:
        center = _chain_colsum(engine, comm, x, n_cols, colmin=colmin)
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
        local_min, local_nan = engine.min_nan(colmin if colmin is not None else x, center, scale)
        if comm.size > 1:
            # ONE host all-reduce (each costs a stream drain and a round trip): min over [the local minimum or +inf where
            # the shard holds a NaN, minus the NaN flag] — min(-flag) = -max(flag)
            gmin, neg_flag = comm.allreduce([float(local_min) if not local_nan else float("inf"), -1.0 if local_nan else 0.0], "min")
            local_min = np.float32(np.nan) if neg_flag < 0 else np.float32(gmin)
        shift = float(np.abs(local_min))  # NaN stays NaN (np.abs(np.min(...)), :208)
    return center, scale, post, shift


def _any_rank(comm, flag):
    if comm.size > 1:
        return bool(comm.allreduce([1.0 if flag else 0.0], "max")[0])
    return bool(flag)


def _verdicts(comm, fell_back, coherent, has_nan, routed_back=False, x8_stats=None):
    """The per-step verdicts of the normalisation as ONE host all-reduce (a stream drain and a round trip each, were they
    separate): did any rank's operand fall back to the float32 layout, is any rank's shard 'mostly one repeated value',
    did any rank see a NaN, did any rank's opt-in f16f8 operand route itself back to the three-product split — and did a
    link of the mailbox chain give up waiting on ANY rank (the sums behind such a link are garbage on every
    later rank, in every log2 mode; all ranks raise together instead of one rank raising and the others hanging in the
    next collective).  sharded_normalize and sharded_normalize_prepare both end with it, so that ranks may mix the two
    entry points (tests/dist_worker.py does)."""
    if comm.size > 1:
        gave_up = bool(getattr(comm, "chain_gave_up", lambda: False)())
        out = comm.allreduce([1.0 if fell_back else 0.0, 1.0 if coherent else 0.0, 1.0 if has_nan else 0.0,
                              1.0 if gave_up else 0.0, 1.0 if routed_back else 0.0] + list(x8_stats or (0.0, 0.0, 0.0)), "max")
        # (always eight values: ranks may mix the two entry points, and an all-reduce of different lengths never returns)
        if x8_stats is not None:  # the opt-in f16f8 layout: the three row-mean maxima of every shard, made global
            x8_stats[:] = out[5:8]
        if out[3] > 0:
            raise _lib.SeekrHipError("rank {}: a link of the column-sum chain gave up waiting for a peer's mailbox store "
                                     "({}): the column statistics of this step are invalid on every rank".format(
                                         comm.rank, "this rank" if gave_up else "another rank"))
        return out[0] > 0, out[1] > 0, out[2] > 0, out[4] > 0
    return bool(fell_back), bool(coherent), bool(has_nan), bool(routed_back)


def sharded_normalize(engine, comm, x, n_total, log2="Log2.post", mean=True, std=True):
    """kmer_counts.py:203-209 on a row shard `x` of a matrix with `n_total` rows, in place.
    (Log2.pre is fused into counting and therefore not handled here.)
    Returns (mean_vec, std_vec, has_nan)."""
    center, scale, post, shift = sharded_stats(engine, comm, x, n_total, log2, mean, std)
    has_nan = False
    if center is not None or scale is not None or post:
        has_nan = engine.apply(x, center, scale, post, shift)
    return center, scale, _verdicts(comm, False, False, has_nan)[2]


def sharded_normalize_prepare(engine, comm, x, n_total, log2="Log2.post", mean=True, std=True, keep_counts=True,
                              op=None):
    """sharded_normalize fused with the Pearson preparation: one pass writes the normalised
    counts (into `x`, when keep_counts) and the row-standardised operand.
    Returns (mean_vec, std_vec, has_nan, operand)."""
    center, scale, post, shift = sharded_stats(engine, comm, x, n_total, log2, mean, std)
    operand, has_nan = engine.prepare(x, center, scale, post, shift, keep_counts=keep_counts, op=op)
    # A rank whose rows need more dynamic range than the split contraction has (skr_operand_kind) comes
    # back with a float32-layout operand; shards are multiplied against each other, so then every rank
    # switches (rare: raw counts of homopolymer-like sequences).  Needs the normalised counts in x.
    # "rows are mostly one repeated value" (skr_operand_coherent) decides how often the contraction restarts its
    # accumulators: a block (g, h) must be treated the same whichever of its two ranks multiplies it, so that flag is
    # made global too (receive buffers adopt it from the local shard in sharded_pearson_*).  The three verdicts travel
    # in ONE host all-reduce (round 3: they were three, each a stream drain and a round trip per step).
    if comm.size > 1:
        fell_back = hasattr(engine, "layout") and engine.layout(operand) == 0 and engine.precision != _lib.PREC_FP32
        coherent = bool(getattr(operand, "coherent", False))
        # the opt-in f16f8 layout (kind 3): a rank whose fill routed its rows back to the three-product split (kind 2:
        # neighbouring cells repeat each other, or the shape has no H / X layout) makes every rank follow
        x8 = getattr(engine, "precision", None) == _lib.PREC_F16F8 and hasattr(engine, "layout")
        routed_back = x8 and engine.layout(operand) == 2
        # ... and the rule on the row means of the rounding residues (skr_operand_x8_stats) has to hold between ANY two rows
        # of the matrix, not just inside a shard: the three maxima ride on the same all-reduce, every rank applies the rule
        # to the global values and routes back together
        stats = list(operand.x8_stats) if x8 and hasattr(operand, "x8_stats") and engine.layout(operand) == 3 else [0.0, 0.0, 0.0]
        any_fell_back, any_coherent, any_nan, any_routed_back = _verdicts(comm, fell_back, coherent, has_nan, routed_back,
                                                                          stats if x8 else None)
        if x8 and hasattr(operand, "x8_pair_bound") and not any_routed_back and engine.layout(operand) == 3:
            operand.x8_stats = stats  # the global maxima; receive buffers take them over in adopt_layout
            if not operand.x8_pair_bound()[1]:  # the library's own rule on them (skr_operand_x8_pair_bound)
                any_routed_back = True
        normalised_in_x = keep_counts or (center is None and scale is None and not post)  # x still holds what was prepared
        if any_fell_back and hasattr(engine, "layout") and engine.layout(operand) != 0:
            if not normalised_in_x:
                raise NotImplementedError("a rank fell back to the float32 contraction; re-run with keep_counts=True")
            operand = engine.prepare_f32(x)
        elif x8 and any_routed_back and engine.layout(operand) == 3:
            if not normalised_in_x:
                raise NotImplementedError("a rank routed its f16f8 operand back to f16x3; re-run with keep_counts=True")
            operand = engine.prepare_f16x3(x, op=operand)
        if hasattr(operand, "coherent") and any_coherent != bool(operand.coherent) and not any_fell_back:
            operand.coherent = any_coherent
        has_nan = any_nan
    return center, scale, bool(has_nan), operand


def sharded_pearson_rowblock(engine, comm, z, bounds, r, recv_bufs):
    """Row block r[n_g, N] = z_g . Z^T / K of the self-comparison, Z = all ranks' rows.

    `z`: this rank's prepared operand; `bounds`: shard_bounds(N, size); `recv_bufs`: two operand
    buffers with at least max-shard rows (double buffer).  Shift s sends our shard to rank+s and
    receives rank-s's; its GEMM overlaps shift s+1."""
    size, rank = comm.size, comm.rank
    tickets = {}
    if hasattr(engine, "adopt_layout"):
        for buf in recv_bufs:
            engine.adopt_layout(buf, z)
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
        # shards that come before ours lie below the diagonal: asked for with their mirror's bits, so that the row block
        # is, bit for bit, rows bounds[rank]:bounds[rank+1] of the one-GPU self-comparison
        engine.gemm(z, engine.view(recv_bufs[s % 2], 0, bounds[src + 1] - bounds[src]), r, bounds[src], lower=src < rank)
    return r


# ------------------------------------------------------------------------------ half ring ---
def half_ring_plan(size, rank, bounds):
    """The cross blocks `rank` multiplies in the symmetric layout, in shift order:
    [(s, peer, a_row0, a_rows, b_row0, b_rows)] — rows a_row0.. of this rank's shard against
    rows b_row0.. of shard `peer` = rank+s.  Every unordered pair of shards {g, h} appears in
    exactly one rank's plan; for even P the pair at distance P/2 is met by both of its ranks
    and is split between them: the lower rank takes its rows against the first half of the
    other's, the higher rank the second half of its rows against all of the lower's."""
    def n(g):
        return bounds[g + 1] - bounds[g]

    plan = []
    for s in range(1, size // 2 + 1):
        peer = (rank + s) % size
        if 2 * s == size:
            half = n(max(rank, peer)) // 2
            if rank < peer:
                plan.append((s, peer, 0, n(rank), 0, half))
            else:
                plan.append((s, peer, half, n(rank) - half, 0, n(peer)))
        else:
            plan.append((s, peer, 0, n(rank), 0, n(peer)))
    return plan


def owned_blocks(size, rank, bounds):
    """Where the symmetric layout keeps which ordered pairs on `rank`:
    [(buffer, buf_row0, buf_col0, nrows, ncols, global_row0, global_col0)], buffer "row" =
    r_row [n_g, N] (global columns), "col" = r_col [N, n_g] (global rows).  Over all ranks the
    blocks tile the N x N matrix exactly once."""
    g0, n_g = bounds[rank], bounds[rank + 1] - bounds[rank]
    out = [("row", 0, g0, n_g, n_g, g0, g0)]
    for _, peer, a0, an, b0, bn in half_ring_plan(size, rank, bounds):
        if an and bn:
            p0 = bounds[peer] + b0
            out.append(("row", a0, p0, an, bn, g0 + a0, p0))
            out.append(("col", p0, a0, bn, an, p0, g0 + a0))
    return out


def sharded_pearson_symmetric(engine, comm, z, bounds, r_row, r_col, recv_bufs, grouped=False, split_first=True):
    """Self-comparison with every unordered pair of shards multiplied once (module docstring).

    `z`: this rank's prepared operand; `r_row` [n_g, N] and `r_col` [N, n_g]: float32 result
    buffers (r_col may be None on one GPU); `recv_bufs`: operand buffers of at least max-shard rows.
    With two buffers shift s sends our shard to rank-s and receives rank+s's, its GEMM overlapping shift
    s+1.  `split_first`: the FIRST shift — the only one with nothing but the small own triangle to hide behind
    (rehearsal at 200 000 rows on 8 ranks: 8.2 ms of transfer behind 2.9 ms of work) — travels as two halves, and
    the first cross block is multiplied half by half, so the wait shrinks to half a transfer.  `grouped` (needs one
    receive buffer per shift): ALL shifts are posted up front as one grouped exchange, so every peer's xGMI link
    carries traffic at once and no later shift can be exposed behind a GEMM that finished early; the first cross
    block then waits for the whole group.  Returns owned_blocks(size, rank, bounds)."""
    size, rank = comm.size, comm.rank
    plan = half_ring_plan(size, rank, bounds)
    tickets = {}
    if hasattr(engine, "adopt_layout"):
        for buf in recv_bufs:
            engine.adopt_layout(buf, z)
    can_group = hasattr(comm, "shift_all") and len(recv_bufs) >= len(plan) > 1
    grouped = bool(grouped) and can_group
    buf_of = (lambda i: recv_bufs[i]) if grouped else (lambda i: recv_bufs[i % 2])

    def n_of(g):
        return bounds[g + 1] - bounds[g]

    def post(i):
        s, peer = plan[i][0], plan[i][1]
        tickets[i] = comm.shift(z, (rank - s) % size, buf_of(i), n_of(peer), peer)

    # halves of the first shift: every rank splits ITS shard at n_g // 2, so sender and receiver agree; only when the
    # first cross block uses both shards whole (always, except the distance-P/2 pair of an even P, i.e. P = 2)
    # The decision must be the same on every rank (a rank's send is half of its left neighbour's receive), so it rests on
    # global facts only: P >= 3 (the distance-1 pair then uses both shards whole on every rank) and no shard below 2 rows.
    halves = None
    if (split_first and not grouped and size >= 3 and plan and hasattr(comm, "shift_part")
            and min(n_of(g) for g in range(size)) >= 2):
        s, peer = plan[0][0], plan[0][1]
        dst = (rank - s) % size
        own_h, peer_h = n_of(rank) // 2, n_of(peer) // 2
        halves = [(0, peer_h, comm.shift_part(z, 0, own_h, dst, buf_of(0), 0, peer_h, peer)),
                  (peer_h, n_of(peer) - peer_h,
                   comm.shift_part(z, own_h, n_of(rank) - own_h, dst, buf_of(0), peer_h, n_of(peer) - peer_h, peer))]
    def cross(a, a0, b, p0, peer):
        """The block (rows a0.. of this shard) x (rows of shard `peer` starting at global row p0) and its mirror.  The rows
        that come FIRST in the matrix take the A side of the contraction — as in the one-GPU self-comparison, whose cell
        (i, j) and its mirror both carry the value computed with row min(i, j) as A (the split contraction names A's halves
        first in the order of its cross products): every cell is then the one-GPU result bit for bit (round 5)."""
        if peer > rank:
            engine.gemm_mirror(a, b, r_row, a0, p0, r_col, p0, a0)
        else:
            engine.gemm_mirror(b, a, r_col, p0, a0, r_row, a0, p0)

    group_ticket = None
    if grouped:
        group_ticket = comm.shift_all([(z, (rank - s) % size, buf_of(i), n_of(peer), peer)
                                       for i, (s, peer, *_rest) in enumerate(plan)])
    elif plan and halves is None:
        post(0)
    engine.gemm(z, z, r_row, bounds[rank], symmetric=True)
    for i, (_, peer, a0, an, b0, bn) in enumerate(plan):
        if i == 0 and halves is not None:
            p0 = bounds[peer]
            for j, (h0, hn, ticket) in enumerate(halves):
                comm.wait(ticket)
                if j == len(halves) - 1 and len(plan) > 1:
                    post(1)
                if hn:
                    cross(z, 0, engine.view(buf_of(0), h0, hn), p0 + h0, peer)
            continue
        if grouped:
            if i == 0:
                comm.wait(group_ticket)
        else:
            comm.wait(tickets.pop(i))
            if i + 1 < len(plan):
                post(i + 1)
        if an and bn:
            a = z if an == engine.rows(z) else engine.view(z, a0, an)
            b = engine.view(buf_of(i), b0, bn)
            cross(a, a0, b, bounds[peer] + b0, peer)
    return owned_blocks(size, rank, bounds)


# ------------------------------------------------------------------------------ edge lists ---
def allgather_operand(engine, comm, z, bounds, full=None):
    """Every rank ends with the prepared rows of all ranks, in global row order: one grouped RCCL
    exchange with all P-1 peers at once (16 KiB per row at k = 6; config 5 is 65 GB in total, which
    each 288 GB GPU can hold).  Returns the full operand (`z` itself on one GPU)."""
    if comm.size == 1:
        return z
    full = engine.empty_operand(bounds[-1], engine.cols(z)) if full is None else full
    if hasattr(engine, "adopt_layout"):
        engine.adopt_layout(full, z)
    comm.wait(comm.allgather_rows(z, full, bounds))
    return full


def sharded_pearson_allgather(engine, comm, z, bounds, r, full=None):
    """Row block r[n_g, N] = z_g . Z^T / K the way the north_star names it: ONE all-gather of the prepared operands
    (every rank's shard to every rank, all xGMI links at once: skr_comm_allgather_rows), then this rank's rows against
    the gathered matrix — the own block as a symmetric product, the columns right of it as plain ones, those left of it with
    their mirror's bits — so the result is the row-block layout's, and the one-GPU self-comparison's, bit for bit.  No transfer overlaps a contraction here:
    the schedule trades the half ring's overlap and its halved work for a single collective, which is what makes it the
    most robust of the three (bench.py --layout allgather).  `full`: an operand of bounds[-1] rows to gather into (kept
    between steps by the caller).  Returns r."""
    size, rank = comm.size, comm.rank
    full = allgather_operand(engine, comm, z, bounds, full)
    lo, hi = bounds[rank], bounds[rank + 1]
    n_total = bounds[-1]
    engine.gemm(z, z, r, lo, symmetric=True)
    if lo > 0:
        engine.gemm(z, engine.view(full, 0, lo), r, 0, lower=True)  # below the diagonal: the mirror's bits (skr_pearson_gemm_op)
    if hi < n_total:
        engine.gemm(z, engine.view(full, hi, n_total - hi), r, hi)
    return r


def stripes_of_rank(n_total, stripe_rows, size, rank):
    """Row stripes [s0, s1) this rank reduces.  Stripes are dealt in zig-zag order (0..P-1, P-1..0,
    ...): with upper_only the work of a stripe shrinks linearly with its index, and the zig-zag
    keeps the ranks' totals within one stripe of each other."""
    out = []
    for t, s0 in enumerate(range(0, n_total, stripe_rows)):
        lap, pos = divmod(t, size)
        owner = pos if lap % 2 == 0 else size - 1 - pos
        if owner == rank:
            out.append((s0, min(n_total, s0 + stripe_rows)))
    return out


def sharded_pearson_edges(engine, comm, z, bounds, cutoff, stripe_rows=8192, upper_only=True, full=None, fuse="auto"):
    """The edge list kmer_leiden builds from r (kmer_leiden.py:91-96: r < cutoff -> 0, zero diagonal,
    non-zero cells are edges) for a row-sharded set, without ever holding an N x N matrix: the
    operands are all-gathered once, then every rank produces and reduces its own row stripes of r
    (stripe x columns at or right of the stripe when `upper_only`), no further communication.
    Returns this rank's (rows, cols, vals) in row-major order of its stripes; the union over the
    ranks is the edge list of the whole matrix."""
    from seekr_amd.consumers import FUSE_MAX_DENSITY
    n_total = bounds[-1]
    full = allgather_operand(engine, comm, z, bounds, full)
    stripe_rows = max(1, min(int(stripe_rows), n_total))
    fused = engine.fused_edges(full) if fuse and hasattr(engine, "fused_edges") else None
    buf = None
    out = ([], [], [])
    seen_cells = seen_edges = 0
    for s0, s1 in stripes_of_rank(n_total, stripe_rows, comm.size, comm.rank):
        c0 = s0 if upper_only else 0
        a = engine.view(full, s0, s1 - s0)
        b = engine.view(full, c0, n_total - c0) if c0 else full
        # "auto": fused while the list stays sparse (consumers.FUSE_MAX_DENSITY), else through the stripe buffer
        use_fused = fused is not None and (fuse is True or seen_edges <= FUSE_MAX_DENSITY * max(seen_cells, 1))
        part = None
        if use_fused:
            if engine.cols(full) > 1024 and buf is None:
                buf = engine.empty_block(stripe_rows, n_total)  # k = 7: the earlier k chunks leave their sums here
            part = fused.block(a, b, cutoff, row_global0=s0, col_global0=c0, upper_only=upper_only, scratch=buf,
                               retry=fuse is True)
            if part is None:  # denser than the list buffers: through the stripe buffer from here on
                fused.free()
                fused = None
        if part is None:
            if buf is None:
                buf = engine.empty_block(stripe_rows, n_total)
            engine.gemm(a, b, buf, c0)
            part = engine.edges(buf, cutoff, s1 - s0, c0, n_total, s0, upper_only)
        seen_cells += (s1 - s0) * (n_total - c0)
        seen_edges += len(part[2])
        for acc, p in zip(out, part):
            acc.append(p)
    if fused:
        fused.free()
    return tuple(np.concatenate(p) if p else np.empty(0, dtype=d)
                 for p, d in zip(out, (np.uint32, np.uint32, np.float32)))
