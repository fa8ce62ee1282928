"""SEEKR_DEVICES — the GPUs of one node behind the reference's own API.

    SEEKR_DEVICES=all  (or 0,1,2,3)   seekr_kmer_counts transcripts.fa -o counts.npy -b -rl
                                      seekr_pearson counts.npy counts.npy -o r.npy -bi -bo
    >>> os.environ["SEEKR_DEVICES"] = "all"          # BEFORE `import seekr_amd` (see _lib.prepare_runtime_env), or exported
    >>> BasicCounter("transcripts.fa").get_counts(); pearson(c, c)

Unset (or naming one device) everything runs as before on one GPU.  With several devices ONE host process drives them
all, one Python thread per GPU (every call into libseekr_hip releases the interpreter lock): the rows — transcripts — are
cut into contiguous ranges, one per GPU, in input order (balanced by bases for FASTA input), every GPU uploads, counts and
downloads its own range over its own PCIe link, and the three steps that need all rows are the ones of
seekr_amd.distributed: the float32 column sums travel from GPU to GPU in row order (mean and std come out bit-identical to
one GPU's, kmer_counts.py:168,174), the Log2.post minimum is a NaN-propagating all-reduce (:208), and for Pearson the
prepared operands are all-gathered over RCCL / xGMI (pearson.py:41) — each GPU then produces its row block of r in row
stripes, stripe s going to the host (or to its place in the .npy file) while stripe s + 1 is contracted.  Blocks below
the diagonal of a self-comparison are asked for with their mirror's bits (skr_pearson_gemm_op_rows), so r does not depend
on the number of GPUs: it is the one-GPU result bit for bit, as the counts and the statistics are.

The same stripe loop serves ONE GPU when r would not fit its memory (pearson.py:41 is limited by host RAM only) or goes
straight to a file: `pearson()` picks it by itself, SEEKR_PEARSON_STRIPE_ROWS forces a stripe height (tests).

Device data moves over RCCL (SEEKR_TRANSPORT=rccl) or as peer copies pulled by the receiver and ordered by events
(SEEKR_TRANSPORT=peer: PeerComm — no RCCL, the SDMA engines); unset = RCCL, and peer copies when RCCL cannot be set up.

Errors keep the reference's types: what one GPU's range raises (a sequence of length k - 1: ZeroDivisionError,
kmer_counts.py:144) is agreed on by all GPU threads before any of them enters a collective, every thread leaves the job,
and the caller sees that one exception, once.
"""
import os
import queue
import threading

import numpy as np

from seekr_amd import _lib
from seekr_amd.distributed import (HipEngine, RcclComm, SingleComm, allgather_operand, shard_bounds, sharded_normalize,
                                   sharded_normalize_prepare)


# ------------------------------------------------------------------------------ which devices --
def _test_hooks():
    return os.environ.get("SEEKR_TEST_HOOKS") == "1"


def requested_devices():
    """The device list SEEKR_DEVICES asks for when it names MORE than one GPU, else None (the one-GPU path: unset, empty,
    or a single device — which _lib.default_context() then uses)."""
    spec = os.environ.get("SEEKR_DEVICES", "").strip()
    if not spec:
        return None
    n = _lib.device_count()
    if spec.lower() == "all":
        devices = list(range(n))
    else:
        try:
            devices = [int(t) for t in spec.split(",") if t.strip()]
        except ValueError:
            raise ValueError("SEEKR_DEVICES must be 'all' or a comma-separated list of device numbers, got {!r}".format(spec))
    bad = [d for d in devices if d < 0 or d >= n]
    if bad:
        raise ValueError("SEEKR_DEVICES={!r} names device {} but {} device(s) are visible".format(spec, bad[0], n))
    if len(set(devices)) != len(devices) and not _test_hooks():
        # (tests put several ranks on the one GPU of a test box, over tests/mock_rccl: RCCL itself refuses that)
        raise ValueError("SEEKR_DEVICES={!r} names a device twice".format(spec))
    return devices if len(devices) > 1 else None


def bounds_by_bases(lengths, size):
    """Contiguous row ranges with near-equal numbers of BASES (counting time follows the bases, not the sequences; real
    transcript sets are length-skewed): bounds[g] = the first sequence whose start lies at or after g / size of the total."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if n == 0:
        return [0] * (size + 1)
    starts = np.concatenate(([0], np.cumsum(lengths)[:-1]))
    total = int(lengths.sum())
    if total == 0:
        return shard_bounds(n, size)
    cuts = [int(np.searchsorted(starts, (total * g + size - 1) // size, side="left")) for g in range(size)] + [n]
    for g in range(1, size + 1):
        cuts[g] = max(cuts[g], cuts[g - 1])
    cuts[0] = 0
    return cuts


# ------------------------------------------------------------------------------ the group ------
class GroupBroken(RuntimeError):
    """A GPU thread left a job without the others' agreement (an error inside a collective phase): the group is discarded."""


class _PeerFailed(Exception):
    """Raised in the threads whose own phase succeeded when another thread's did not: they leave the job with it."""


class HostCollectives:
    """all-reduce / barrier of a few host numbers between the threads of a DeviceGroup: shared slots and a
    threading.Barrier — no device round trip (the RCCL form costs a stream drain and a kernel per call)."""

    def allreduce(self, values, op):
        rows = self.group.gather(self.rank, [float(v) for v in values])
        fn = {"sum": sum, "max": max, "min": min}[op]
        return [fn(r[i] for r in rows) for i in range(len(rows[0]))]

    def barrier(self):
        self.group.gather(self.rank, None)


class GroupRcclComm(HostCollectives, RcclComm):
    """RcclComm for the threads of one process: device data (statistic vectors, operand shards) moves over RCCL exactly
    as between processes; the handful of host scalars is exchanged in memory."""

    def __init__(self, group, ctx, rank, size):
        RcclComm.__init__(self, ctx, rank, size)
        self.group = group

    def chain_for(self, engine, n_cols):
        # the peer-mailbox chain exchanges HIP IPC handles, which a process cannot open on itself: send/recv here
        self._chain, self._chain_note = False, "one process drives all GPUs: send/recv chain"
        return None

    def barrier(self):
        self.ctx.sync()
        HostCollectives.barrier(self)


class PeerComm(HostCollectives):
    """The communicator of the in-process group WITHOUT RCCL (SEEKR_TRANSPORT=peer, or RCCL could not be set up): every
    transfer is a peer copy over xGMI PULLED by the receiver on its own communication stream (skr_peer_copy_rows:
    hipMemcpyPeerAsync — the SDMA engines, no CU taken from the contraction), ordered by events (skr_event_*).  The sender's
    half of an exchange is a note in the receiver's mailbox: (the matrix, an event behind the kernels that produced its
    rows).  Same interface as RcclComm where seekr_amd.multi / distributed use it: send_vec / recv_vec (the column-sum
    chain), allgather_rows (the prepared operands), wait, allreduce / barrier (host, HostCollectives)."""
    SEND_TICKET_OPTIONAL = True

    def __init__(self, group, ctx, rank, size):
        self.group, self.ctx, self.rank, self.size = group, ctx, rank, size
        self._tickets, self._next, self._unacked = {}, 0, []
        self._chain_note = "peer copies (one process drives all GPUs)"

    def _post(self, dst, mat, rows, ack):
        self.group.mail[(self.rank, dst)].put((mat, rows, _lib.Event(self.ctx, comm=False), ack))
        if ack:
            self._unacked.append(dst)

    def _pull(self, src, dst_mat, drow0):
        """The next note of `src`: its rows into dst_mat[drow0:] on my communication stream; returns the rows copied."""
        try:
            mat, rows, ready, ack = self.group.mail[(src, self.rank)].get(timeout=self.group.mail_timeout)
        except queue.Empty:
            raise GroupBroken("GPU thread {} waited {} s for rows of GPU thread {}".format(self.rank, self.group.mail_timeout, src)) from None
        ready.wait_on(self.ctx, comm=True)
        _lib.peer_copy_rows(dst_mat, drow0, mat, 0, rows)
        if ack:  # the sender may want to overwrite its rows: tell it where my copy ends
            self.group.acks[(src, self.rank)].put(_lib.Event(self.ctx, comm=True))
        return rows

    def _drain_acks(self):
        """My compute stream waits for every peer's copy of the vectors I sent (they may be overwritten after this)."""
        for dst in self._unacked:
            try:
                done = self.group.acks[(self.rank, dst)].get(timeout=self.group.mail_timeout)
            except queue.Empty:
                raise GroupBroken("GPU thread {} waited {} s for GPU thread {} to fetch a vector".format(self.rank, self.group.mail_timeout, dst)) from None
            done.wait_on(self.ctx, comm=False)
        self._unacked = []

    def _ticket(self):
        self._next += 1
        self._tickets[self._next] = _lib.Event(self.ctx, comm=True)
        return self._next

    def send_vec(self, v, dst, want_ticket=True):
        """The receiver pulls; a ticket, when waited for, makes my compute stream wait for EVERY peer's copy of the vectors
        sent so far (RcclComm's send ticket, for transfers that run on the receivers' streams).  Without a ticket the
        acknowledgement is collected at the next ticketed wait or barrier."""
        self._post(dst, v, 1, ack=True)
        if not want_ticket:
            return None
        self._next += 1
        self._tickets[self._next] = None  # a send ticket: drain the acknowledgements
        return self._next

    def recv_vec(self, v, src):
        self._pull(src, v, 0)
        self.wait(self._ticket())

    def allgather_rows(self, shard, full, bounds):
        """shard / full: Operands (their float32-typed storage travels) or Matrices."""
        s = shard.as_matrix() if hasattr(shard, "as_matrix") else shard
        f = full.as_matrix() if hasattr(full, "as_matrix") else full
        for step in range(1, self.size):  # (no acknowledgement: the shards live until the job's closing barrier)
            self._post((self.rank - step) % self.size, s, s.rows, ack=False)
        if s.rows:  # the own shard: a device copy, behind the kernels that produced it
            _lib.Event(self.ctx, comm=False).wait_on(self.ctx, comm=True)
            _lib.peer_copy_rows(f, bounds[self.rank], s, 0, s.rows)
        for step in range(1, self.size):
            src = (self.rank + step) % self.size
            got = self._pull(src, f, bounds[src])
            assert got == bounds[src + 1] - bounds[src]
        return self._ticket()

    def wait(self, ticket):
        if ticket is None:
            return
        ev = self._tickets.pop(ticket)
        if ev is None:
            self._drain_acks()
        else:
            ev.wait_on(self.ctx, comm=False)

    def barrier(self):
        self._drain_acks()
        self.ctx.sync()
        HostCollectives.barrier(self)


class Rank:
    """What a job sees of its GPU: ctx, comm, and the agreement primitive."""

    def __init__(self, group, rank, ctx, comm):
        self.group, self.rank, self.size, self.ctx, self.comm = group, rank, group.size, ctx, comm
        self.agreed_error = None

    def engine(self, precision=_lib.PREC_F16X3, row_standardize=True):
        return ApiHipEngine(self.ctx, precision, row_standardize=row_standardize)

    def phase(self, fn):
        """Run fn() — local work that may raise the reference's exceptions — then agree with every other GPU thread: if
        any raised, ALL leave the job here, before the next collective (the failing ones with their own exception)."""
        err, out = None, None
        try:
            out = fn()
        except BaseException as e:  # noqa: BLE001 - re-raised below, after the agreement
            err = e
        errors = self.group.gather(self.rank, err)
        first = next((e for e in errors if e is not None), None)
        if first is not None:
            self.agreed_error = err if err is not None else _PeerFailed(first)
            raise self.agreed_error
        return out


def _hip_backend(group, rank):
    ctx = _lib.Context(group.devices[rank])
    size = group.size
    if group.transport == "rccl":
        _lib.comm_init(ctx, size, rank, group.uid)
        comm = GroupRcclComm(group, ctx, rank, size)
    else:
        comm = PeerComm(group, ctx, rank, size)
    st = Rank(group, rank, ctx, comm)

    def ring():  # one vector round the ring, checked: the first multi-GPU call diagnoses itself instead of hanging later
        send = ctx.from_numpy(np.full((1, 256), float(rank + 1), np.float32))
        recv = ctx.zeros(1, 256)
        if group.transport == "rccl":
            t = _lib.comm_sendrecv(ctx, send, 0, 1, (rank + 1) % size, recv, 0, 1, (rank - 1) % size)
            _lib.comm_wait(ctx, t)
        else:
            t = comm.send_vec(send, (rank + 1) % size)
            comm.recv_vec(recv, (rank - 1) % size)
            comm.wait(t)
        got, want = recv.to_numpy(), float((rank - 1) % size + 1)
        if not (got == want).all():
            raise _lib.SeekrHipError("GPU {} (rank {} of {}): the {} ring self-test delivered {} instead of {}".format(
                group.devices[rank], rank, size, group.transport, got[0, 0], want))
        comm.barrier()  # (peer transport: `send` is read by the neighbour's copy until here)
        # how many ranks the transport itself saw: a one-element all-reduce on the devices over RCCL (what "did RCCL carry
        # data between N GPUs" is answered from: group_info()), the host gather for peer copies
        if group.transport == "rccl":
            seen = int(round(_lib.comm_allreduce(ctx, [1.0], "sum")[0]))
        else:
            seen = len(group.gather(rank, 1))
        if seen != size:
            raise _lib.SeekrHipError("GPU {} (rank {} of {}): the {} all-reduce counted {} ranks".format(
                group.devices[rank], rank, size, group.transport, seen))
        if rank == 0:
            group.n_ranks_seen = seen
    st.phase(ring)
    return st


class DeviceGroup:
    """One thread per GPU, alive for the life of the process (RCCL's set-up costs seconds: it is paid once).  run(fn, spec)
    executes fn(rank_state, spec) on every thread and returns the results in rank order; the lowest rank's exception, if
    any, is re-raised in the caller."""

    def __init__(self, devices, backend=None, uid=None, transport="rccl"):
        self.devices, self.size = list(devices), len(devices)
        self.uid, self.transport = uid, transport
        # peer transport: one mailbox per ordered pair of GPU threads (the sender's half of a transfer is a note in it)
        self.mail = {(a, b): queue.Queue() for a in range(self.size) for b in range(self.size) if a != b}
        self.acks = {(a, b): queue.Queue() for a in range(self.size) for b in range(self.size) if a != b}  # (sender, receiver)
        self.mail_timeout = 120.0
        self.broken = False
        self.n_ranks_seen = None  # set by the set-up's all-reduce (_hip_backend)
        self._backend = backend or _hip_backend
        self._barrier = threading.Barrier(self.size)
        self._slots = [None] * self.size
        self._inbox = [queue.Queue() for _ in range(self.size)]
        self._outbox = queue.Queue()
        self._lock = threading.Lock()
        self._threads = [threading.Thread(target=self._worker, args=(rank,), daemon=True, name="seekr-gpu%d" % d)
                         for rank, d in enumerate(self.devices)]
        for t in self._threads:
            t.start()
        self._collect("set-up")

    # -- host-side exchange between the threads: everyone deposits, everyone reads
    def gather(self, rank, value):
        try:
            self._slots[rank] = value
            self._barrier.wait()
            out = list(self._slots)
            self._barrier.wait()  # nobody overwrites a slot before everybody has read
            return out
        except threading.BrokenBarrierError:
            raise GroupBroken("another GPU thread failed inside a collective phase") from None

    def _worker(self, rank):
        state = None
        try:
            state = self._backend(self, rank)
            self._outbox.put((rank, True, None))
        except BaseException as e:  # noqa: BLE001
            agreed = state is not None and state.agreed_error is e
            self._fail(rank, e, agreed or isinstance(e, _PeerFailed))
            return
        while True:
            job = self._inbox[rank].get()
            if job is None:
                return
            fn, spec = job
            state.agreed_error = None
            try:
                self._outbox.put((rank, True, fn(state, spec)))
            except BaseException as e:  # noqa: BLE001
                if not self._fail(rank, e, state.agreed_error is e):
                    return

    def _fail(self, rank, error, agreed):
        """Report; an error nobody agreed on may have left the others in a host barrier: break it, the group is done."""
        if not agreed and not isinstance(error, GroupBroken):
            self.broken = True
            self._barrier.abort()
        if isinstance(error, GroupBroken):
            self.broken = True
        self._outbox.put((rank, False, error))
        return not self.broken

    def _collect(self, what):
        results, errors = [None] * self.size, {}
        for _ in range(self.size):
            rank, ok, value = self._outbox.get()
            if ok:
                results[rank] = value
            else:
                errors[rank] = value
        real = {r: e for r, e in errors.items() if not isinstance(e, (_PeerFailed, GroupBroken))}
        if real:
            raise real[min(real)]
        if errors:
            raise GroupBroken("the multi-GPU {} failed: {}".format(what, errors[min(errors)]))
        return results

    def run(self, fn, spec):
        with self._lock:
            if self.broken:
                raise GroupBroken("this device group is no longer usable")
            for q in self._inbox:
                q.put((fn, spec))
            return self._collect("job")

    def close(self):
        for q in self._inbox:
            q.put(None)


_group = None
_group_lock = threading.Lock()


def requested_transport():
    """SEEKR_TRANSPORT: 'rccl' (ncclSend / ncclRecv between the GPU threads: what the north_star names), 'peer' (peer copies
    over xGMI pulled by the receiver, no RCCL: PeerComm), or unset = 'auto': RCCL, and if it cannot be set up — the library
    is missing, ncclCommInitRank or the ring self-test fails — peer copies, with one line on stderr saying so."""
    t = os.environ.get("SEEKR_TRANSPORT", "auto").strip().lower() or "auto"
    if t not in ("auto", "rccl", "peer"):
        raise ValueError("SEEKR_TRANSPORT must be rccl, peer or auto, got {!r}".format(t))
    return t


def group_info():
    """What is actually behind SEEKR_DEVICES in this process, for records (bench.py's e2e.seekr_devices_all, the tests'
    info.json): the transport that carries device data — 'rccl' or 'peer', never the one merely asked for — the number of
    ranks its set-up all-reduce counted, and the IPC mode the HIP runtime started with."""
    g = _group
    if g is None:
        return {"group_size": 0, "transport": None, "n_ranks_seen": None, "ipc_env_at_load": _lib.ipc_env_at_load()}
    return {"group_size": g.size, "transport": g.transport, "n_ranks_seen": g.n_ranks_seen,
            "n_ranks_seen_by": "RCCL all-reduce on the devices" if g.transport == "rccl" else "host gather (peer copies: no RCCL)",
            "transport_asked": requested_transport(), "ipc_env_at_load": _lib.ipc_env_at_load()}


def late_ipc_note():
    """Why RCCL may have failed through no fault of its own: HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, which RCCL's
    peer-to-peer set-up needs on these nodes) was not in the environment when the HIP runtime started — SEEKR_DEVICES was set
    inside the process, after seekr_amd had been used.  The package sets the variable itself only when it sees SEEKR_DEVICES
    before its first HIP call (seekr_amd._lib.prepare_runtime_env).  '' when the runtime did start with it."""
    if _lib.ipc_env_at_load() == "0":
        return ""
    return ("{0} was {1!r}, not '0', when the HIP runtime started: SEEKR_DEVICES was first seen after seekr_amd's first HIP "
            "call, too late to set it — export SEEKR_DEVICES (or {0}=0) before the process starts, or set os.environ"
            "['SEEKR_DEVICES'] before importing seekr_amd".format(_lib.IPC_VAR, _lib.ipc_env_at_load()))


def group_for(devices):
    """The process-wide DeviceGroup for this device list (created on first use, replaced when the list, the transport
    asked for changes or the group broke)."""
    global _group
    want = requested_transport()
    with _group_lock:
        if _group is not None and (_group.broken or _group.devices != list(devices) or (want != "auto" and _group.transport != want)):
            _group.close()
            _group = None
        if _group is None:
            if want in ("auto", "rccl"):
                try:
                    _group = DeviceGroup(devices, uid=_lib.comm_unique_id(), transport="rccl")
                except Exception as e:  # noqa: BLE001
                    late = late_ipc_note()
                    if want == "rccl":
                        if late:
                            raise type(e)("{} [{}]".format(e, late)) from e
                        raise
                    import sys
                    print("seekr_amd: RCCL could not be set up between the GPUs of SEEKR_DEVICES ({}: {}){}; using peer copies "
                          "(SEEKR_TRANSPORT=peer)".format(type(e).__name__, e, " — " + late if late else ""), file=sys.stderr)
            if _group is None:
                _group = DeviceGroup(devices, transport="peer")
        return _group


# ------------------------------------------------------------------------------ the engine -----
class ApiHipEngine(HipEngine):
    """HipEngine plus what the API-level jobs need around the kernels: host <-> device rows, counting from the caller's
    sequences, stripe buffers and the two sinks of r."""

    def count(self, source, lo, hi, k, log2_pre, alphabet, two_bit):
        kind, payload = source
        if kind == "fasta":
            packed = payload.pack(self.ctx, lo, hi - lo, alphabet)
        elif two_bit:
            packed = self.ctx.pack(payload[lo:hi], alphabet)
        else:
            return _lib.count_generic(self.ctx, payload[lo:hi], alphabet, k, np.float32, log2_pre=log2_pre)
        x = _lib.count_per_kb(self.ctx, packed, k, log2_pre=log2_pre)
        packed.free()
        return x

    def user_vector(self, vec, n_cols, n_rows=None):
        from seekr_amd.kmer_counts import _as_device_vector
        return _as_device_vector(self.ctx, vec, n_cols, n_rows)

    def vec_to_host(self, v):
        return v.vector()

    def upload(self, rows):
        return self.ctx.from_numpy(rows) if len(rows) else self.ctx.empty(0, rows.shape[1], rows.dtype)

    def download(self, x, out):
        if x.rows:
            x.to_numpy(out=out)

    # -- float64 rows (CSV / integer inputs, pearson.py:35-41 in float64)
    def rows_f64(self, rows, row_standardize, pad):
        """Device rows ready for skr_pearson_gemm_f64: standardised in the rows' OWN dtype (float32 stays float32 until
        the product promotes it, as numpy does), float64 afterwards, zero-padded to whole 16-column stages when `pad`."""
        x = self.upload(rows)
        cols = rows.shape[1]
        if rows.dtype == np.float32:
            if row_standardize and x.rows:
                x = _lib.row_standardize(self.ctx, x)
            return self.upload(x.to_numpy().astype(np.float64) if x.rows else np.empty((0, cols), np.float64))
        if not row_standardize:
            return x
        z = self.ctx.zeros(x.rows, (cols + 15) // 16 * 16 if pad else cols, np.float64)
        if x.rows:
            _lib.row_standardize(self.ctx, x, z)
        return z

    def gather_room(self, z, rows):
        """Where the rows of all GPUs will be gathered (allocated inside a phase: see pearson_job)."""
        return self.ctx.empty(rows, z.cols, z.dtype)

    def allgather_matrix(self, comm, z, bounds, full=None):
        if comm.size == 1:
            return z
        full = self.ctx.empty(bounds[-1], z.cols, z.dtype) if full is None else full
        comm.wait(comm.allgather_rows(z, full, bounds))
        return full

    def gemm_f64(self, a, b, r, K, col0=0, symmetric=False):
        _lib.pearson_gemm_f64(self.ctx, a, b, r, K, symmetric=symmetric, col0=col0)

    def view_matrix(self, x, row0, nrows):
        return x.view(row0, nrows)

    def block(self, rows, cols, dtype):
        return self.ctx.empty(rows, cols, dtype)

    def free_bytes(self):
        return self.ctx.mem_info()[0]

    def mark(self):
        return self.ctx.mark()

    def release_mark(self, mark):
        self.ctx.mark_release(mark)


class HostSink:
    """r lands in the caller's array: every GPU copies its stripes straight into its rows, over its own PCIe link."""

    def __init__(self, out):
        self.out = out

    def put(self, buf, nrows, row0, mark):
        buf.to_numpy_at(mark, self.out[row0:row0 + nrows])


class NpySink:
    """r lands in a .npy file, stripe by stripe, never whole in host memory (np.save(outfile, dist), pearson.py:43)."""

    def __init__(self, path, dtype, rows, cols):
        # np.save(outfile, dist) touches the file only once dist exists (pearson.py:41-43): the stripes go to a temporary
        # file beside it, which takes the name when the job is done (commit) and disappears when it fails (discard) — a
        # failed job neither leaves a full-size, valid-looking file of zeros nor destroys what was there
        self.final = _lib.npy_path(path)
        self.path = "{}.part{}.npy".format(self.final[:-4], os.getpid())
        self.row_bytes = int(cols) * np.dtype(dtype).itemsize
        self.offset = _lib.npy_create(self.path, dtype, rows, cols)

    def put(self, buf, nrows, row0, mark):
        buf.write_rows_at(mark, self.path, self.offset + row0 * self.row_bytes, 0, nrows)

    def commit(self):
        os.replace(self.path, self.final)

    def discard(self):
        try:
            os.unlink(self.path)
        except OSError:
            pass


# ------------------------------------------------------------------------------ get_counts -----
class CountSpec:
    def __init__(self, source, bounds, k, log2, mean, std, alphabet, two_bit, out):
        self.source, self.bounds, self.k, self.log2, self.mean, self.std = source, bounds, k, log2, mean, std
        self.alphabet, self.two_bit, self.out = alphabet, two_bit, out


def counts_job(st, spec):
    """BasicCounter.get_counts (kmer_counts.py:194-209) on this GPU's row range: count (+ Log2.pre), the column statistics
    over ALL ranges (rank-chained float32 sums), centre / standardise / Log2.post, rows to the caller's matrix."""
    eng = st.engine()
    lo, hi = spec.bounds[st.rank], spec.bounds[st.rank + 1]
    n_total = spec.bounds[-1]
    x = st.phase(lambda: eng.count(spec.source, lo, hi, spec.k, spec.log2 == "Log2.pre", spec.alphabet, spec.two_bit))
    n_cols = eng.cols(x)
    mean = spec.mean if isinstance(spec.mean, bool) else eng.user_vector(spec.mean, n_cols, int(spec.bounds[-1]))
    std = spec.std if isinstance(spec.std, bool) else eng.user_vector(spec.std, n_cols, int(spec.bounds[-1]))
    # Log2.pre went into the counting flush: the normaliser only has the post step left to do
    log2 = "Log2.post" if spec.log2 == "Log2.post" else "Log2.none"
    center, scale, has_nan = sharded_normalize(eng, st.comm, x, n_total, log2, mean, std)
    eng.download(x, spec.out[lo:hi])
    st.comm.barrier()  # nothing of this job is freed while a peer may still be copying from it (peer transport)
    first = st.rank == 0
    return (eng.vec_to_host(center) if first and spec.mean is True else None,
            eng.vec_to_host(scale) if first and spec.std is True else None, bool(has_nan))


def run_counts(run, size, source, lengths, k, log2, mean, std, alphabet, two_bit, n_cols):
    """Cut the sequences into `size` ranges, run counts_job on every GPU thread (`run` = DeviceGroup.run), return
    (counts, mean, std, has_nan) — mean / std None where they were not computed."""
    bounds = bounds_by_bases(lengths, size)
    out = _lib.host_pool.empty((len(lengths), n_cols), np.float32)
    parts = run(counts_job, CountSpec(source, bounds, k, log2, mean, std, alphabet, two_bit, out))
    return out, parts[0][0], parts[0][1], any(p[2] for p in parts)


def counter_get_counts(counter, devices):
    """BasicCounter.get_counts() over several GPUs: fills counter.counts / .mean / .std; returns has_nan."""
    if counter._fasta is not None:
        source, lengths = ("fasta", counter._fasta), counter._fasta.lengths()
    else:
        if counter._seqs is None:
            raise TypeError("BasicCounter has no sequences: pass infasta or assign `seqs`")
        seqs = list(counter._seqs)
        source, lengths = ("strings", seqs), np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    # True = compute, False = skip, anything else = the caller's vector (kmer_counts.py:104-117); `is`-tests as the reference's
    mean = True if counter.mean is True else (False if counter.mean is False else counter.mean)
    std = True if counter.std is True else (False if counter.std is False else counter.std)
    counts, mean_out, std_out, has_nan = run_counts(group_for(devices).run, len(devices), source, lengths, counter.k,
                                                    counter.log2, mean, std, counter.alphabet, counter._two_bit,
                                                    len(counter.alphabet) ** counter.k)
    counter.counts = counts
    if mean_out is not None:
        counter.mean = mean_out
    if std_out is not None:
        counter.std = std_out
    return has_nan


# ------------------------------------------------------------------------------ pearson --------
class PearsonSpec:
    def __init__(self, c1, c2, bounds1, bounds2, f64, pad, precision, row_standardize, sink, stripe_rows, out_dtype):
        self.c1, self.c2, self.bounds1, self.bounds2, self.f64, self.pad = c1, c2, bounds1, bounds2, f64, pad
        self.precision, self.row_standardize, self.sink, self.stripe_rows = precision, row_standardize, sink, stripe_rows
        self.out_dtype = np.dtype(out_dtype)


STRIPE_TARGET_BYTES = 2 << 30  # per stripe buffer (two of them): large enough to fill the chip, small beside 288 GB


def stripe_height(rows, n_out, itemsize, free_bytes, forced=None):
    """Rows per stripe of r: the forced height (SEEKR_PEARSON_STRIPE_ROWS / tests), else what two stripe buffers of
    ~2 GiB — or 40 % of the free device memory, if that is less — hold, in multiples of the contraction's 256-row tile."""
    if rows <= 0:
        return 1
    if forced:
        return max(1, min(int(forced), rows))
    row_bytes = max(1, n_out * itemsize)
    budget = min(STRIPE_TARGET_BYTES, int(free_bytes * 0.4))
    h = max(256, budget // row_bytes // 256 * 256)
    return rows if h >= rows else h


def _match_pair(eng, x1, z1, x2, z2):
    """Two operands filled separately must share a storage kind (operand.hip: match_layouts; every rank holds the same
    global flags by now, so every rank decides alike): one in the float32 layout -> so is the other; two f16f8 ones whose
    row means do not go together, or an f16f8 one beside a three-product one -> both three-product."""
    k1, k2 = eng.layout(z1), eng.layout(z2)
    if k1 == 3 and k2 == 3 and not z1.x8_pair_bound(z2)[1]:
        return eng.prepare_f16x3(x1, op=z1), eng.prepare_f16x3(x2, op=z2)
    if k1 == k2:
        return z1, z2
    if k1 != 0 and k2 != 0:
        return (eng.prepare_f16x3(x1, op=z1), z2) if k1 == 3 else (z1, eng.prepare_f16x3(x2, op=z2))
    return (z1, eng.prepare_f32(x2)) if k1 == 0 else (eng.prepare_f32(x1), z2)


def pearson_job(st, spec):
    """This GPU's row block of pearson(c1, c2) (pearson.py:35-41): its rows of c1 against ALL rows of c2, the latter
    prepared range by range on the GPUs and all-gathered; r leaves in row stripes through spec.sink."""
    eng = st.engine(spec.precision, spec.row_standardize)
    comm, g = st.comm, st.rank
    b1, b2 = spec.bounds1, spec.bounds2
    lo, hi = b1[g], b1[g + 1]
    same = spec.c2 is None
    n_out, K = b2[-1], spec.c1.shape[1]
    rows = hi - lo
    # Everything this GPU must allocate or upload — its rows, the gathered operand, the stripe buffers — happens inside
    # phases: a failure on ONE GPU (out of memory, say) is agreed on by all before anybody enters a device collective.
    # Outside a phase the other GPU threads would sit in ncclSend / ncclRecv with no timeout, run() would never return, and
    # — pearson() holding the API lock — neither would the process.
    if spec.f64:
        def local_f64():
            a = eng.rows_f64(spec.c1[lo:hi], spec.row_standardize, spec.pad)
            own = a if same else eng.rows_f64(spec.c2[b2[g]:b2[g + 1]], spec.row_standardize, spec.pad)
            return a, own, (eng.gather_room(own, n_out) if comm.size > 1 else None)
        z1, own, room = st.phase(local_f64)
        full = eng.allgather_matrix(comm, own, b2, room)
    else:
        def local_f32():
            return eng.upload(spec.c1[lo:hi]), (None if same else eng.upload(spec.c2[b2[g]:b2[g + 1]]))
        x1, x2 = st.phase(local_f32)
        z1 = sharded_normalize_prepare(eng, comm, x1, b1[-1], "Log2.none", False, False, keep_counts=False)[3]
        if same:
            z2 = z1
        else:
            z2 = sharded_normalize_prepare(eng, comm, x2, n_out, "Log2.none", False, False, keep_counts=False)[3]
            z1, z2 = _match_pair(eng, x1, z1, x2, z2)
        room = st.phase(lambda: eng.empty_operand(n_out, eng.cols(z2)) if comm.size > 1 else None)
        full = allgather_operand(eng, comm, z2, b2, room)

    def stripe_buffers():
        h = stripe_height(rows, n_out, spec.out_dtype.itemsize, eng.free_bytes(), spec.stripe_rows)
        return h, ([eng.block(h, n_out, spec.out_dtype) for _ in range(2 if rows > h else 1)] if rows else [])
    height, bufs = st.phase(stripe_buffers)
    _stripe_loop(eng, spec, z1, full, bufs, rows, height, lo, n_out, K, same)
    st.comm.barrier()  # the operand shards stay alive until every GPU has pulled them (peer transport)
    return None


def _stripe_loop(eng, spec, z1, full, bufs, rows, height, lo, n_out, K, same):
    """Stripe s of this GPU's rows is contracted while stripe s - 1 leaves through the sink, behind its mark.  A mark whose
    stripe never reaches the sink (an exception on the way) is handed back: marks are single-use slots of the ctx."""
    pending = None
    try:
        for i, s0 in enumerate(range(0, rows, height)):
            m, buf = min(height, rows - s0), bufs[i % 2]
            g0 = lo + s0  # global index of the stripe's first row
            if spec.f64:
                a = eng.view_matrix(z1, s0, m)
                if same:  # float64 products round once: any tiling gives the one-block call's bits
                    if g0:
                        eng.gemm_f64(a, eng.view_matrix(full, 0, g0), buf, K)
                    eng.gemm_f64(a, a, buf, K, col0=g0, symmetric=True)
                    if g0 + m < n_out:
                        eng.gemm_f64(a, eng.view_matrix(full, g0 + m, n_out - g0 - m), buf, K, col0=g0 + m)
                else:
                    eng.gemm_f64(a, full, buf, K)
            elif same:
                eng.gemm_rows(eng.view(z1, s0, m), full, g0, buf)
            else:
                eng.gemm(eng.view(z1, s0, m), full, buf, 0)
            mark = eng.mark()
            if pending is not None:  # stripe s - 1 leaves while stripe s is contracted
                leaving, pending = pending, (buf, m, g0, mark)
                spec.sink.put(*leaving)
            else:
                pending = (buf, m, g0, mark)
        if pending is not None:
            leaving, pending = pending, None
            spec.sink.put(*leaving)
    except BaseException:
        if pending is not None:
            eng.release_mark(pending[3])
        raise


class _Solo:
    """The stripe loop on ONE GPU (r larger than its memory, or straight to a file): a 'group' of the default context."""
    rank, size = 0, 1

    def __init__(self, ctx):
        self.ctx, self.comm = ctx, SingleComm()

    def engine(self, precision=_lib.PREC_F16X3, row_standardize=True):
        return ApiHipEngine(self.ctx, precision, row_standardize=row_standardize)

    def phase(self, fn):
        return fn()  # nobody to agree with


def forced_stripe_rows():
    v = os.environ.get("SEEKR_PEARSON_STRIPE_ROWS", "").strip()
    return int(v) if v else None


def run_pearson(c1, c2, w1, w2, row_standardize, precision, devices, out=None, outfile=None):
    """pearson(c1, c2) by row stripes, on the GPUs of `devices` (None: the default context alone).  c2 None = the
    self-comparison.  The result goes to `out` (a host array, created when None and no outfile) or to the .npy file."""
    same = c2 is None
    f64 = not (w1 == np.float32 and w2 == np.float32)
    out_dtype = np.float64 if f64 else np.float32
    m_rows, n_rows = c1.shape[0], (c1 if same else c2).shape[0]
    size = len(devices) if devices else 1
    # float64: skr_pearson pads standardised rows to whole 16-column stages — unless the two inputs differ in dtype (then
    # the float32 one was standardised on its own and the product runs on the rows as they are)
    pad = f64 and w1 == np.float64 and w2 == np.float64 and row_standardize
    c1 = np.ascontiguousarray(c1, dtype=w1)
    c2 = None if same else np.ascontiguousarray(c2, dtype=w2)
    if outfile is not None and out is None:
        sink = NpySink(outfile, out_dtype, m_rows, n_rows)
    else:
        out = _lib.host_pool.empty((m_rows, n_rows), out_dtype) if out is None else out
        sink = HostSink(out)
    spec = PearsonSpec(c1, c2, shard_bounds(m_rows, size), shard_bounds(n_rows, size), f64, pad, precision, row_standardize,
                       sink, forced_stripe_rows(), out_dtype)
    try:
        if m_rows and n_rows:
            if devices:
                group_for(devices).run(pearson_job, spec)
            else:
                pearson_job(_Solo(_lib.default_context()), spec)
    except BaseException:
        if isinstance(sink, NpySink):
            sink.discard()
        raise
    if isinstance(sink, NpySink):
        sink.commit()
    return out
