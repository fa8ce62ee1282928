"""Process-group bootstrap for one-process-per-GPU runs (torchrun-style environment).

Reads RANK / LOCAL_RANK / WORLD_SIZE, binds the rank to its GPU and creates the RCCL
communicator.  The 128-byte ncclUniqueId is handed from rank 0 to the others through a small
file (no torch, MPI or extra socket needed) in a per-user directory of mode 0700, keyed by the
launcher's PID (all workers of one `torch.distributed.run` share a parent), MASTER_PORT and the
restart count.  The file carries a header the readers verify — magic, launcher PID, port, the PID
of the rank 0 that wrote it (must be alive) and a checksum — so a stale file left by a crashed
earlier launch is ignored, and rank 0 removes whatever is there before it publishes.
"""
import hashlib
import os
import stat
import struct
import tempfile
import time

# the host driver only supports dmabuf IPC; RCCL's peer-to-peer setup fails without this
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from seekr_amd import _lib  # noqa: E402
from seekr_amd.distributed import RcclComm, SingleComm  # noqa: E402

_MAGIC = b"SKRRCCL2"
_HEADER = struct.Struct("<8sqqq")  # magic, launcher pid, port, rank-0 pid


def _test_hooks():
    return os.environ.get("SEEKR_TEST_HOOKS") == "1"


def world():
    rank = int(os.environ.get("RANK", "0"))
    size = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if _test_hooks() and "SEEKR_FORCE_DEVICE" in os.environ:  # test hook: several ranks on one GPU (with tests/mock_rccl)
        local = int(os.environ["SEEKR_FORCE_DEVICE"])
    return rank, size, local


def _rendezvous_path():
    base = os.path.join(tempfile.gettempdir(), "seekr_amd_{}".format(os.getuid()))
    os.makedirs(base, mode=0o700, exist_ok=True)
    st = os.lstat(base)  # lstat: a pre-planted symlink to some other 0700 directory of this user must not pass
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError("{} must be a directory of this user with mode 0700 (not a symlink)".format(base))
    key = "{}_{}_{}".format(os.getppid(), os.environ.get("MASTER_PORT", "0"),
                            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return os.path.join(base, "rccl_id_" + key)


def _pack(uid):
    head = _HEADER.pack(_MAGIC, os.getppid(), int(os.environ.get("MASTER_PORT", "0")), os.getpid())
    return head + uid + hashlib.sha256(head + uid).digest()


def _unpack(blob):
    """The 128-byte id, or None when the file is not (yet) a complete record of THIS launch by a live rank 0."""
    if len(blob) != _HEADER.size + 128 + 32:
        return None
    head, uid, digest = blob[:_HEADER.size], blob[_HEADER.size:_HEADER.size + 128], blob[-32:]
    magic, ppid, port, pid0 = _HEADER.unpack(head)
    if magic != _MAGIC or ppid != os.getppid() or port != int(os.environ.get("MASTER_PORT", "0")):
        return None
    if hashlib.sha256(head + uid).digest() != digest:
        return None
    try:
        os.kill(pid0, 0)  # the rank 0 that wrote it is still running
    except ProcessLookupError:
        return None
    except PermissionError:
        pass
    return uid


def selftest(ctx, comm):
    """One all-reduce and one ring send/recv over the new communicator, checked: the first multi-GPU run diagnoses
    itself — a failure raises with RCCL's own error string instead of hanging in the first shift."""
    rank, size = comm.rank, comm.size
    failure = None
    try:
        total = comm.allreduce([float(rank)], "sum")[0]
        if total != size * (size - 1) / 2.0:
            raise _lib.SeekrHipError("all-reduce returned {} instead of {}".format(total, size * (size - 1) / 2.0))
        import numpy as np
        send = ctx.from_numpy(np.full((1, 256), float(rank + 1), np.float32))
        recv = ctx.zeros(1, 256)
        t = _lib.comm_sendrecv(ctx, send, 0, 1, (rank + 1) % size, recv, 0, 1, (rank - 1) % size)
        _lib.comm_wait(ctx, t)
        got = recv.to_numpy()
        want = float((rank - 1) % size + 1)
        if not (got == want).all():
            raise _lib.SeekrHipError("ring send/recv delivered {} instead of {}".format(got[0, 0], want))
        send.free()
        recv.free()
    except Exception as e:  # noqa: BLE001
        failure = e
    # every rank learns whether ANY rank failed before anyone leaves: a rank that raised alone would leave the others
    # blocked in the barrier that follows (the all-reduce itself failing is the one case only the launcher's
    # kill-on-failure can end: bench.py's parent kills the process group when a child exits non-zero)
    bad = 1.0 if failure is not None else 0.0
    try:
        bad = comm.allreduce([bad], "max")[0]
    except Exception as e:  # noqa: BLE001
        failure = failure or e
    if failure is not None:
        raise _lib.SeekrHipError(
            "rank {} of {}: RCCL self-test failed: {} (HSA_ENABLE_IPC_MODE_LEGACY={}, device {})".format(
                rank, size, failure, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), ctx.device)) from failure
    if bad:
        raise _lib.SeekrHipError("rank {} of {}: RCCL self-test passed here but failed on another rank".format(rank, size))


def init(timeout_s=120.0):
    """Returns (ctx, comm).  comm is a SingleComm when WORLD_SIZE == 1."""
    rank, size, local = world()
    ctx = _lib.Context(local)
    if size == 1:
        return ctx, SingleComm()
    path = _rendezvous_path()
    if rank == 0:
        uid = _lib.comm_unique_id()
        try:
            os.remove(path)  # whatever an earlier launch with the same key left behind
        except FileNotFoundError:
            pass
        tmp = path + ".tmp{}".format(os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as fh:
            fh.write(_pack(uid))
        os.replace(tmp, path)  # atomic publish
    else:
        deadline = time.time() + timeout_s
        uid = None
        while uid is None:
            try:
                with open(path, "rb") as fh:
                    uid = _unpack(fh.read())
            except FileNotFoundError:
                uid = None
            if uid is None:
                if time.time() > deadline:
                    raise TimeoutError("rank {}: no valid RCCL id at {} after {} s".format(rank, path, timeout_s))
                time.sleep(0.01)
    _lib.comm_init(ctx, size, rank, uid)
    comm = RcclComm(ctx, rank, size)
    selftest(ctx, comm)
    comm.barrier()
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return ctx, comm
