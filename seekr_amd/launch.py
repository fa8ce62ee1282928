"""Process-group bootstrap for one-process-per-GPU runs (torchrun-style environment).

Reads RANK / LOCAL_RANK / WORLD_SIZE, binds the rank to its GPU and creates the RCCL
communicator.  The 128-byte ncclUniqueId is handed from rank 0 to the others through a small
file in /tmp keyed by the launcher's PID (all workers of one `torch.distributed.run` share a
parent) and MASTER_PORT, so no torch, MPI or extra socket is needed and stale files of earlier
launches cannot be picked up.
"""
import os
import tempfile
import time

# the host driver only supports dmabuf IPC; RCCL's peer-to-peer setup fails without this
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from seekr_amd import _lib  # noqa: E402
from seekr_amd.distributed import RcclComm, SingleComm  # noqa: E402


def world():
    rank = int(os.environ.get("RANK", "0"))
    size = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if "SEEKR_FORCE_DEVICE" in os.environ:  # test hook: several ranks on one GPU (with tests/mock_rccl)
        local = int(os.environ["SEEKR_FORCE_DEVICE"])
    return rank, size, local


def _rendezvous_path():
    key = "{}_{}_{}".format(os.getppid(), os.environ.get("MASTER_PORT", "0"),
                            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return os.path.join(tempfile.gettempdir(), "seekr_amd_rccl_id_" + key)


def init(timeout_s=120.0):
    """Returns (ctx, comm).  comm is a SingleComm when WORLD_SIZE == 1."""
    rank, size, local = world()
    ctx = _lib.Context(local)
    if size == 1:
        return ctx, SingleComm()
    path = _rendezvous_path()
    if rank == 0:
        uid = _lib.comm_unique_id()
        tmp = path + ".tmp{}".format(os.getpid())
        with open(tmp, "wb") as fh:
            fh.write(uid)
        os.replace(tmp, path)  # atomic publish
    else:
        deadline = time.time() + timeout_s
        while not os.path.exists(path):
            if time.time() > deadline:
                raise TimeoutError("rank {}: no RCCL id at {} after {} s".format(rank, path, timeout_s))
            time.sleep(0.01)
        with open(path, "rb") as fh:
            uid = fh.read()
        assert len(uid) == 128
    _lib.comm_init(ctx, size, rank, uid)
    comm = RcclComm(ctx, rank, size)
    comm.barrier()
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return ctx, comm
