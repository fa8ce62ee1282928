"""The orchestration behind SEEKR_DEVICES (seekr_amd/multi.py) without a GPU: the DeviceGroup's threads, the agreement
primitive, the row ranges (ragged, empty) and the two jobs — BasicCounter.get_counts and pearson by row stripes — run over a
numpy engine whose arithmetic is the oracle's and a communicator of in-process queues.  What the GPU tests add is the HIP
engine and RCCL (tests/test_gpu_multi_devices.py); what is checked HERE is that the plumbing neither loses, duplicates nor
reorders a row, that statistics chained across the ranges are bit-identical to the oracle's single pass, and that an
exception raised for one range reaches the caller once, as itself, with every thread back in its loop
(reference: kmer_counts.py:194-209, pearson.py:32-44)."""
import os
import queue
import threading

import numpy as np
import pytest

from oracle import seekr_oracle as orc
from dist_worker import NumpyEngine
from seekr_amd import multi
from seekr_amd.distributed import shard_bounds


class QueueComm(multi.HostCollectives):
    """send/recv of numpy rows between the threads of a group: one queue per ordered pair of ranks."""

    def __init__(self, group, rank, pipes):
        self.group, self.rank, self.size, self.pipes = group, rank, group.size, pipes
        self._pending, self._next = {}, 0

    def send_vec(self, v, dst):
        self.pipes[(self.rank, dst)].put(np.array(v, copy=True))

    def recv_vec(self, v, src):
        v[...] = self.pipes[(src, self.rank)].get(timeout=60)

    def allgather_rows(self, shard, full, bounds):
        full[bounds[self.rank]:bounds[self.rank + 1]] = shard
        for peer in range(self.size):
            if peer != self.rank:
                self.pipes[(self.rank, peer)].put(np.array(shard, copy=True))
        self._next += 1
        self._pending[self._next] = (full, bounds)
        return self._next

    def wait(self, ticket):
        full, bounds = self._pending.pop(ticket)
        for peer in range(self.size):
            if peer != self.rank:
                full[bounds[peer]:bounds[peer + 1]] = self.pipes[(peer, self.rank)].get(timeout=60)


class ApiNumpyEngine(NumpyEngine):
    """The protocol of multi.ApiHipEngine over numpy (arithmetic: the oracle's / dist_worker.NumpyEngine's)."""
    precision = 0

    def __init__(self, row_standardize=True):
        self.rs = row_standardize

    def count(self, source, lo, hi, k, log2_pre, alphabet, two_bit):
        seqs = source[1][lo:hi]
        x = orc.raw_counts(seqs, k, alphabet) if len(seqs) else np.zeros((0, len(alphabet) ** k), np.float32)
        return orc.log2_plus_one(x) if log2_pre else x

    def user_vector(self, vec, n_cols, n_rows=None):
        return np.broadcast_to(np.asarray(vec), (n_cols,))

    def upload(self, rows):
        return np.array(rows, copy=True)

    def download(self, x, out):
        out[...] = x

    def prepare(self, x, center=None, scale=None, post=False, shift=0.0, keep_counts=True, op=None):
        assert center is None and scale is None and not post
        with np.errstate(all="ignore"):
            return (self.row_standardize(x) if self.rs else x), False

    def layout(self, op):
        return 0

    def rows_f64(self, rows, row_standardize, pad):
        with np.errstate(all="ignore"):
            z = self.row_standardize(rows) if row_standardize and len(rows) else rows
        return np.asarray(z, dtype=np.float64)

    def gather_room(self, z, rows):
        return np.zeros((rows, z.shape[1]), z.dtype)

    def allgather_matrix(self, comm, z, bounds, full=None):
        if comm.size == 1:
            return z
        full = np.zeros((bounds[-1], z.shape[1]), z.dtype) if full is None else full
        comm.wait(comm.allgather_rows(z, full, bounds))
        return full

    def gemm_rows(self, a, full, a_row0, r):
        with np.errstate(all="ignore"):
            r[:len(a)] = np.inner(a, full) / a.shape[1]

    def gemm_f64(self, a, b, r, K, col0=0, symmetric=False):
        with np.errstate(all="ignore"):
            r[:len(a), col0:col0 + len(b)] = np.inner(a, b) / K

    def view_matrix(self, x, row0, nrows):
        return x[row0:row0 + nrows]

    def block(self, rows, cols, dtype):
        return np.full((rows, cols), np.nan, dtype)

    def free_bytes(self):
        return 1 << 40

    def mark(self):
        return -1

    def release_mark(self, mark):
        pass


class ArraySink:
    def __init__(self, out):
        self.out, self.puts = out, []

    def put(self, buf, nrows, row0, mark):
        self.out[row0:row0 + nrows] = buf[:nrows]
        self.puts.append((row0, nrows))


class NumpyRank(multi.Rank):
    def engine(self, precision=0, row_standardize=True):
        return ApiNumpyEngine(row_standardize)


def numpy_group(size):
    pipes = {(a, b): queue.Queue() for a in range(size) for b in range(size) if a != b}
    return multi.DeviceGroup(list(range(size)), backend=lambda group, rank: NumpyRank(group, rank, None, QueueComm(group, rank, pipes)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def ragged_seqs(n, seed=5, lo=6, hi=400):
    rng = np.random.default_rng(seed)
    return ["".join(rng.choice(list("ACGT"), size=int(rng.integers(lo, hi)))) for _ in range(n)]


# ------------------------------------------------------------------------------ the group ------
def test_bounds_by_bases_are_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    for n, size in [(0, 3), (1, 4), (5, 8), (100, 3), (1000, 8)]:
        lengths = rng.integers(0, 5000, n)
        b = multi.bounds_by_bases(lengths, size)
        assert len(b) == size + 1 and b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(size))
        if n >= 100:
            per = [int(lengths[b[g]:b[g + 1]].sum()) for g in range(size)]
            assert max(per) - min(per) <= 2 * 5000, per  # within a sequence or two of each other
    skew = np.array([10 ** 6] + [10] * 99)  # one huge transcript: it gets a range of its own, nobody gets a negative one
    b = multi.bounds_by_bases(skew, 4)
    assert b[1] == 1 and b[-1] == 100 and all(b[i] <= b[i + 1] for i in range(4))
    assert multi.bounds_by_bases(np.zeros(7, np.int64), 3) == shard_bounds(7, 3)


def test_requested_devices_parsing(monkeypatch):
    monkeypatch.setattr(multi._lib, "device_count", lambda: 4)
    for spec, want in [("", None), ("2", None), ("0,1", [0, 1]), ("all", [0, 1, 2, 3]), (" 3 , 1 ", [3, 1])]:
        monkeypatch.setenv("SEEKR_DEVICES", spec)
        assert multi.requested_devices() == want
    monkeypatch.delenv("SEEKR_TEST_HOOKS", raising=False)
    for spec in ("0,0", "0,7", "gpu0", "-1,0"):
        monkeypatch.setenv("SEEKR_DEVICES", spec)
        with pytest.raises(ValueError):
            multi.requested_devices()
    monkeypatch.setenv("SEEKR_TEST_HOOKS", "1")  # several ranks on one GPU: tests only
    monkeypatch.setenv("SEEKR_DEVICES", "0,0,0")
    assert multi.requested_devices() == [0, 0, 0]


def test_a_failing_phase_reaches_the_caller_once_and_the_group_lives_on():
    group = numpy_group(4)
    calls = []

    def job(st, spec):
        def local():
            if st.rank in spec:
                raise ZeroDivisionError("division by zero (rank %d)" % st.rank)
            return st.rank
        got = st.phase(local)
        calls.append(st.rank)  # only reached when nobody failed
        return st.comm.allreduce([got], "sum")[0]

    assert group.run(job, ()) == [6.0] * 4
    calls.clear()
    with pytest.raises(ZeroDivisionError, match=r"rank 1"):  # ranks 1 and 3 fail: the lowest rank's exception, itself
        group.run(job, (3, 1))
    assert calls == [] and not group.broken
    assert group.run(job, ()) == [6.0] * 4  # every thread is back in its loop
    group.close()


def test_an_error_outside_a_phase_breaks_the_group_instead_of_hanging():
    group = numpy_group(3)

    def job(st, spec):
        if st.rank == 2:
            raise MemoryError("out of device memory")
        return st.comm.allreduce([1.0], "sum")[0]  # the others are waiting here when rank 2 dies

    with pytest.raises(MemoryError):
        group.run(job, None)
    assert group.broken
    with pytest.raises(multi.GroupBroken):
        group.run(job, None)


# ------------------------------------------------------------------------------ get_counts -----
@pytest.mark.parametrize("size,n", [(2, 40), (3, 41), (8, 5), (4, 1), (5, 64)])
@pytest.mark.parametrize("log2", ["Log2.post", "Log2.pre", "Log2.none"])
def test_counts_job_equals_the_oracle(size, n, log2):
    seqs = ragged_seqs(n)
    lengths = [len(s) for s in seqs]
    std = n > 1
    group = numpy_group(size)
    with np.errstate(all="ignore"):
        counts, mean, sd, has_nan = multi.run_counts(group.run, size, ("strings", seqs), lengths, 3, log2, True, std, "AGTC", True, 64)
        want, wmean, wstd = orc.get_counts(seqs, k=3, log2=log2, std=std)
    group.close()
    assert np.array_equal(bits(mean), bits(wmean))
    if std:
        assert np.array_equal(bits(sd), bits(wstd))
    assert counts.shape == want.shape and np.array_equal(bits(counts), bits(want)) or np.allclose(counts, want, equal_nan=True, rtol=0, atol=0)
    assert has_nan == bool(np.isnan(want).any())


def test_counts_job_with_user_vectors_and_an_empty_range():
    seqs = ragged_seqs(3)  # 3 sequences on 5 ranges: two ranges are empty
    want, wmean, wstd = orc.get_counts(seqs, k=2)
    group = numpy_group(5)
    with np.errstate(all="ignore"):
        counts, mean, sd, _ = multi.run_counts(group.run, 5, ("strings", seqs), [len(s) for s in seqs], 2, "Log2.post", wmean, wstd, "AGTC", True, 16)
        again, _, _ = orc.get_counts(seqs, k=2, mean=wmean, std=wstd)
    group.close()
    assert mean is None and sd is None  # supplied vectors are not returned as computed ones
    assert np.array_equal(bits(counts), bits(again))


def test_a_sequence_of_length_k_minus_1_raises_the_references_exception_once():
    seqs = ragged_seqs(30)
    seqs[17] = "AC"  # k = 3: W = 0 -> ZeroDivisionError (kmer_counts.py:144), in the range of one rank only
    group = numpy_group(4)
    with pytest.raises(ZeroDivisionError):
        multi.run_counts(group.run, 4, ("strings", seqs), [len(s) for s in seqs], 3, "Log2.post", True, True, "AGTC", True, 64)
    assert not group.broken
    seqs[17] = "ACGTAC"
    counts = multi.run_counts(group.run, 4, ("strings", seqs), [len(s) for s in seqs], 3, "Log2.none", False, False, "AGTC", True, 64)[0]
    group.close()
    assert np.array_equal(bits(counts), bits(orc.raw_counts(seqs, 3)))


# ------------------------------------------------------------------------------ pearson --------
def run_pearson_numpy(group, size, c1, c2, f64, row_standardize, stripe):
    same = c2 is None
    out = np.full((len(c1), len(c1 if same else c2)), np.nan, np.float64 if f64 else np.float32)
    sink = ArraySink(out)
    spec = multi.PearsonSpec(c1, c2, shard_bounds(len(c1), size), shard_bounds(out.shape[1], size), f64, False, 0, row_standardize,
                             sink, stripe, out.dtype)
    group.run(multi.pearson_job, spec)
    return out, sink


@pytest.mark.parametrize("size,m,n,stripe", [(2, 37, None, 5), (3, 10, 23, 4), (8, 5, None, None), (4, 33, 2, 3)])
@pytest.mark.parametrize("f64", [False, True])
def test_pearson_job_places_every_stripe(size, m, n, stripe, f64):
    rng = np.random.default_rng(7)
    dt = np.float64 if f64 else np.float32
    c1 = rng.standard_normal((m, 48)).astype(dt)
    c1[m // 2] = 2.0  # a constant row: NaN row (and column, in the self-comparison)
    c2 = None if n is None else rng.standard_normal((n, 48)).astype(dt)
    group = numpy_group(size)
    got, sink = run_pearson_numpy(group, size, c1, c2, f64, True, stripe)
    group.close()
    with np.errstate(all="ignore"):
        want = orc.pearson(c1, c1 if c2 is None else c2)
    assert np.allclose(got, want, rtol=1e-5 if not f64 else 1e-12, atol=1e-6 if not f64 else 1e-12, equal_nan=True)
    rows = sorted(sink.puts)
    assert rows[0][0] == 0 and sum(k for _, k in rows) == m  # every row exactly once
    assert all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
    if stripe:
        assert max(k for _, k in rows) <= stripe


@pytest.mark.parametrize("where", ["upload", "gather_room", "block"])
@pytest.mark.parametrize("f64", [False, True])
def test_pearson_job_one_gpu_out_of_memory_is_agreed_on_before_any_collective(where, f64):
    """a failure of ONE GPU's local work in pearson_job — its upload, the room for the gathered rows, its stripe
    buffers — must reach the caller as itself, once, with every thread back in its loop; outside a phase the other threads
    would be left inside a device collective that has no timeout (over RCCL: a hang with the API lock held)."""
    if where == "upload" and f64:
        where = "rows_f64"
    if where == "gather_room" and not f64:
        where = "empty_operand"
    rng = np.random.default_rng(3)
    c1 = rng.standard_normal((29, 32)).astype(np.float64 if f64 else np.float32)
    failing = {"on": True}

    class Flaky(ApiNumpyEngine):
        rank = None

        def __getattribute__(self, name):
            fn = object.__getattribute__(self, name)
            if name == where and object.__getattribute__(self, "rank") == 2 and failing["on"]:
                def boom(*a, **kw):
                    raise MemoryError("GPU 2 is out of memory in %s" % name)
                return boom
            return fn

    class FlakyRank(multi.Rank):
        def engine(self, precision=0, row_standardize=True):
            e = Flaky(row_standardize)
            e.rank = self.rank
            return e

    pipes = {(a, b): queue.Queue() for a in range(4) for b in range(4) if a != b}
    group = multi.DeviceGroup(list(range(4)), backend=lambda grp, rank: FlakyRank(grp, rank, None, QueueComm(grp, rank, pipes)))
    out = np.full((29, 29), np.nan, c1.dtype)
    spec = multi.PearsonSpec(c1, None, shard_bounds(29, 4), shard_bounds(29, 4), f64, False, 0, True, ArraySink(out), 4, out.dtype)
    with pytest.raises(MemoryError, match="GPU 2 is out of memory"):
        group.run(multi.pearson_job, spec)
    assert not group.broken
    failing["on"] = False
    group.run(multi.pearson_job, spec)  # the same group, the same job: every thread was back in its loop
    group.close()
    with np.errstate(all="ignore"):
        assert np.allclose(out, orc.pearson(c1, c1), rtol=1e-5, atol=1e-6)


def test_stripe_height():
    assert multi.stripe_height(1000, 1000, 4, 1 << 40) == 1000                  # fits: one stripe
    h = multi.stripe_height(200_000, 200_000, 4, 250 << 30)
    assert h % 256 == 0 and 2 * h * 200_000 * 4 <= 2 * multi.STRIPE_TARGET_BYTES
    assert multi.stripe_height(200_000, 200_000, 4, 1 << 30) % 256 == 0          # little memory: still whole tiles
    assert multi.stripe_height(200_000, 200_000, 4, 1 << 30) * 200_000 * 4 <= 0.4 * (1 << 30) + 256 * 800_000
    assert multi.stripe_height(50, 10, 4, 1 << 40, forced=7) == 7
    assert multi.stripe_height(0, 10, 4, 1 << 40) == 1


# ------------------------------------------------------------------ the environment RCCL's set-up needs, in time ----
IPC_CHILD = r'''
import ctypes, json, os, sys
sys.path.insert(0, sys.argv[1])
seen = {}
real_cdll = ctypes.CDLL
def spy(path, *a, **kw):  # what stands in the environment at the moment the HIP library enters the process
    if "libseekr_hip" in str(path):
        seen["at_cdll"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    return real_cdll(path, *a, **kw)
ctypes.CDLL = spy
late = sys.argv[2] == "late"
import seekr_amd
from seekr_amd import _lib, multi
seen["after_import"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
if late:
    _lib.lib()                              # the package is used first ...
    os.environ["SEEKR_DEVICES"] = "0,1"     # ... and SEEKR_DEVICES set afterwards, as multi.py's docstring once showed
_lib.lib()
seen["at_load_recorded"] = _lib.ipc_env_at_load()
seen["late_note"] = multi.late_ipc_note()
seen["names_several"] = _lib.names_several_devices()
print(json.dumps(seen))
'''


def _ipc_child(mode, **env):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = {k: v for k, v in os.environ.items() if k not in ("SEEKR_DEVICES", "HSA_ENABLE_IPC_MODE_LEGACY")}
    full.update(env)
    res = subprocess.run([sys.executable, "-c", IPC_CHILD, root, mode], env=full, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_ipc_mode_is_in_the_environment_before_the_hip_library_is_loaded():
    """RCCL's peer-to-peer set-up between the GPU threads needs HSA_ENABLE_IPC_MODE_LEGACY=0, and the
    HIP / HSA runtime reads its environment when it starts.  With SEEKR_DEVICES naming several devices the package sets the
    variable at import — before ctypes.CDLL(libseekr_hip.so), spied on here — and leaves a value the user exported alone;
    with one device (or none) it touches nothing."""
    for spec in ("0,1", "all", " 3 , 5 ,"):
        got = _ipc_child("early", SEEKR_DEVICES=spec)
        assert got["after_import"] == "0" and got["at_cdll"] == "0" and got["at_load_recorded"] == "0", (spec, got)
        assert got["late_note"] == "" and got["names_several"] is True
    got = _ipc_child("early", SEEKR_DEVICES="0,1", HSA_ENABLE_IPC_MODE_LEGACY="1")  # the user's own value stands
    assert got["at_cdll"] == "1" and "was '1'" in got["late_note"]
    for spec in ("", "0", "2,"):
        got = _ipc_child("early", SEEKR_DEVICES=spec)
        assert got["after_import"] is None and got["at_cdll"] is None and got["names_several"] is False, (spec, got)


def test_seekr_devices_set_after_the_first_use_is_named_as_the_cause():
    """SEEKR_DEVICES put into os.environ after the library is in the process: too late for the variable, and the one-line
    notice / the exception of group_for says so instead of blaming RCCL."""
    got = _ipc_child("late")
    assert got["at_cdll"] is None and got["at_load_recorded"] is None and got["names_several"] is True
    assert "SEEKR_DEVICES was first seen after" in got["late_note"] and "HSA_ENABLE_IPC_MODE_LEGACY" in got["late_note"]


def test_a_failed_pearson_to_file_leaves_no_file_of_zeros_and_keeps_the_old_one(tmp_path, monkeypatch):
    """np.save(outfile, dist) only touches the file once dist exists (pearson.py:41-43).  The stripe writer lays the file
    out first — under a temporary name: a job that fails leaves the previous file as it was and nothing else behind; a job
    that succeeds renames."""
    out = str(tmp_path / "r.npy")
    np.save(out, np.arange(6.0))
    c1 = np.ones((5, 8), np.float32)
    monkeypatch.setattr(multi._lib, "default_context", lambda: None)

    def boom(st, spec):
        assert os.path.exists(spec.sink.path) and spec.sink.path != out  # the stripes' file exists, under another name
        raise MemoryError("out of device memory")
    monkeypatch.setattr(multi, "pearson_job", boom)
    with pytest.raises(MemoryError):
        multi.run_pearson(c1, None, np.float32, np.float32, True, 4, None, outfile=out)
    assert sorted(os.listdir(str(tmp_path))) == ["r.npy"] and np.array_equal(np.load(out), np.arange(6.0))
    monkeypatch.setattr(multi, "pearson_job", lambda st, spec: None)  # "succeeds" (writes no stripe: the laid-out zeros stay)
    multi.run_pearson(c1, None, np.float32, np.float32, True, 4, None, outfile=str(tmp_path / "r"))  # np.save appends .npy
    got = np.load(out)
    assert sorted(os.listdir(str(tmp_path))) == ["r.npy"] and got.shape == (5, 5) and got.dtype == np.float32
