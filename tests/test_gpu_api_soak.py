"""An API-level resource soak: 2 000 alternating BasicCounter.get_counts() / pearson() /
consumers.pearson_edges() calls of varying shapes and dtypes in ONE process, an exception path every 50th call — and at the
end the GPU's free memory is what it was (plus at most the ctx's grown workspaces) and the process holds no more file
descriptors.  Then the same from two Python threads sharing the default context.

What the build guarantees about re-entrancy (the reference is "not re-entrant but no globals"): the library
keeps no global mutable state but the lazily loaded RCCL entry points (behind a mutex) and a thread-local error string; a
skr_ctx is ONE stream and its scratch, and calls through one ctx from several threads are NOT safe against each other at the
library level — the Python layer therefore serialises the API entry points that share the default context on a lock
(seekr_amd._lib.API_LOCK), which is what the two-thread half of this test exercises.  Different contexts (SEEKR_DEVICES: one
per GPU thread) run concurrently.  Needs a real MI355X: run with `-m gpu`."""
import contextlib
import io
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def n_fds():
    return len(os.listdir("/proc/self/fd"))


def one_call(i, rng, seqs_pool):
    """Call number i of the soak; returns a small checksum so that results can be compared between runs."""
    from seekr_amd import consumers
    from seekr_amd.kmer_counts import BasicCounter
    from seekr_amd.pearson import pearson
    kind = i % 5
    if i % 50 == 49:  # an exception path: each must leave nothing behind
        which = (i // 50) % 3
        if which == 0:
            c = BasicCounter(k=4, silent=True)
            c.seqs = ["ACGTACGTAC", "ACG", "ACGTTTGACA"]  # len == k - 1 (kmer_counts.py:144)
            with pytest.raises(ZeroDivisionError):
                c.get_counts()
        elif which == 1:
            with pytest.raises(ValueError, match="not aligned"):
                pearson(np.ones((3, 16), np.float32), np.ones((3, 17), np.float32))
        else:
            x = rng.standard_normal((6, 64)).astype(np.float32)
            x[2] = 1.0  # a constant row: NaN row and column, no exception
            r = pearson(x, x)
            assert np.isnan(r[2]).all() and np.isfinite(r[0, 1])
        return 0.0
    if kind in (0, 1):
        k = int(rng.integers(1, 7))
        n = int(rng.integers(2, 60))
        seqs = [seqs_pool[int(j)] for j in rng.integers(0, len(seqs_pool), n)]
        c = BasicCounter(k=k, silent=True, log2=["Log2.post", "Log2.pre", "Log2.none"][i % 3], mean=bool(i % 7), std=bool(i % 11),
                         alphabet="AGTC" if kind == 0 else "ACGTN"[:int(rng.integers(2, 6))] if k <= 4 else "AGTC")
        c.seqs = seqs
        with contextlib.redirect_stdout(io.StringIO()):
            c.get_counts()
        return float(np.nan_to_num(c.counts).sum())
    if kind in (2, 3):
        cols = int(rng.choice([16, 64, 100, 256, 1000, 4096]))
        m, n = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        dt = [np.float32, np.float32, np.float64, np.int64][i % 4]
        a = (rng.standard_normal((m, cols)) * 3).astype(dt)
        b = a if i % 3 == 0 else (rng.standard_normal((n, cols)) * 3).astype(dt)
        return float(np.nan_to_num(pearson(a, b, row_standardize=bool(i % 13))).sum())
    from seekr_amd import _lib
    x = rng.standard_normal((int(rng.integers(20, 300)), 256)).astype(np.float32)
    stripe = int(rng.choice([16, 64, 512]))
    with _lib.API_LOCK:  # device handles on the shared default context: the caller's own critical section
        ctx = _lib.default_context()
        op, _ = _lib.operand_fill(ctx, ctx.from_numpy(x))
        rows, cols, vals = consumers.pearson_edges(op, 0.1, stripe_rows=stripe)
        op.free()
    return float(vals.sum()) + len(rows)


@pytest.fixture(scope="module")
def seqs_pool():
    rng = np.random.default_rng(0)
    return ["".join(rng.choice(list("ACGT"), size=int(rng.integers(8, 700)))) for _ in range(300)]


def test_two_thousand_api_calls_leave_nothing_behind(seqs_pool):
    from seekr_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(1)
    for i in range(200):  # warm-up: the ctx's workspaces, the LDS attributes, the kernels' code objects
        one_call(i, rng, seqs_pool)
    import gc
    gc.collect()
    ctx.sync()
    free0, fds0 = ctx.mem_info()[0], n_fds()
    sums = [one_call(i, rng, seqs_pool) for i in range(2000)]
    gc.collect()
    ctx.sync()
    free1, fds1 = ctx.mem_info()[0], n_fds()
    assert np.isfinite(sums).all()
    assert fds1 <= fds0, (fds0, fds1)
    assert free0 - free1 <= 64 << 20, "device memory shrank by %.1f MiB over 2 000 calls" % ((free0 - free1) / 2 ** 20)
    # the same 2 000 calls again: the same results (no state carried from call to call), still nothing lost
    rng = np.random.default_rng(1)
    for i in range(200):
        one_call(i, rng, seqs_pool)
    again = [one_call(i, rng, seqs_pool) for i in range(2000)]
    assert sums == again
    gc.collect()
    ctx.sync()
    assert ctx.mem_info()[0] >= free1 - (16 << 20) and n_fds() <= fds0


def test_two_threads_share_the_default_context(seqs_pool):
    from seekr_amd import _lib
    ctx = _lib.default_context()
    want = {}
    for t in range(2):
        rng = np.random.default_rng(100 + t)
        want[t] = [one_call(i, rng, seqs_pool) for i in range(300)]
    got, errors = {}, []

    def work(t):
        try:
            rng = np.random.default_rng(100 + t)
            got[t] = [one_call(i, rng, seqs_pool) for i in range(300)]
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    ctx.sync()
    free0 = ctx.mem_info()[0]
    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
    assert not errors, errors[:1]
    assert got == want  # every call computed what it computes alone
    import gc
    gc.collect()
    ctx.sync()
    assert free0 - ctx.mem_info()[0] <= 64 << 20
