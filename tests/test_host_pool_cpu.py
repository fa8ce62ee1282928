"""The pool behind the arrays the API returns (seekr_amd._lib.HostPool): what the caller holds is an ordinary ndarray; its
memory goes back to the pool — not to the operating system — when the array and all its views are gone, and the next result
of about that size lands in the same (touched, on a GPU box: registered) pages.  The reference returns plain numpy arrays
(pearson.py:41-44, kmer_counts.py:196): everything a caller may do with one must work with these."""
import gc
import pickle
import threading

import numpy as np
import pytest

from seekr_amd import _lib


@pytest.fixture()
def pool(monkeypatch):
    monkeypatch.setenv("SEEKR_RESULT_POOL_MB", "64")
    p = _lib.HostPool()
    yield p
    p.clear()


def test_a_result_is_an_ordinary_array_and_its_pages_come_back(pool):
    a = pool.empty((600, 1000), np.float32)
    assert type(a) is np.ndarray and a.shape == (600, 1000) and a.dtype == np.float32 and a.flags.c_contiguous and a.flags.writeable
    a[:] = 3.5
    where = a.ctypes.data
    view, col = a[10:20], a[:, 3]
    del a
    gc.collect()
    assert pool.stats["kept_bytes"] == 0 and float(view[0, 0]) == 3.5  # views keep the memory leased
    del view, col
    gc.collect()
    assert pool.stats["kept_bytes"] == 600 * 1000 * 4
    b = pool.empty((500, 1100), np.float32)  # about the size: the same pages
    assert b.ctypes.data == where and pool.stats["reused"] == 1 and pool.stats["kept_bytes"] == 0
    c = pool.empty((600, 1000), np.float64)  # twice the size: its own allocation
    assert pool.stats["fresh"] == 2
    # what callers do with a result
    b[:] = 1.0
    assert np.array_equal(pickle.loads(pickle.dumps(b)), b) and b.copy().flags.owndata and (b @ b.T).shape == (500, 500)
    assert np.array_equal(np.asarray(b), b) and b.T.shape == (1100, 500) and b.astype(np.float64).dtype == np.float64
    b -= b.mean(axis=0)  # in place, like BasicCounter.center on a returned matrix
    del b, c
    gc.collect()
    assert pool.stats["kept_bytes"] == 600 * 1000 * 4 + 600 * 1000 * 8


def test_small_and_oversized_results_and_a_switched_off_pool_are_plain_arrays(pool, monkeypatch):
    assert pool.empty((10, 10), np.float32).flags.owndata            # below 1 MiB
    assert pool.empty((5000, 5000), np.float32).flags.owndata        # 100 MB > the 64 MB cap of this test
    assert pool.empty((0, 4096), np.float32).shape == (0, 4096)
    monkeypatch.setenv("SEEKR_RESULT_POOL_MB", "0")
    off = _lib.HostPool()
    assert off.empty((600, 1000), np.float32).flags.owndata and off.stats["fresh"] == 0


def test_the_cap_bounds_what_is_kept(pool):
    held = [pool.empty((1000, 4000), np.float32) for _ in range(6)]  # 6 x 16 MB leased at once
    del held
    gc.collect()
    assert pool.stats["kept_bytes"] <= 64 << 20 and pool.stats["dropped"] == 2  # 4 kept, 2 handed back to the system
    for _ in range(50):  # a loop of calls neither grows the pool nor allocates again
        pool.empty((1000, 4000), np.float32)
    assert pool.stats["fresh"] == 6 and pool.stats["kept_bytes"] <= 64 << 20


def test_leases_may_die_on_any_thread(pool):
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(200):
                a = pool.empty((int(rng.integers(300, 700)), 1000), np.float32)
                a[0, 0] = seed
                assert a[0, 0] == seed
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    gc.collect()
    assert not errors and pool.stats["kept_bytes"] <= 64 << 20
