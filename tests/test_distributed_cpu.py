"""world_size 2 and 3 runs of the row-sharded pipeline on CPU (gloo): the sharded result must be
bit-identical to the single-process oracle for everything that is IEEE arithmetic (column
mean/std chained across ranks in row order, normalised counts) and allclose for Pearson r."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from oracle import seekr_oracle as orc
import dist_worker


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(size, n_rows, n_cols, log2, out_dir, with_nan=False):
    ctx = mp.get_context("spawn")
    port = free_port()
    procs = [ctx.Process(target=dist_worker.run, args=(rank, size, port, n_rows, n_cols, log2, str(out_dir), with_nan))
             for rank in range(size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "rank exited with {}".format(p.exitcode)
    return [np.load(os.path.join(str(out_dir), "rank%d.npz" % r)) for r in range(size)]


def reference(n_rows, n_cols, log2, with_nan):
    rng = np.random.default_rng(1234)
    full = (rng.binomial(60, 0.04, size=(n_rows, n_cols)) * np.float32(1000 / 595)).astype(np.float32)
    if with_nan:
        full[:, 3] = 0.0
    with np.errstate(all="ignore"):
        x, mean, std = orc.normalize(full, log2=log2)
        r = orc.pearson(x, x)
    return x, mean, std, r


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("size,n_rows,log2", [(2, 301, "Log2.post"), (3, 500, "Log2.none"), (2, 64, "Log2.post")])
def test_sharded_pipeline_matches_single_process(size, n_rows, log2, tmp_path):
    n_cols = 64
    parts = launch(size, n_rows, n_cols, log2, tmp_path)
    x, mean, std, r = reference(n_rows, n_cols, log2, False)
    assert sum(int(p["hi"]) - int(p["lo"]) for p in parts) == n_rows
    for p in parts:
        lo, hi = int(p["lo"]), int(p["hi"])
        assert np.array_equal(bits(p["mean"]), bits(mean))  # the chain crossed ranks in row order
        assert np.array_equal(bits(p["std"]), bits(std))
        assert np.array_equal(bits(p["x"]), bits(x[lo:hi]))
        assert not bool(p["has_nan"])
        assert np.allclose(p["r"], r[lo:hi], rtol=1e-5, atol=2e-6)
    full_r = np.concatenate([p["r"] for p in parts], axis=0)
    assert full_r.shape == (n_rows, n_rows)


def test_sharded_nan_propagation(tmp_path):
    parts = launch(2, 120, 32, "Log2.post", tmp_path, with_nan=True)
    for p in parts:
        assert bool(p["has_nan"])  # every rank learns about the NaN even if its own shard...
        assert np.isnan(p["x"]).all()  # ...and np.min's NaN propagates to the whole matrix (:208)


def test_shard_bounds():
    from seekr_amd.distributed import shard_bounds
    assert shard_bounds(10, 3) == [0, 4, 7, 10]
    assert shard_bounds(8, 8) == list(range(9))
    assert shard_bounds(5, 8)[-1] == 5 and len(shard_bounds(5, 8)) == 9
