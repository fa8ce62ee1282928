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
import dist_worker_x8


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


def assemble_symmetric(parts, n_rows):
    """Put every rank's owned blocks of the symmetric layout into one N x N matrix; `hits`
    counts how many ranks claimed each cell."""
    full = np.zeros((n_rows, n_rows), np.float32)
    hits = np.zeros((n_rows, n_rows), np.int32)
    for p in parts:
        for which, br, bc, nr, nc, gr, gc in p["blocks"]:
            buf = p["r_row"] if which == 0 else p["r_col"]
            full[gr:gr + nr, gc:gc + nc] = buf[br:br + nr, bc:bc + nc]
            hits[gr:gr + nr, gc:gc + nc] += 1
    return full, hits


@pytest.mark.parametrize("size,n_rows,log2", [(2, 301, "Log2.post"), (3, 500, "Log2.none"), (2, 64, "Log2.post"),
                                              (4, 203, "Log2.none")])
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
        assert np.allclose(p["r_ag"], p["r"], rtol=1e-6, atol=1e-6)   # the all-gather schedule: the same row block
    full_r = np.concatenate([p["r"] for p in parts], axis=0)
    assert full_r.shape == (n_rows, n_rows)
    # symmetric (half-ring) layout: every ordered pair lives on exactly one rank, same values
    sym, hits = assemble_symmetric(parts, n_rows)
    assert (hits == 1).all()
    assert np.array_equal(bits(sym), bits(full_r))       # numpy engine: same arithmetic either way
    assert np.array_equal(bits(sym), bits(sym.T.copy()))
    owned = sum(int(nr) * int(nc) for p in parts for _, _, _, nr, nc, _, _ in p["blocks"])
    assert owned == n_rows * n_rows
    # striped edge lists: the union over the ranks is the thresholded, zero-diagonal matrix's non-zeros
    want = full_r.copy()
    want[want < 0.05] = 0
    np.fill_diagonal(want, 0)
    for key, ref in (("e_all", want), ("e_up", np.triu(want, 1))):
        got = np.zeros_like(want)
        n_edges = 0
        for p in parts:
            i, j, v = p[key + "_i"], p[key + "_j"], p[key + "_v"]
            assert not got[i, j].any()  # no cell reported twice
            got[i, j] = v
            n_edges += len(i)
        assert n_edges > 0
        # BLAS results differ in the last bit between a stripe and the full matrix: compare away from
        # the cutoff (the GPU test compares bit for bit: its contraction does not depend on the tiling)
        clear = np.abs(full_r - 0.05) > 1e-5
        assert np.array_equal(got[clear] != 0, ref[clear] != 0)
        assert np.allclose(got[clear], ref[clear], rtol=1e-5, atol=1e-6)


def test_sharded_nan_propagation(tmp_path):
    parts = launch(2, 120, 32, "Log2.post", tmp_path, with_nan=True)
    for p in parts:
        assert bool(p["has_nan"])  # every rank learns about the NaN even if its own shard...
        assert np.isnan(p["x"]).all()  # ...and np.min's NaN propagates to the whole matrix (:208)


@pytest.mark.parametrize("size", list(range(1, 10)))
def test_half_ring_plan_tiles_the_matrix_once(size):
    from seekr_amd.distributed import half_ring_plan, owned_blocks, shard_bounds
    for n_rows in (size, 7 * size + 3, 101):
        bounds = shard_bounds(n_rows, size)
        hits = np.zeros((n_rows, n_rows), np.int32)
        work = []
        for rank in range(size):
            for _, br, bc, nr, nc, gr, gc in owned_blocks(size, rank, bounds):
                hits[gr:gr + nr, gc:gc + nc] += 1
            n_g = bounds[rank + 1] - bounds[rank]
            work.append(n_g * n_g / 2 + sum(an * bn for _, _, _, an, _, bn in half_ring_plan(size, rank, bounds)))
            # shift s pairs rank with rank+s: the peer's plan must name the matching send
            for s, peer, *_ in half_ring_plan(size, rank, bounds):
                assert peer == (rank + s) % size
        assert (hits == 1).all(), (size, n_rows)
        if n_rows >= 7 * size:  # multiplications are balanced to within the raggedness of the shards
            assert max(work) <= 1.35 * min(work), (size, n_rows, work)
        assert len({len(half_ring_plan(size, r, bounds)) for r in range(size)}) == 1  # same number of shifts


def test_stripes_are_dealt_once_and_balanced():
    from seekr_amd.distributed import stripes_of_rank
    for size in (1, 2, 3, 8):
        for n, stripe in ((1000, 64), (8192 * 9 + 5, 8192), (10, 64)):
            seen, work = [], []
            for rank in range(size):
                mine = stripes_of_rank(n, stripe, size, rank)
                seen += mine
                work.append(sum((s1 - s0) * (n - s0) for s0, s1 in mine))  # upper-only cells
            assert sorted(seen) == [(s0, min(n, s0 + stripe)) for s0 in range(0, n, stripe)]
            if n // stripe >= 4 * size:
                assert max(work) - min(work) <= stripe * n


def test_shard_bounds():
    from seekr_amd.distributed import shard_bounds
    assert shard_bounds(10, 3) == [0, 4, 7, 10]
    assert shard_bounds(8, 8) == list(range(9))
    assert shard_bounds(5, 8)[-1] == 5 and len(shard_bounds(5, 8)) == 9


@pytest.mark.parametrize("scenario,size", [("keep", 2), ("pair", 2), ("local", 2), ("pair", 3)])
def test_f16f8_routing_is_decided_on_global_maxima(scenario, size, tmp_path):
    """The opt-in f16f8 layout across ranks (round 4): the three row-mean maxima of every shard ride on the verdict
    all-reduce of the normalisation, the rule is applied to the GLOBAL values — two shards that pass it on their own may
    fail it together (the two-operand case of tests/golden/regress_r4_f16f8_two_operands.npz, here as two shards) — and
    all ranks keep the layout, with the global maxima set on every shard, or all route back."""
    from seekr_amd import _lib
    ctx = mp.get_context("spawn")
    port = free_port()
    procs = [ctx.Process(target=dist_worker_x8.run, args=(rank, size, port, str(tmp_path), scenario)) for rank in range(size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "rank exited with {}".format(p.exitcode)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(size)]
    if scenario == "keep":
        assert all(int(p["kind"]) == 3 and not bool(p["refilled"]) for p in parts)
        want = (1.0, 0.006, 0.0002)   # the element-wise maxima over the shards
        assert all(np.allclose(p["stats"], want) for p in parts)
    elif scenario == "pair":
        assert all(float(p["own_bound"]) <= dist_worker_x8.X8_MEANS_LIMIT for p in parts)     # every shard passes on its own
        assert all(int(p["kind"]) == 2 and bool(p["refilled"]) for p in parts)     # ... and none keeps the layout
    else:
        assert [int(p["kind"]) for p in parts] == [2, 2]
        assert [bool(p["refilled"]) for p in parts] == [False, True]               # rank 0's fill had routed it already

