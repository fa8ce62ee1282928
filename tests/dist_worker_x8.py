"""One rank of the gloo test of the opt-in f16f8 routing ACROSS ranks (seekr_amd.distributed.sharded_normalize_prepare):
the numpy engine of dist_worker.py dressed as an engine of precision f16f8 whose operands carry the layout kind and the
three row-mean maxima the HIP fill would report (skr_operand_x8_stats) — here taken from the environment of the test, so
that the plumbing (maxima all-reduced with the step's verdicts, rule applied to the GLOBAL values, every rank routing back
together or every rank keeping the layout with the global maxima set) is checked without a GPU."""
import math
import os

import numpy as np

from dist_worker import GlooComm, NumpyEngine


X8_MEANS_LIMIT = 0.6 * 2e-6  # kX8MeansLimit


class FakeOperand:
    def __init__(self, z, kind, stats):
        self.z, self.kind, self.coherent, self._stats = z, kind, False, tuple(stats)
        self.cols = z.shape[1]

    @property
    def x8_stats(self):
        return self._stats

    @x8_stats.setter
    def x8_stats(self, v):
        self._stats = tuple(float(t) for t in v)

    def x8_pair_bound(self, other=None):
        # the library's rule (skr_operand_x8_pair_bound, common.hpp: skr_x8_pair_bound / kX8MeansLimit) restated for the fake
        d, l, dl = self._stats
        s = 2.0 ** math.floor(math.log2(32768.0 / math.sqrt(4096)))  # the rule is tested at the k = 6 scale whatever `cols` is
        bound = 2.0 * (d * (l + dl) + (d + l) * dl) / (s * s)
        return bound, bound <= X8_MEANS_LIMIT


class X8Engine(NumpyEngine):
    def __init__(self, stats, kind):
        from seekr_amd import _lib
        self.precision = _lib.PREC_F16F8
        self._stats, self._kind = stats, kind
        self.refilled = False

    def layout(self, op):
        return op.kind

    def prepare(self, x, center=None, scale=None, post=False, shift=0.0, keep_counts=True, op=None):
        z, has_nan = NumpyEngine.prepare(self, x, center, scale, post, shift, keep_counts, None)
        return FakeOperand(z, self._kind, self._stats if self._kind == 3 else (0.0, 0.0, 0.0)), has_nan

    def prepare_f16x3(self, x, op=None):
        self.refilled = True
        return FakeOperand(self.row_standardize(x), 2, (0.0, 0.0, 0.0))


def run(rank, size, port, out_dir, scenario):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from seekr_amd.distributed import shard_bounds, sharded_normalize_prepare
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        rng = np.random.default_rng(99)
        n_rows, n_cols = 120, 64
        full = (rng.binomial(60, 0.04, size=(n_rows, n_cols)) * np.float32(1000 / 595)).astype(np.float32)
        bounds = shard_bounds(n_rows, size)
        x = full[bounds[rank]:bounds[rank + 1]].copy()
        # (max |mean(hi - 128 h8)|, max |mean lo|, max |mean(lo - l8 / 16)|) per shard, kind the local fill ended with
        table = {
            # every shard passes on its own AND together: the layout is kept, every shard ends with the global maxima
            "keep": [((1.0, 0.004, 0.0001), 3), ((0.6, 0.006, 0.0002), 3)],
            # each passes on its own (bounds 0.29 of the bar: the two operands of tests/golden/regress_r4_f16f8_two_operands),
            # together mean(hi - 128 h8) of one meets mean(lo) of the other: every rank routes back
            "pair": [((0.548 * 2, 0.0346 * 2, 0.00002), 3), ((5.287 * 2, 0.0035 * 2, 0.00002), 3)],
            # one rank's fill already routed its rows back: the others follow (the pre-existing flag)
            "local": [((1.0, 0.004, 0.0001), 2), ((0.6, 0.006, 0.0002), 3)],
        }[scenario]
        stats, kind = table[rank % 2]
        engine, comm = X8Engine(stats, kind), GlooComm(dist, torch)
        mean, std, has_nan, z = sharded_normalize_prepare(engine, comm, x, n_rows, "Log2.post", True, True)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), kind=np.array(z.kind), stats=np.array(z.x8_stats),
                 refilled=np.array(engine.refilled), own_bound=np.array(FakeOperand(z.z, 3, stats).x8_pair_bound()[0]))
        comm.barrier()
    finally:
        dist.destroy_process_group()
