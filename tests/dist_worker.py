"""Worker for the world_size>1 CPU tests: runs seekr_amd.distributed's orchestration with a numpy
engine (arithmetic = the oracle's) and a gloo communicator, so the sharding logic — row ranges,
the rank-to-rank float32 sum chain, NaN-propagating min, shift schedule and block placement —
is exercised without a GPU.  The production run swaps in HipEngine + RcclComm."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


class NumpyEngine:
    def zeros_vec(self, n):
        return np.zeros(n, dtype=np.float32)

    def vec_to_host(self, v):
        return v

    def cols(self, x):
        return x.shape[1]

    def rows(self, x):
        return x.shape[0]

    def view(self, x, row0, nrows):
        return x[row0:row0 + nrows]

    def colsum(self, x, acc, center=None, center2=None, square=False):
        for i in range(x.shape[0]):
            t = x[i]
            if center is not None:
                t = (t - center).astype(np.float32)
            if square:
                d = (t - center2).astype(np.float32) if center2 is not None else t
                t = (d * d).astype(np.float32)
            acc += t  # float32 += float32: one rounding per row, in row order

    def finish(self, v, n, take_sqrt=False):
        v /= np.float32(n)
        if take_sqrt:
            np.sqrt(v, out=v)

    def _z(self, x, center, scale):
        z = x
        with np.errstate(all="ignore"):
            if center is not None:
                z = (z - center).astype(np.float32)
            if scale is not None:
                z = (z / scale).astype(np.float32)
        return z

    def min_nan(self, x, center, scale):
        z = self._z(x, center, scale)
        if z.size == 0:  # a rank without rows: the identity of the minimum
            return np.float32(np.inf), False
        return np.float32(np.min(z)), bool(np.isnan(z).any())

    def apply(self, x, center, scale, post, shift):
        z = self._z(x, center, scale)
        has_nan = bool(np.isnan(z).any()) if scale is not None else False
        if post:
            with np.errstate(all="ignore"):
                z = (z + np.float32(shift)).astype(np.float32)
                z = (z + np.float32(1)).astype(np.float32)
                z = np.log2(z)
        x[...] = z
        return has_nan

    def row_standardize(self, x, z=None):
        with np.errstate(all="ignore"):
            c = (x.T - np.mean(x, axis=1)).T
            out = (c.T / np.std(c, axis=1)).T
        if z is not None:
            z[...] = out
            return z
        return out

    def empty_operand(self, rows, cols):
        return np.zeros((rows, cols), dtype=np.float32)

    def prepare(self, x, center=None, scale=None, post=False, shift=0.0, keep_counts=True, op=None):
        y = x if keep_counts else x.copy()
        has_nan = self.apply(y, center, scale, post, shift)
        return self.row_standardize(y, op), has_nan

    def gemm(self, a, b, r, col0, symmetric=False, lower=False):
        r[:a.shape[0], col0:col0 + b.shape[0]] = np.inner(a, b) / a.shape[1]

    def empty_block(self, rows, cols):
        return np.zeros((rows, cols), dtype=np.float32)

    def edges(self, r, cutoff, nrows, col_begin, col_end, row_global0, upper_only):
        blk = r[:nrows, col_begin:col_end]
        gi = row_global0 + np.arange(nrows)[:, None]
        gj = col_begin + np.arange(col_end - col_begin)[None, :]
        with np.errstate(invalid="ignore"):
            keep = ~(blk < cutoff) & (blk != 0) & ((gj > gi) if upper_only else (gj != gi))
        i, j = np.nonzero(keep)
        return (row_global0 + i).astype(np.uint32), (col_begin + j).astype(np.uint32), blk[i, j]

    def gemm_mirror(self, a, b, r, row0, col0, rt, trow0, tcol0):
        blk = (np.inner(a, b) / a.shape[1]).astype(np.float32)
        r[row0:row0 + a.shape[0], col0:col0 + b.shape[0]] = blk
        rt[trow0:trow0 + b.shape[0], tcol0:tcol0 + a.shape[0]] = blk.T


class GlooComm:
    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self._pending = {}
        self._next = 0

    def send_vec(self, v, dst):
        self.dist.send(self.torch.from_numpy(v), dst)

    def recv_vec(self, v, src):
        self.dist.recv(self.torch.from_numpy(v), src)

    def allreduce(self, values, op):
        t = self.torch.tensor(list(values), dtype=self.torch.float64)
        ops = {"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN}
        self.dist.all_reduce(t, op=ops[op])
        return t.tolist()

    def shift(self, send, dst, recv, recv_rows, src):
        works = [self.dist.isend(self.torch.from_numpy(np.ascontiguousarray(send)), dst),
                 self.dist.irecv(self.torch.from_numpy(recv[:recv_rows]), src)]
        self._next += 1
        self._pending[self._next] = works
        return self._next

    def shift_part(self, send, srow0, snrows, dst, recv, drow0, dnrows, src):
        works = []
        if snrows:
            works.append(self.dist.isend(self.torch.from_numpy(np.ascontiguousarray(send[srow0:srow0 + snrows])), dst))
        if dnrows:
            works.append(self.dist.irecv(self.torch.from_numpy(recv[drow0:drow0 + dnrows]), src))
        self._next += 1
        self._pending[self._next] = works
        return self._next

    def allgather_rows(self, shard, full, bounds):
        full[bounds[self.rank]:bounds[self.rank + 1]] = shard
        works = []
        for s in range(1, self.size):
            dst, src = (self.rank - s) % self.size, (self.rank + s) % self.size
            if shard.shape[0]:
                works.append(self.dist.isend(self.torch.from_numpy(np.ascontiguousarray(shard)), dst))
            if bounds[src + 1] > bounds[src]:
                works.append(self.dist.irecv(self.torch.from_numpy(full[bounds[src]:bounds[src + 1]]), src))
        self._next += 1
        self._pending[self._next] = works
        return self._next

    def wait(self, ticket):
        for w in self._pending.pop(ticket):
            w.wait()

    def barrier(self):
        self.dist.barrier()


def run(rank, size, port, n_rows, n_cols, log2, out_dir, with_nan):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from seekr_amd.distributed import (shard_bounds, sharded_normalize, sharded_normalize_prepare,
                                       sharded_pearson_allgather, sharded_pearson_edges, sharded_pearson_rowblock,
                                       sharded_pearson_symmetric)

    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        rng = np.random.default_rng(1234)  # every rank builds the same full matrix, keeps its shard
        full = (rng.binomial(60, 0.04, size=(n_rows, n_cols)) * np.float32(1000 / 595)).astype(np.float32)
        if with_nan:
            full[:, 3] = 0.0  # zero-variance column -> 0/0 -> NaN everywhere after Log2.post
        bounds = shard_bounds(n_rows, size)
        lo, hi = bounds[rank], bounds[rank + 1]
        x = full[lo:hi].copy()
        engine, comm = NumpyEngine(), GlooComm(dist, torch)
        if rank % 2 == 0 or size == 2:  # both entry points must agree: fused on some ranks ...
            mean, std, has_nan, z = sharded_normalize_prepare(engine, comm, x, n_rows, log2, True, True)
        else:  # ... separate normalise + standardise on the others
            mean, std, has_nan = sharded_normalize(engine, comm, x, n_rows, log2, True, True)
            z = engine.row_standardize(x)
        r = np.zeros((hi - lo, n_rows), dtype=np.float32)
        max_shard = max(bounds[g + 1] - bounds[g] for g in range(size))
        recv = [np.zeros((max_shard, n_cols), np.float32), np.zeros((max_shard, n_cols), np.float32)]
        with np.errstate(all="ignore"):
            sharded_pearson_rowblock(engine, comm, z, bounds, r, recv)
        r_ag = np.zeros((hi - lo, n_rows), dtype=np.float32)
        with np.errstate(all="ignore"):
            sharded_pearson_allgather(engine, comm, z, bounds, r_ag)
        # the symmetric layout: NaN-filled buffers so that unowned cells are recognisable
        r_row = np.full((hi - lo, n_rows), np.float32(-7.0))
        r_col = np.full((n_rows, hi - lo), np.float32(-7.0))
        with np.errstate(all="ignore"):
            blocks = sharded_pearson_symmetric(engine, comm, z, bounds, r_row, r_col, recv)
        cutoff = 0.05
        with np.errstate(all="ignore"):
            e_up = sharded_pearson_edges(engine, comm, z, bounds, cutoff, stripe_rows=37, upper_only=True)
            e_all = sharded_pearson_edges(engine, comm, z, bounds, cutoff, stripe_rows=50, upper_only=False)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, mean=mean, std=std, r=r, r_row=r_row, r_col=r_col,
                 e_up_i=e_up[0], e_up_j=e_up[1], e_up_v=e_up[2], e_all_i=e_all[0], e_all_j=e_all[1], e_all_v=e_all[2],
                 blocks=np.array([(0 if b[0] == "row" else 1,) + tuple(b[1:]) for b in blocks], dtype=np.int64),
                 has_nan=np.array(has_nan), lo=np.array(lo), hi=np.array(hi), r_ag=r_ag)
        comm.barrier()
    finally:
        dist.destroy_process_group()
