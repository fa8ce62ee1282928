"""One rank of the multi-rank-on-one-GPU test: the production stack (launch.init -> RcclComm ->
HipEngine -> seekr_amd.distributed) with tests/mock_rccl standing in for librccl."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, n_total, length, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from seekr_amd import _lib, launch
    from seekr_amd.distributed import (HipEngine, shard_bounds, sharded_normalize_prepare, sharded_pearson_allgather,
                                       sharded_pearson_edges, sharded_pearson_rowblock, sharded_pearson_symmetric)
    from seekr_amd.synthetic import synthetic_ascii
    ctx, comm = launch.init()
    rank, size = comm.rank, comm.size
    bounds = shard_bounds(n_total, size)
    lo, hi = bounds[rank], bounds[rank + 1]
    blob, offsets = synthetic_ascii(11, hi - lo, length, start=lo)
    raw_mode = os.environ.get("MOCK_RAW_HOMOPOLYMERS") == "1"
    if raw_mode and rank == 0:  # two identical homopolymers: one-hot raw count rows, only on this rank
        blob = np.array(blob, copy=True)
        blob[:2 * length] = ord("A")
    packed = _lib.PackedSeqs.from_buffer(ctx, blob, offsets, "AGTC")
    x = _lib.count_per_kb(ctx, packed, k)
    x8_mode = os.environ.get("MOCK_X8", "")
    if x8_mode:
        # the opt-in f16f8 precision across ranks: "keep" = every shard keeps the H / X layout (kind 3); "route" = rank 0's
        # rows are few-valued, its fill routes them back to the three-product split, and every rank must follow
        from x8_case import x8_matrix
        full = x8_matrix(n_total, x.cols, x8_mode == "route", bounds[1])
        x = ctx.from_numpy(full[lo:hi])
        engine = HipEngine(ctx, _lib.PREC_F16F8)
        _, _, has_nan, z = sharded_normalize_prepare(engine, comm, x, n_total, "Log2.none", False, False, keep_counts=False)
        max_shard = max(bounds[g + 1] - bounds[g] for g in range(size))
        recv = [engine.empty_operand(max_shard, x.cols) for _ in range(2)]
        r = ctx.zeros(hi - lo, n_total)
        sharded_pearson_rowblock(engine, comm, z, bounds, r, recv)
        r_row, r_col = ctx.zeros(hi - lo, n_total), ctx.zeros(n_total, hi - lo)
        blocks = sharded_pearson_symmetric(engine, comm, z, bounds, r_row, r_col, recv)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), r=r.to_numpy(), kind=np.array(z.kind), lo=np.array(lo), hi=np.array(hi),
                 r_row=r_row.to_numpy(), r_col=r_col.to_numpy(),
                 blocks=np.array([(0 if b[0] == "row" else 1,) + tuple(b[1:]) for b in blocks], dtype=np.int64))
        comm.barrier()
        ctx.sync()
        return
    engine = HipEngine(ctx)
    coherent_mode = os.environ.get("MOCK_COHERENT_LAST_RANK") == "1"
    if coherent_mode:
        # only the LAST rank holds rows that are mostly one repeated value (tools/margin_probe.py's worst case):
        # the flag must become global or a cross block would be accumulated differently by its two ranks
        from coherent_case import coherent_matrix
        full = coherent_matrix(n_total, x.cols, size)
        x = ctx.from_numpy(full[lo:hi])
        mean, std, has_nan, z = sharded_normalize_prepare(engine, comm, x, n_total, "Log2.none", False, False)
        assert z.kind != 0 and z.coherent, "rank %d: coherent flag not global (kind %d)" % (rank, z.kind)
        mean = std = ctx.zeros(1, x.cols)
    elif raw_mode:  # raw counts straight into Pearson: the shape that needs the fp32 kernel's dynamic range
        mean, std, has_nan, z = sharded_normalize_prepare(engine, comm, x, n_total, "Log2.none", False, False)
        assert z.kind == 0, "every rank must have switched to the float32 layout"
        mean = std = ctx.zeros(1, x.cols)
    else:
        mean, std, has_nan, z = sharded_normalize_prepare(engine, comm, x, n_total, "Log2.post", True, True)
    max_shard = max(bounds[g + 1] - bounds[g] for g in range(size))
    grouped = os.environ.get("MOCK_GROUPED_SHIFTS") == "1"  # one receive buffer per shift, all shifts in one ncclGroup
    recv = [engine.empty_operand(max_shard, x.cols) for _ in range(max(2, size // 2) if grouped else 2)]
    r_row, r_col = ctx.zeros(hi - lo, n_total), ctx.zeros(n_total, hi - lo)
    blocks = sharded_pearson_symmetric(engine, comm, z, bounds, r_row, r_col, recv, grouped=grouped)
    recv = recv[:2]
    r = ctx.zeros(hi - lo, n_total)
    sharded_pearson_rowblock(engine, comm, z, bounds, r, recv)
    r_ag = ctx.zeros(hi - lo, n_total)
    sharded_pearson_allgather(engine, comm, z, bounds, r_ag)
    e = sharded_pearson_edges(engine, comm, z, bounds, 0.05, stripe_rows=128, upper_only=True)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mean=mean.vector(), std=std.vector(), x=x.to_numpy(),
             r_row=r_row.to_numpy(), r_col=r_col.to_numpy(), r=r.to_numpy(), has_nan=np.array(has_nan),
             blocks=np.array([(0 if b[0] == "row" else 1,) + tuple(b[1:]) for b in blocks], dtype=np.int64),
             e_i=e[0], e_j=e[1], e_v=e[2], lo=np.array(lo), hi=np.array(hi),
             chain_note=np.array(getattr(comm, "_chain_note", "")), r_ag=r_ag.to_numpy())
    comm.barrier()
    ctx.sync()


if __name__ == "__main__":
    main()
