"""What one rank of the rank-to-rank float32 column-sum chain costs (kmer_counts.py:168,174 across GPUs).

The running sums of a column pass through the ranks in row order, so per pass the node pays, one after the other,
every rank's walk over its shard plus a hop: the kernel's fixed cost (launch, the staging ring filling up before the
walker has its first tile) and the 16 KiB (k = 6) / 64 KiB (k = 7) vector travelling to the next rank.  On ONE GPU:

  whole    one kernel over all n rows                                  -> the chain's floor on this kernel
  pieces   P kernels over n/P rows each, every one started from the    -> + (P - 1) x kernel fixed cost
           accumulator the previous one left (carried accumulator)
  hops     the same with an RCCL send/recv-to-self of the vector       -> + (P - 1) x transfer (ncclSend/Recv of a few
           between two pieces, waited for by the compute stream            KiB + two stream dependencies)
  mailbox  P chain links (skr_colsum_seq_chain, the ranks' mailboxes   -> the peer-mailbox chain's fixed cost per link (on
           connected inside this process): the kernel of piece g + 1      one GPU the kernels still run one after the other:
           reads what piece g stored into its mailbox                      the wait inside the kernel is over at once)

What cannot be shown on one GPU is the overlap the mailbox chain is built for: between GPUs rank g + 1's kernel is resident,
its staging ring full, while rank g still walks.  Two links do not fit one GPU side by side (a workgroup per CU with ~220
registers per lane: a second one finds no room), so a waiting link would keep the link it waits for from ever starting
(tools/micro/spin_pair.hip shows the same with plain kernels once the waiter fills the chip).

    python tools/chain_bench.py [--rows 200000] [--k 6] [--ranks 8] [--rounds 9]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200000)
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=9)
    args = ap.parse_args()
    n, cols, P = args.rows, 4 ** args.k, args.ranks
    ctx = _lib.Context(0)
    _lib.comm_init(ctx, 1, 0, _lib.comm_unique_id())
    rng = np.random.default_rng(0)
    x = ctx.empty(n, cols)
    block = (rng.binomial(1995, 1.0 / 4096, size=(8192, cols)) * np.float32(0.5)).astype(np.float32)  # count-like values
    for r0 in range(0, n, 8192):  # the same block everywhere: the values do not matter for the timing
        x.upload(block[:min(8192, n - r0)], row0=r0)
    bounds = [n * g // P for g in range(P + 1)]
    shards = [x.view(bounds[g], bounds[g + 1] - bounds[g]) for g in range(P)]
    chains = [_lib.Chain(ctx, cols) for _ in range(P)]
    for g, c in enumerate(chains):
        c.connect_local(g, chains)
    mean = ctx.zeros(1, cols)
    _lib.colsum_seq(ctx, x, mean)
    _lib.vec_finish(ctx, mean, n)
    want = None

    def run(mode, square):
        """One pass; returns (wall ms, kernel ms)."""
        acc = ctx.zeros(1, cols)
        inbox = ctx.zeros(1, cols)
        ctx.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        t0 = time.perf_counter()
        if mode == "whole":
            _lib.colsum_seq(ctx, x, acc, mean if square else None, None, square)
        elif mode == "mailbox":
            accs = [ctx.zeros(1, cols) for _ in range(P)]
            for g in range(P):  # rank g's link ...
                chains[g].colsum(shards[g], accs[g], mean if square else None, None, square, defer_result=True)
            for g in range(P - 1):  # ... and, the last rank's link being in the stream, the copies out of the result boxes
                chains[g].result(accs[g])
            acc = accs[0]  # a non-last rank's copy of the finished sums (arrived through its result box)
        else:
            for g in range(P):
                _lib.colsum_seq(ctx, shards[g], acc, mean if square else None, None, square)
                if mode == "hops" and g + 1 < P:
                    # the vector leaves for the next rank and arrives there: send + recv to self, then the next piece
                    # waits for the arrival (what rank g+1 does) and continues from the received copy
                    t = _lib.comm_sendrecv(ctx, acc, 0, 1, 0, inbox, 0, 1, 0)
                    _lib.comm_wait(ctx, t)
                    acc, inbox = inbox, acc
        ctx.sync()
        wall = (time.perf_counter() - t0) * 1e3
        ctx.prof_enable(False)
        kern = sum(ctx.prof_query(nm)[0] for nm in ctx.prof_names() if nm.startswith("colsum"))
        return wall, kern, acc.vector()

    print("%d rows x %d columns (%.2f GB), %d ranks of %d rows" % (n, cols, n * cols * 4 / 1e9, P, n // P))
    for square in (False, True):
        res = {}
        for mode in ("whole", "pieces", "hops", "mailbox"):
            walls, kerns = [], []
            for _ in range(args.rounds):
                w, kms, vec = run(mode, square)
                walls.append(w)
                kerns.append(kms)
                if want is None:
                    want = {}
                key = (square,)
                if key not in want:
                    want[key] = vec
                assert np.array_equal(vec.view(np.uint32), want[key].view(np.uint32)), "chained result differs from the single pass"
            res[mode] = (float(np.median(walls[1:])), float(np.median(kerns[1:])))
        name = "squared-deviation pass" if square else "plain sum pass"
        print("%s: whole %.3f ms wall (%.3f ms kernel); %d pieces %.3f ms wall (%.3f ms in kernels); with hops %.3f ms wall"
              % (name, *res["whole"], P, *res["pieces"], res["hops"][0]))
        per_piece = (res["pieces"][0] - res["whole"][0]) / max(P - 1, 1) * 1e3
        per_hop = (res["hops"][0] - res["pieces"][0]) / max(P - 1, 1) * 1e3
        print("    -> fixed cost per extra kernel %.1f us, per hop (send/recv of %d KiB + stream waits) %.1f us; chain of %d ranks = %.3f ms "
              "against %.3f ms for the same rows on one GPU" % (per_piece, cols * 4 // 1024, per_hop, P, res["hops"][0], res["whole"][0]))
        print("    mailbox chain (%d links + result copies, one stream): %.3f ms wall" % (P, res["mailbox"][0]))
    print("chain bench ok")


if __name__ == "__main__":
    main()
