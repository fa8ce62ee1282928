"""Counting kernel A/B bench: variants (environment knobs, re-read through skr_ctx_reload_knobs before each launch) interleaved in ONE
process, median and min of HIP-event times per launch (cdna_hip_programming.md rule 24).

    python tools/count_bench.py [--rows 50000] [--length 2000] [-k 6] [--rounds 15] NAME=ENV=VAL[,ENV=VAL] ...

e.g.  python tools/count_bench.py base= flushonly=SEEKR_COUNT_EXP=1
Also times the worst cases of the judge's list: homopolymer rows ("A"*5000, "T"*70000) inside a ragged set.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402
from seekr_amd.synthetic import synthetic_ascii  # noqa: E402


PRE = {"op": None}


def time_variants(ctx, packed, k, out, variants, rounds, kernel="count_kmers_f32"):
    res = {name: [] for name, _ in variants}
    for _ in range(rounds):
        for name, env in variants:
            for key, val in env.items():
                os.environ[key] = val
            ctx.reload_knobs()
            if PRE["op"] is not None:
                PRE["op"]()  # what runs right before the counting kernel in a pipeline step (not timed)
            ctx.prof_reset()
            ctx.prof_enable(True)
            _lib.count_per_kb(ctx, packed, k, out=out)
            ctx.sync()
            ctx.prof_enable(False)
            ms = sum(ctx.prof_query(n)[0] for n in ctx.prof_names() if n.startswith("count"))
            res[name].append(ms)
            for key in env:
                os.environ.pop(key, None)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--length", type=int, default=2000)
    ap.add_argument("-k", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--ragged", action="store_true", help="also time the homopolymer / ragged worst cases")
    ap.add_argument("--pre", default="none", choices=["none", "memset", "gemm"],
                    help="run this right before every timed launch: a 2 GB fill of another buffer, or a 16 384^2 contraction "
                         "(what precedes the counting kernel in a bench step: cache state and clock are then the pipeline's)")
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()
    variants = []
    for spec in args.variants or ["base="]:
        name, _, envs = spec.partition("=")
        env = dict(e.split("=", 1) for e in envs.split(",") if e)
        variants.append((name, env))
    ctx = _lib.default_context()
    blob, off = synthetic_ascii(2, args.rows, args.length)
    packed = _lib.PackedSeqs.from_buffer(ctx, blob, off, "AGTC")
    out = ctx.empty(args.rows, 4 ** args.k)
    bytes_alg = args.rows * (args.length * 0.25 + 8 + 4.0 * 4 ** args.k)
    if args.pre == "memset":
        other = ctx.empty(125000, 4096)
        PRE["op"] = lambda: _lib.check(_lib.lib().skr_mat_fill_zero(other._h))
    elif args.pre == "gemm":
        xs = np.random.default_rng(0).normal(size=(int(os.environ.get('PRE_GEMM_ROWS', '16384')), 4096)).astype(np.float32)
        zop, _ = _lib.operand_fill(ctx, ctx.from_numpy(xs), precision=_lib.PREC_F16X3)
        rbuf = ctx.empty(xs.shape[0], xs.shape[0])
        PRE["op"] = lambda: _lib.pearson_gemm_op(ctx, zop, zop, rbuf, symmetric=True)
    res = time_variants(ctx, packed, args.k, out, variants, args.rounds)
    # calibration of this box: a plain fill of the same matrix (hipMemsetAsync), timed by the host around a sync
    import time
    fills = []
    for _ in range(8):
        if PRE["op"] is not None:
            PRE["op"]()
        ctx.sync()
        t0 = time.perf_counter()
        _lib.check(_lib.lib().skr_mat_fill_zero(out._h))
        ctx.sync()
        fills.append((time.perf_counter() - t0) * 1e3)
    print("fill of the same %.0f MB (host-timed, includes ~10 us of launch + sync): min %.4f ms -> %.0f GB/s"
          % (args.rows * 4.0 * 4 ** args.k / 1e6, min(fills), args.rows * 4.0 * 4 ** args.k / min(fills) / 1e6))
    for name, ts in res.items():
        ts = np.array(ts[2:])
        med = float(np.median(ts))
        print("%-14s median %.4f ms  min %.4f ms  -> %.0f GB/s = %.3f of 8 TB/s, %.1f Gbases/s"
              % (name, med, ts.min(), bytes_alg / med / 1e6, bytes_alg / med / 1e6 / 8000, args.rows * args.length / med / 1e6))
    if args.ragged:
        rng = np.random.default_rng(5)
        seqs = ["A" * 5000, "T" * 70000, "AT" * 20000, "ACG" * 10000]
        lens = rng.integers(200, 6000, size=20000)
        letters = np.frombuffer(b"ACGT", dtype=np.uint8)
        seqs += [letters[rng.integers(0, 4, size=n)].tobytes().decode() for n in lens]
        seqs += ["A" * 5000] * 2000 + ["T" * 70000] * 64
        packed2 = _lib.PackedSeqs.from_strings(ctx, seqs, "AGTC")
        out2 = ctx.empty(len(seqs), 4 ** args.k)
        res = time_variants(ctx, packed2, args.k, out2, variants, max(5, args.rounds // 2))
        total = sum(len(s) for s in seqs)
        for name, ts in res.items():
            ts = np.array(ts[1:])
            print("ragged+homopolymer %-14s median %.4f ms (%d sequences, %.1f Mbases) -> %.1f Gbases/s"
                  % (name, float(np.median(ts)), len(seqs), total / 1e6, total / float(np.median(ts)) / 1e6))


if __name__ == "__main__":
    main()
