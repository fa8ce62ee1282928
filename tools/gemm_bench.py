"""Pearson contraction A/B bench: variants (environment knobs, re-read through skr_ctx_reload_knobs before each launch) interleaved in ONE
process, median and min of HIP-event times per launch (cdna_hip_programming.md rule 24), on random
normalised-count-like operands (never zeros: MI355X_MICROARCH.md, DVFS give-back).

    python tools/gemm_bench.py [--rows 50000] [--cols 4096] [--mode self|plain] [--rounds 7] NAME=ENV=VAL[,ENV=VAL] ...
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--rows-b", type=int, default=0)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--mode", default="self", choices=["self", "plain"])
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--normalised", action="store_true", help="operand rows = Log2.post of column-standardised counts (what the "
                    "pipeline feeds the contraction: every cell a different value) instead of log2 of the raw counts (few-valued rows)")
    ap.add_argument("--also", default="", help="a second precision (e.g. f16f8) whose operands are prepared from the same rows and "
                    "timed alternately with the first in the same process")
    ap.add_argument("--diag-lib", action="store_true", help="run on libseekr_hip_diag.so (holds the 4-wave arm, SEEKR_GEMM_WAVE_TILE=1)")
    ap.add_argument("--tile-operand", action="store_true", help="make one random 8 192-row chunk and reuse it, columns rotated, for every other chunk (large shapes under the profiler: the host generator would dominate the run)")
    ap.add_argument("--lib", default="", help="another build of the library to run on (A/B across source versions, one process each)")
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()
    if args.diag_lib:
        _lib.LIB_PATH = _lib.DIAG_LIB_PATH
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    variants = []
    for spec in args.variants or ["base="]:
        name, _, envs = spec.partition("=")
        variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
    ctx = _lib.default_context()
    prec = _lib.PRECISIONS[args.precision]
    rng = np.random.default_rng(0)
    m = args.rows
    n = args.rows_b or m

    def operand(rows, seed, prec=prec):
        # binomial counts -> Log2-like values: what the pipeline feeds the contraction
        chunk = 8192
        op = _lib.Operand(ctx, rows, args.cols, prec)
        base = None
        for r0 in range(0, rows, chunk):
            nr = min(chunk, rows - r0)
            if args.tile_operand and base is not None:
                x = np.roll(base[:nr], 32 * (r0 // chunk) + seed, axis=1)
            else:
                x = rng.binomial(1995, 1.0 / 4096, size=(nr, args.cols)).astype(np.float32) * np.float32(0.5)
                if args.normalised:
                    z = (x - x.mean(0)) / np.maximum(x.std(0), 1e-6)
                    x = np.log2(z + np.abs(z.min()) + 1.0).astype(np.float32)
                else:
                    x = np.log2(x + 1.0)
                base = x
            d = ctx.from_numpy(x.astype(np.float32))
            _lib.operand_fill(ctx, d, op=op.view(r0, nr), precision=prec)
            d.free()
        return op

    a = operand(m, 1)
    b = a if args.mode == "self" else operand(n, 2)
    ops = {name: (a, b) for name, _ in variants}
    if args.also:
        rng = np.random.default_rng(0)  # the same rows again
        a2 = operand(m, 1, _lib.PRECISIONS[args.also])
        b2 = a2 if args.mode == "self" else operand(n, 2, _lib.PRECISIONS[args.also])
        print("%s operands: storage kind %d" % (args.also, a2.kind))
        variants = variants + [(args.also, {})]
        ops[args.also] = (a2, b2)
    r = ctx.empty(m, n)
    res = {name: [] for name, _ in variants}
    for _ in range(args.rounds):
        for name, env in variants:
            a, b = ops[name]
            os.environ.update(env)
            ctx.reload_knobs()
            ctx.prof_reset()
            ctx.prof_enable(True)
            _lib.pearson_gemm_op(ctx, a, b, r, symmetric=args.mode == "self")
            ctx.sync()
            ctx.prof_enable(False)
            res[name].append(sum(ctx.prof_query(nm)[0] for nm in ctx.prof_names() if nm.startswith("pearson_gemm")))
            for key in env:
                os.environ.pop(key, None)
    pairs = float(m) * n
    mult = pairs if args.mode == "plain" else m * (m + 256) / 2.0
    for name, ts in res.items():
        ts = np.array(ts[1:])
        med = float(np.median(ts))
        units = 2 if name == "f16f8" else 3
        print("%-16s median %.3f ms  min %.3f ms  -> %.1f G pairs/s delivered, MFMA executed %.0f TF in 16-bit product-units (%.3f of 2.5 PF)"
              % (name, med, ts.min(), pairs / med / 1e6, units * 2 * args.cols * mult / med / 1e9, units * 2 * args.cols * mult / med / 1e9 / 2500))


if __name__ == "__main__":
    main()
