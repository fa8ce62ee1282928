"""Fused normalise + standardise + split pass (skr_operand_fill) A/B bench: variants interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seekr_amd import _lib
from seekr_amd.distributed import HipEngine, SingleComm, sharded_stats
from seekr_amd.synthetic import synthetic_ascii

rows, k = int(os.environ.get("ROWS", "50000")), int(os.environ.get("K", "6"))
length = 2000 if k == 6 else 5000
ctx = _lib.default_context()
blob, off = synthetic_ascii(2, rows, length)
x0 = _lib.count_per_kb(ctx, _lib.PackedSeqs.from_buffer(ctx, blob, off, "AGTC"), k)
engine = HipEngine(ctx)
center, scale, post, shift = sharded_stats(engine, SingleComm(), x0, rows, "Log2.post", True, True)
x = ctx.empty(rows, 4 ** k)
op = engine.empty_operand(rows, 4 ** k)
variants = []
for spec in sys.argv[1:] or ["base="]:
    name, _, envs = spec.partition("=")
    variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
res = {n: [] for n, _ in variants}
raw = x0.to_numpy()
for _ in range(9):
    for name, env in variants:
        os.environ.update(env)
        ctx.reload_knobs()
        x.upload(raw)
        ctx.sync()
        ctx.prof_reset(); ctx.prof_enable(True)
        engine.prepare(x, center, scale, post, shift, keep_counts=True, op=op)
        ctx.sync(); ctx.prof_enable(False)
        res[name].append(sum(ctx.prof_query(n)[0] for n in ctx.prof_names() if n.startswith("operand_fill")))
        for key in env: os.environ.pop(key, None)
bytes_alg = rows * 4 ** k * 4 * 3.0
for name, ts in res.items():
    ts = np.array(ts[2:]); med = float(np.median(ts))
    print("%-12s median %.4f ms  min %.4f ms -> %.0f GB/s = %.3f of 8 TB/s" % (name, med, ts.min(), bytes_alg / med / 1e6, bytes_alg / med / 1e6 / 8000))
