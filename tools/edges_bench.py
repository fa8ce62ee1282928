"""Striped edge extraction at sizes where r cannot materialise: fused (threshold inside the contraction's epilogue)
against the two-step path (contraction into a stripe buffer, then skr_edges).

    python tools/edges_bench.py [--rows 300000] [--cutoff 0.03]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib, consumers  # noqa: E402
from seekr_amd.distributed import HipEngine, SingleComm, sharded_normalize_prepare  # noqa: E402
from seekr_amd.synthetic import synthetic_ascii  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=300000)
ap.add_argument("--length", type=int, default=2000)
ap.add_argument("-k", type=int, default=6)
ap.add_argument("--cutoff", type=float, default=0.03)
ap.add_argument("--stripe-rows", type=int, default=8192)
args = ap.parse_args()
ctx = _lib.default_context()
n = args.rows
# the edge lists live in HOST memory (12 bytes per edge, several copies while they are compared): refuse shapes whose
# expected list is large — a dense cutoff at 300 000 rows is hundreds of GB and takes the machine down
import math
sigma = 1.0 / math.sqrt(4 ** args.k)
expected = 0.5 * n * n * 0.5 * math.erfc(args.cutoff / sigma / math.sqrt(2.0))
if expected > 1.5e9:
    raise SystemExit("expected ~%.2g edges (%.0f GB of host lists): choose a higher cutoff or fewer rows" % (expected, expected * 12 * 4 / 1e9))
x = ctx.empty(n, 4 ** args.k)
step = 50000
for r0 in range(0, n, step):
    nr = min(step, n - r0)
    blob, off = synthetic_ascii(3, nr, args.length, start=r0)
    _lib.count_per_kb(ctx, _lib.PackedSeqs.from_buffer(ctx, blob, off, "AGTC"), args.k, out=x.view(r0, nr))
engine = HipEngine(ctx)
z = sharded_normalize_prepare(engine, SingleComm(), x, n, "Log2.post", True, True, keep_counts=False)[3]
x.free()
res = {}
for name, fuse in (("two-step", False), ("fused", True), ("auto", "auto"), ("two-step", False), ("fused", True), ("auto", "auto")):
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    e = consumers.pearson_edges(z, args.cutoff, stripe_rows=args.stripe_rows, upper_only=True, fuse=fuse)
    ctx.sync()
    wall = time.perf_counter() - t0
    ctx.prof_enable(False)
    kern = {nm: ctx.prof_query(nm) for nm in ctx.prof_names()}
    res[name] = (wall, e, kern)
    print("%-9s wall %.3f s, %d edges; kernels: %s" % (name, wall, len(e[2]), ", ".join(
        "%s %.0f ms" % (nm, ms) for nm, (ms, c) in sorted(kern.items()) if c)), flush=True)
a, b = res["two-step"][1], res["fused"][1]
same = all(np.array_equal(u.view(np.uint32), v.view(np.uint32)) for u, v in zip(a, b))
print("edge lists bit-identical:", same, " fused / two-step wall: %.3f" % (res["fused"][0] / res["two-step"][0]))
assert same
