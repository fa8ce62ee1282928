"""In-kernel anatomy of the split-fp16 contraction (libseekr_hip_diag.so: `python -m seekr_amd.build --diag`): per tile, the
shader cycles of the k loop and of the epilogue (s_memtime) and the in-kernel clock (delta s_memtime /
delta s_memrealtime x 100 MHz), after >= 2 s of back-to-back launches on random data
(MI355X_MICROARCH.md, DVFS give-back item 6).

    python tools/gemm_diag.py [--rows 50000] [--mode self]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402

if not os.path.exists(_lib.DIAG_LIB_PATH):
    raise SystemExit("build the diagnostic library first: python -m seekr_amd.build --diag")
_lib.LIB_PATH = _lib.DIAG_LIB_PATH  # before the first call: this process runs on the diagnostic library

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=50000)
ap.add_argument("--cols", type=int, default=4096)
ap.add_argument("--mode", default="self", choices=["self", "plain"])
ap.add_argument("--data", default="counts", choices=["counts", "zeros", "ones"],
                help="operand values: normalised-count-like (default), all zero, or all one (constant rows standardise to NaN -> use raw fill)")
ap.add_argument("--no-dma", action="store_true", help="k loop without its LDS-DMA staging (timing experiment; r is garbage)")
ap.add_argument("--same-tile", action="store_true", help="every stage re-loads k tile 0: the staging traffic stays, its source is the nearest cache (timing experiment; r is garbage)")
ap.add_argument("--two-units", action="store_true", help="TIMING CEILING of a two-product-unit split (VERDICT r3 #5): one v_mfma_i32_16x16x64_i8 in place of the two cross products, same staging and LDS reads; r is garbage.  Printed next to a stamps-only launch of the same process")
ap.add_argument("--half-x", action="store_true", help="f16f8 operands; TIMING CEILING of an H / X layout whose X line carries lo8 alone (staged half); r is garbage.  Printed next to a stamps-only launch of the f16f8 kernel")
ap.add_argument("--no-mirror", action="store_true", help="self mode without the mirror stores (timing experiment; the lower triangle stays unwritten)")
ap.add_argument("--no-stores", action="store_true", help="NO epilogue store at all (round 6): the timing ceiling of hiding the epilogue under the next tile's k loop; r stays unwritten.  Printed next to a stamps-only launch of the same process")
args = ap.parse_args()
ctx = _lib.default_context()
rng = np.random.default_rng(0)
PREC = _lib.PREC_F16F8 if args.half_x else _lib.PREC_F16X3
op = _lib.Operand(ctx, args.rows, args.cols, PREC)
for r0 in range(0, args.rows, 8192):
    nr = min(8192, args.rows - r0)
    x = np.log2(rng.binomial(1995, 1.0 / 4096, size=(nr, args.cols)).astype(np.float32) * np.float32(0.5) + 1.0)
    if args.half_x:  # every cell a different value (column-standardised counts), or the fill routes the rows back to f16x3
        c = rng.binomial(1995, 1.0 / 4096, size=(nr, args.cols)).astype(np.float32)
        zc = (c - c.mean(0)) / np.maximum(c.std(0), 1e-6)
        x = np.log2(zc + np.abs(zc.min()) + 1.0).astype(np.float32)
    if args.data == "zeros":
        x[:] = 0
    d = ctx.from_numpy(x.astype(np.float32))
    # zeros: rows used as they are (row standardisation of a constant row is 0/0)
    _lib.operand_fill(ctx, d, op=op.view(r0, nr), precision=PREC, row_standardize=args.data == "counts")
    d.free()
b = op
if args.mode == "plain":
    b = op.view(0, args.rows)  # a distinct handle: PLAIN mode multiplies every tile
r = ctx.empty(args.rows, args.rows)
sym = args.mode == "self"
t_end = time.time() + 2.0
n = 0
while time.time() < t_end:  # warm the chip up to its steady clock
    _lib.pearson_gemm_op(ctx, op, b, r, symmetric=sym)
    ctx.sync()
    n += 1
def one_launch(mode, label):
    _lib.check(_lib.lib().skr_gemm_diag_mode(ctx._h, mode))
    ctx.prof_reset()
    ctx.prof_enable(True)
    _lib.pearson_gemm_op(ctx, op, b, r, symmetric=sym)
    ctx.sync()
    ctx.prof_enable(False)
    ms = sum(ctx.prof_query(nm)[0] for nm in ctx.prof_names() if nm.startswith("pearson_gemm"))
    _lib.check(_lib.lib().skr_gemm_diag_mode(ctx._h, 0))
    rec = np.zeros((65536, 8), dtype=np.uint64)
    cnt = C.c_int64(0)
    _lib.check(_lib.lib().skr_gemm_diag_read(ctx._h, rec.ctypes.data_as(C.c_void_p), 65536, C.byref(cnt)))
    rec = rec[:min(cnt.value, 65536)].astype(np.float64)
    t0, rt0, k0, k1, t1, rt1 = (rec[:, i] for i in range(6))
    clock = (t1 - t0) / (rt1 - rt0) * 100e6 / 1e9
    print("---- %s: launch %.3f ms (HIP events); %d tile records" % (label, ms, len(rec)))
    print("in-kernel clock: median %.3f GHz (5-95 %%: %.3f - %.3f)" % (np.median(clock), *np.percentile(clock, [5, 95])))
    for name, v in (("tile total", t1 - t0), ("prologue (slot + first stage)", k0 - t0), ("k loop", k1 - k0), ("epilogue + store drain", t1 - k1)):
        print("%-30s median %8.0f cycles = %6.1f us   (5-95 %%: %.0f - %.0f)"
              % (name, np.median(v), np.median(v) / np.median(clock) / 1e3, *np.percentile(v, [5, 95])))
    kt = (args.cols + 31) // 32
    units = 64 if (mode == 5 or args.half_x) else 96
    mfma = kt * 2 * units * 16
    print("MFMA cycles per tile and SIMD (2 waves x %d MFMA x 16 cycles x %d k tiles): %d = %.3f of the k loop, %.3f of the tile"
          % (units, kt, mfma, mfma / np.median(k1 - k0), mfma / np.median(t1 - t0)))
    return ms


print("%d warm-up launches" % n)
mode = 2 if args.no_dma else (3 if args.same_tile else (4 if args.no_mirror else (5 if args.two_units else 1)))
if args.half_x:
    assert op.kind == 3, "the operand was routed back to f16x3 (kind %d)" % op.kind
    for rep in range(3):
        a_ms = one_launch(1, "f16f8 as shipped (stamps only)")
        b_ms = one_launch(6, "f16f8 with the X line staged half (timing ceiling)")
        print("==== launch %.3f -> %.3f ms: %+.1f %%" % (a_ms, b_ms, (b_ms / a_ms - 1) * 100))
elif args.no_stores:
    for rep in range(3):
        a_ms = one_launch(1, "as shipped (stamps only)")
        b_ms = one_launch(7, "no epilogue stores (timing ceiling)")
        print("==== launch %.3f -> %.3f ms: %+.1f %%" % (a_ms, b_ms, (b_ms / a_ms - 1) * 100))
elif mode == 5:
    for rep in range(3):  # alternate in one process: stamps only / two product-units
        a_ms = one_launch(1, "three product-units (shipped arithmetic, stamps only)")
        b_ms = one_launch(5, "two product-units (timing ceiling)")
        print("==== launch %.3f -> %.3f ms: %+.1f %%" % (a_ms, b_ms, (b_ms / a_ms - 1) * 100))
else:
    one_launch(mode, "diag mode %d" % mode)
