"""The lean epilogue of the split contraction (pearson_bf16.hip, round 6; SEEKR_GEMM_EPILOGUE=0 sends every tile through the
general loop) against the general one: the SAME BITS in SELF, PLAIN, CROSS and LOWER (row-stripe) mode, for both split
precisions, with whole tiles and ragged edges in one launch, several k chunks (K = 16 384: only the first chunk of a call
may take the lean path), unaligned row pitches (a view of r whose rows start off a 16-byte boundary: lean must decline)
and the thresholding mode (no stores: untouched).  Also against the oracle, within the bar."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import seekr_oracle as orc  # noqa: E402  (checker only)
from seekr_amd import _lib as L  # noqa: E402
from seekr_amd import consumers  # noqa: E402

ctx = L.default_context()
rng = np.random.default_rng(6)


ARMS = ("0", "1", "2", "3")  # the general loop; lean with 64- / 128- / 256-byte row runs


def both(fn):
    """[general, lean 64, lean 128, lean 256]; asserts nothing itself."""
    out = []
    for val in ARMS:
        os.environ["SEEKR_GEMM_EPILOGUE"] = val
        ctx.reload_knobs()
        out.append(fn())
    os.environ.pop("SEEKR_GEMM_EPILOGUE")
    ctx.reload_knobs()
    return out


n_checked = 0
for prec in (L.PREC_F16X3, L.PREC_BF16X3, L.PREC_F16F8):
    for rows, rows_b, cols in ((1500, 1111, 4096), (1024, 768, 4096), (700, 513, 16384), (1280, 1024, 8192), (600, 300, 1024), (2051, 2051, 256), (6200, 5000, 64), (9000, 4100, 96)):
        if prec == L.PREC_BF16X3 and cols < 1024:
            continue
        xa = (rng.binomial(40, 0.06, size=(rows, cols)) * np.float32(0.5)).astype(np.float32)
        xb = (rng.binomial(40, 0.06, size=(rows_b, cols)) * np.float32(0.5)).astype(np.float32)
        za, _ = L.operand_fill(ctx, ctx.from_numpy(xa), precision=prec)
        zb, _ = L.operand_fill(ctx, ctx.from_numpy(xb), precision=prec)

        def self_block():
            r = ctx.zeros(rows, rows)
            L.pearson_gemm_op(ctx, za, za, r, symmetric=True)
            return r.to_numpy()

        def plain_block():
            r = ctx.zeros(rows, rows_b)
            L.pearson_gemm_op(ctx, za, zb, r)
            return r.to_numpy()

        def plain_into_wide(col0=3):  # a block written at a column offset that is not a multiple of 4: its rows start off a 16-byte boundary
            r = ctx.zeros(rows, rows_b + 8)
            L.pearson_gemm_op(ctx, za, zb, r, row0=0, col0=col0)
            return r.to_numpy()

        def cross_block():
            r, rt = ctx.zeros(rows, rows_b), ctx.zeros(rows_b, rows)
            L.pearson_gemm_op_mirror(ctx, za, zb, r, 0, 0, rt, 0, 0)
            return np.concatenate([r.to_numpy().ravel(), rt.to_numpy().ravel()])

        def stripes():  # r by row stripes: LOWER left of the diagonal block, SELF on it, PLAIN right of it
            r = ctx.zeros(rows, rows)
            for r0 in range(0, rows, 512):
                nr = min(512, rows - r0)
                L.pearson_gemm_op_rows(ctx, za.view(r0, nr), za, r0, r, r0)
            return r.to_numpy()

        def edge_list():
            fe = consumers.FusedEdges(ctx)
            got = fe.block(za, zb, 0.02, row_global0=7, col_global0=11)
            fe.free()
            return np.concatenate([g.view(np.uint32) for g in got])

        cases = [("self", self_block), ("plain", plain_block), ("plain_unaligned", plain_into_wide), ("cross", cross_block),
                 ("edges", edge_list), ("stripes", stripes)]
        for name, fn in cases:
            base, *arms = both(fn)
            for val, arm in zip(ARMS[1:], arms):
                assert base.shape == arm.shape and np.array_equal(base.view(np.uint32), arm.view(np.uint32)), (name, val, prec, rows, cols)
                n_checked += 1
        assert np.array_equal(both(stripes)[3].view(np.uint32), both(self_block)[3].view(np.uint32)), "stripes != self"
        if prec == L.PREC_F16X3:  # the default precision against the oracle, within the bar (the other two: bit identity only here)
            want = orc.pearson(xa, xb)
            got = both(plain_block)[3]
            assert np.allclose(got, want, rtol=1e-5, atol=2e-6), (prec, rows, cols, np.abs(got - want).max())
        sym = both(self_block)[3]
        assert np.array_equal(sym, sym.T)
print("epilogue check ok: %d comparisons bit-identical" % n_checked)
