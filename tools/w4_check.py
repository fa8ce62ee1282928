"""The 4-wave / 128 x 128 wave-tile arm of the split contraction (SEEKR_GEMM_WAVE_TILE=1, libseekr_hip_diag.so only)
against the default 8-wave kernel: the same bits in SELF, PLAIN, CROSS and thresholding (EDGES) mode, with several k
chunks (K = 16 384: later chunks add into C) and ragged edges.  `python -m seekr_amd.build --diag` first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import seekr_oracle as orc  # noqa: E402  (checker only)
from seekr_amd import _lib as L  # noqa: E402

L.LIB_PATH = L.DIAG_LIB_PATH  # before the first call: this process runs on the diagnostic library
from seekr_amd import consumers  # noqa: E402

ctx = L.default_context()
rng = np.random.default_rng(3)


def both(fn):
    out = []
    for val in ("0", "1"):
        os.environ["SEEKR_GEMM_WAVE_TILE"] = val
        ctx.reload_knobs()
        out.append(fn())
    os.environ.pop("SEEKR_GEMM_WAVE_TILE")
    ctx.reload_knobs()
    return out


for rows, rows_b, cols in ((1500, 1111, 4096), (700, 513, 16384), (300, 300, 1024)):
    xa = (rng.binomial(40, 0.06, size=(rows, cols)) * np.float32(0.5)).astype(np.float32)
    xb = (rng.binomial(40, 0.06, size=(rows_b, cols)) * np.float32(0.5)).astype(np.float32)
    za, _ = L.operand_fill(ctx, ctx.from_numpy(xa), precision=L.PREC_F16X3)
    zb, _ = L.operand_fill(ctx, ctx.from_numpy(xb), precision=L.PREC_F16X3)

    def self_block():
        r = ctx.zeros(rows, rows)
        L.pearson_gemm_op(ctx, za, za, r, symmetric=True)
        return r.to_numpy()

    def plain_block():
        r = ctx.zeros(rows, rows_b)
        L.pearson_gemm_op(ctx, za, zb, r)
        return r.to_numpy()

    def cross_block():
        r, rt = ctx.zeros(rows, rows_b), ctx.zeros(rows_b, rows)
        L.pearson_gemm_op_mirror(ctx, za, zb, r, 0, 0, rt, 0, 0)
        return np.concatenate([r.to_numpy().ravel(), rt.to_numpy().ravel()])

    def edge_list():
        fe = consumers.FusedEdges(ctx)
        got = fe.block(za, zb, 0.02, row_global0=7, col_global0=11)
        fe.free()
        return np.concatenate([g.view(np.uint32) for g in got])

    for name, fn in (("self", self_block), ("plain", plain_block), ("cross", cross_block), ("edges", edge_list)):
        base, arm = both(fn)
        assert base.shape == arm.shape and np.array_equal(base.view(np.uint32), arm.view(np.uint32)), (name, rows, cols)
    want = orc.pearson(xa, xb)
    assert np.allclose(both(plain_block)[1], want, rtol=1e-5, atol=2e-6)
print("w4 check ok")
