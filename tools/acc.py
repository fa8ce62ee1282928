"""Error of each contraction mode against float64 truth on config-2 data (8192 rows)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from seekr_amd import _lib as L
from seekr_amd.synthetic import synthetic_ascii

ctx = L.default_context()
n, length, k = 8192, 2000, 6
blob, off = synthetic_ascii(2, n, length)
x = L.count_per_kb(ctx, L.PackedSeqs.from_buffer(ctx, blob, off, "AGTC"), k)
L.normalize(ctx, x, "Log2.post", 1, None, 1, None)
xs = x.to_numpy().astype(np.float64)
c = (xs.T - xs.mean(axis=1)).T
z = (c.T / c.std(axis=1)).T
truth = z[:2048] @ z.T / 4096
ref32 = None
for prec in ("fp32", "bf16x3", "f16x3"):
    r = L.pearson(ctx, x, x, precision=L.PRECISIONS[prec]).to_numpy()[:2048]
    e = np.abs(r - truth)
    diag = np.abs(np.diag(r[:, :2048]) - 1.0)
    off = e.copy()
    off[np.arange(2048), np.arange(2048)] = 0
    tol = 2e-6 + 1e-5 * np.abs(truth)
    print("%-7s max|err| %.3e  diag max %.3e  offdiag max %.3e rms %.3e  worst err/tol %.3f" % (
        prec, e.max(), diag.max(), off.max(), np.sqrt((off ** 2).mean()), (e / tol).max()))
