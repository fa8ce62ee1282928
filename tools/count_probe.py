"""Times the counting kernel alone on config-2 data (debug helper)."""
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from seekr_amd import _lib as L
from seekr_amd.synthetic import synthetic_ascii
ctx = L.default_context()
n, length, k = 50000, 2000, 6
blob, off = synthetic_ascii(2, n, length)
packed = L.PackedSeqs.from_buffer(ctx, blob, off, "AGTC")
x = ctx.empty(n, 4 ** k)
for _ in range(3): L.count_per_kb(ctx, packed, k, out=x)
ctx.sync(); ctx.prof_enable(True)
for _ in range(10): L.count_per_kb(ctx, packed, k, out=x)
ms, cnt = ctx.prof_query("count_kmers_f32")
print("SEEKR_DBG=%s  %.4f ms  %.0f GB/s" % (os.environ.get("SEEKR_DBG", "0"), ms / cnt, n * (16384 + 508) / (ms / cnt * 1e-3) / 1e9))
