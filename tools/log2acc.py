import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from seekr_amd import _lib as L
ctx = L.default_context()
rng = np.random.default_rng(0)
vals = np.concatenate([np.float32(1) + np.float32(2.0) ** rng.uniform(-23, 5, 2_000_000).astype(np.float32),
                       rng.uniform(1, 64, 2_000_000).astype(np.float32)]).astype(np.float32)
x = (vals - np.float32(1)).astype(np.float32).reshape(-1, 4000)   # apply computes log2((x + 0) + 1)
d = ctx.from_numpy(x)
L.apply(ctx, d, post=True, shift=0.0)
got = d.to_numpy().astype(np.float64).reshape(-1)
arg = (x.reshape(-1) + np.float32(1)).astype(np.float32)
truth = np.log2(arg.astype(np.float64))
cr = truth.astype(np.float32).astype(np.float64)
nz = truth != 0
rel = np.abs(got - truth)[nz] / np.abs(truth[nz])
print('post log2: max rel err %.3e  max abs err %.3e  frac != correctly rounded %.4f' % (rel.max(), np.abs(got - truth).max(), (got != cr).mean()))
npv = np.log2(arg).astype(np.float64)
print('numpy f32 log2: max rel err %.3e  frac != CR %.4f   gpu==numpy frac %.4f' % ((np.abs(npv - truth)[nz] / np.abs(truth[nz])).max(), (npv != cr).mean(), (got == npv).mean()))
