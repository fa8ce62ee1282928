"""The result pool under stress on the GPU box: hundreds of pearson() / get_counts() results of random sizes (1 MB ... 1 GB),
kept for random lengths of time, views outliving their arrays, two threads calling at once, the cap forcing drops — every
result checked against a float64 product of the same rows (cheap sizes) or by symmetry / unit diagonal, the pool's
registered blocks released at the end.  `python tools/pool_stress.py [seconds]`"""
import gc
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from seekr_amd import _lib  # noqa: E402
from seekr_amd.kmer_counts import BasicCounter  # noqa: E402
from seekr_amd.pearson import pearson  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
os.environ.setdefault("SEEKR_RESULT_POOL_MB", "3000")  # small enough that the cap is hit
errors, done = [], {"pearson": 0, "counts": 0}
t_end = time.time() + budget


def worker(seed):
    rng = np.random.default_rng(seed)
    held = []
    try:
        while time.time() < t_end:
            kind = rng.integers(0, 4)
            if kind < 3:
                n = int(rng.choice([300, 700, 1500, 4000, 9000, 16000]))
                cols = int(rng.choice([64, 256, 1000, 4096]))
                x = rng.standard_normal((n, cols)).astype(np.float32)
                r = pearson(x, x)
                assert r.shape == (n, n) and r.dtype == np.float32
                assert np.array_equal(r, r.T) and np.abs(np.diag(r) - 1).max() < 1e-5
                if n <= 1500:
                    z = (x - x.mean(1, keepdims=True)) / x.std(1, keepdims=True)
                    want = (z.astype(np.float64) @ z.astype(np.float64).T) / cols
                    assert np.abs(r - want).max() < 2e-5, np.abs(r - want).max()
                done["pearson"] += 1
                held.append(r[rng.integers(0, n):] if rng.integers(0, 2) else r)  # sometimes only a view survives
            else:
                n = int(rng.choice([200, 3000, 20000]))
                seqs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(50, 400)))) for _ in range(min(n, 400))] * (n // min(n, 400))
                c = BasicCounter(k=int(rng.choice([3, 5, 6])), silent=True, mean=False, std=False, log2="Log2.none")
                c.seqs = seqs
                c.get_counts()
                assert c.counts.shape[0] == len(seqs) and np.isfinite(c.counts).all()
                assert np.array_equal(c.counts[0], c.counts[min(n, 400)]) if len(seqs) > 400 else True
                done["counts"] += 1
                held.append(c.counts)
            while len(held) > rng.integers(0, 6):
                held.pop(int(rng.integers(0, len(held))))
            if rng.integers(0, 8) == 0:
                gc.collect()
    except BaseException:  # noqa: BLE001
        import traceback
        errors.append(traceback.format_exc())
        raise


threads = [threading.Thread(target=worker, args=(s,)) for s in (1, 2)]
for t in threads:
    t.start()
for t in threads:
    t.join()
gc.collect()
print("pool stress:", done, _lib.host_pool.stats, "errors:", len(errors))
for e in errors:
    print(e)
_lib.host_pool.clear()
assert not errors
print("pool stress ok")
