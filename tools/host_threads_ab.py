"""A/B of SEEKR_HOST_THREADS (reader + packer threads, default 16) on the config-2 FASTA file, GPU box: the native open
(parse only), pack + upload, and BasicCounter(infasta).get_counts() to a numpy matrix; interleaved, medians of 5."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seekr_amd import _lib
from seekr_amd.kmer_counts import BasicCounter
from seekr_amd.synthetic import synthetic_ascii

d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "c2.fa")
blob, off = synthetic_ascii(2, 50000, 2000)
with open(path, "wb") as fh:
    for i in range(50000):
        fh.write(b">s%d\n" % i)
        fh.write(blob[off[i]:off[i + 1]].tobytes())
        fh.write(b"\n")
ctx = _lib.default_context()
res = {}
for rep in range(5):
    for th in (8, 16, 32, 64, 128):
        os.environ["SEEKR_HOST_THREADS"] = str(th)
        t0 = time.perf_counter(); f = _lib.FastaFile(path); t1 = time.perf_counter()
        p = f.pack(ctx, alphabet="AGTC"); ctx.sync(); t2 = time.perf_counter()
        p.free(); del f
        t3 = time.perf_counter()
        c = BasicCounter(path, k=6, mean=False, std=False, log2="Log2.none", silent=True); c.get_counts()
        t4 = time.perf_counter()
        del c
        res.setdefault(th, []).append((t1 - t0, t2 - t1, t4 - t3))
for th, rows in res.items():
    m = np.median(np.array(rows), axis=0) * 1e3
    print("threads %3d  open %.1f ms  pack+H2D %.1f ms  BasicCounter(file).get_counts() %.1f ms = %.0f Mbases/s" % (th, m[0], m[1], m[2], 100e3 / m[2]))
