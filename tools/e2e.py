"""End-to-end (PCIe-inclusive) timings through the drop-in Python API, for DESIGN.md."""
import os
import sys
import time

sys.path.insert(0, '/root/repo')
import numpy as np
from seekr_amd.kmer_counts import BasicCounter
from seekr_amd.pearson import pearson
from seekr_amd.synthetic import synthetic_codes, LETTERS

n, L = 50_000, 2000
path = '/tmp/cfg2.fa'
t0 = time.time()
codes = synthetic_codes(2, n, L)
asc = LETTERS[codes]
with open(path, 'wb') as fh:
    for i in range(n):
        fh.write(b'>s%d\n' % i)
        fh.write(asc[i].tobytes())
        fh.write(b'\n')
print('wrote fasta', round(time.time() - t0, 2), 's', os.path.getsize(path) / 1e6, 'MB')
c = None
for rep in range(3):
    c = None  # the previous counter's 819 MB are released outside the timed span
    t0 = time.time()
    c = BasicCounter(path, k=6, silent=True)
    t1 = time.time()
    c.get_counts()
    t2 = time.time()
    print('rep', rep, 'BasicCounter(infasta): parse+pack+H2D %.3f s; get_counts (count+normalise+D2H of %.0f MB) %.3f s -> %.1f Mbases/s end-to-end'
          % (t1 - t0, c.counts.nbytes / 1e6, t2 - t1, n * L / (t2 - t0) / 1e6))
x = c.counts
for prec in ('fp32', 'bf16x3'):
    os.environ['SEEKR_PRECISION'] = prec
    for m in (8000, 20000):
        t0 = time.time()
        r = pearson(x[:m], x[:m])
        dt = time.time() - t0
        print(prec, 'pearson host->host', m, 'rows: %.3f s -> %.1f M pairs/s (result %.1f GB over PCIe)' % (dt, m * m / dt / 1e6, r.nbytes / 1e9))
        del r
