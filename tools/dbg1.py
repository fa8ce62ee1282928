import sys, os, io, contextlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import numpy as np
from oracle import seekr_oracle as orc
from inputs import synth_2000
from seekr_amd.kmer_counts import BasicCounter
from seekr_amd.pearson import pearson
g4 = np.load('/root/repo/tests/golden/g4_synth2000.npz')
seqs = synth_2000()
c = BasicCounter(k=6, log2='Log2.pre', silent=True); c.seqs = seqs; c.get_counts()
for name, a, b in (('mean', c.mean, g4['mean_pre']), ('std', c.std, g4['std_pre'])):
    d = np.abs(a - b); i = int(np.argmax(d / np.abs(b)))
    print(name, 'max abs', d.max(), 'max rel', (d/np.abs(b)).max(), 'at', i, a[i], b[i])
raw = BasicCounter(k=6, mean=False, std=False, log2='Log2.pre', silent=True); raw.seqs = seqs; raw.get_counts()
ref_raw = orc.log2_plus_one(orc.raw_counts(seqs, 6))
d = np.abs(raw.counts - ref_raw); print('log2pre raw: max abs', d.max(), 'n diff', (d>0).sum(), 'of', d.size)
vals, idx = np.unique(ref_raw, return_index=True)
for v in vals[:8]:
    m = ref_raw == v
    print(' value', repr(v), 'gpu', repr(raw.counts[m][0]), 'ulps', int(raw.counts[m][0].view(np.int32)) - int(v.view(np.int32)))
# pearson error margins
x = c.counts
for tag in ('post','none','pre'):
    cc = BasicCounter(k=6, log2='Log2.'+tag, silent=True); cc.seqs = seqs; cc.get_counts()
    r = pearson(cc.counts[:256], cc.counts[:256])
    truth = orc.pearson_f64_truth(cc.counts[:256], cc.counts[:256])
    ref = orc.pearson(cc.counts[:256], cc.counts[:256])
    print(tag, 'gpu-vs-truth', np.abs(r-truth).max(), 'numpy-vs-truth', np.abs(ref-truth).max(), 'gpu-vs-golden', np.abs(r-g4['pearson256_'+tag]).max())
