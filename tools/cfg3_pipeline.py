"""SURVEY §8 config 3 stand-in (the GENCODE lncRNA file cannot be fetched offline): a length-skewed
synthetic transcript set (log-normal lengths 200 nt .. 1e5 nt, per-transcript base composition,
low-complexity stretches, a few N) pushed through the reference's own three-step recipe with the
public API: norm vectors (Log2.post) -> counts with those vectors -> Pearson against itself.
Times every stage and checks the results against the oracle (raw counts on a prefix; the normalised
matrix and a Pearson block at full size from the oracle's numpy restatement)."""
import argparse
import io
import contextlib
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import seekr_oracle as orc  # noqa: E402  (checker only)
from seekr_amd.kmer_counts import BasicCounter  # noqa: E402
from seekr_amd.pearson import pearson  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=18000)
ap.add_argument("--k", type=int, default=6)
ap.add_argument("--check-prefix", type=int, default=600)
args = ap.parse_args()

rng = np.random.default_rng(33)
lengths = np.clip(np.exp(rng.normal(np.log(1100.0), 0.95, args.rows)), 200, 100000).astype(np.int64)
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
d = tempfile.mkdtemp(dir="/tmp")
fa = os.path.join(d, "lnc.fa")
seqs_prefix = []
with open(fa, "wb") as fh:
    for i, L in enumerate(lengths):
        p = rng.dirichlet([26.0, 24.0, 24.0, 26.0])
        s = letters[rng.choice(4, size=int(L), p=p)]
        if rng.integers(0, 4) == 0:  # a low-complexity stretch (poly-A tail, dinucleotide repeat)
            unit = letters[rng.integers(0, 4, int(rng.integers(1, 4)))]
            run = int(rng.integers(10, 80))
            at = int(rng.integers(0, max(1, L - run)))
            s[at:at + run] = np.resize(unit, run)[: len(s[at:at + run])]
        if rng.integers(0, 50) == 0:
            s[int(rng.integers(0, L))] = ord("N")
        fh.write(b">ENST%08d.1|lnc-%d|%d\n" % (i, i, L))
        for j in range(0, int(L), 60):
            fh.write(s[j:j + 60].tobytes())
            fh.write(b"\n")
        if i < args.check_prefix:
            seqs_prefix.append(s.tobytes().decode())
bases = int(lengths.sum())
print("synthetic lncRNA-like set: %d transcripts, %.1f Mbases, median %d nt, max %d nt"
      % (args.rows, bases / 1e6, int(np.median(lengths)), int(lengths.max())))


def timed(label, fn):
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        out = fn()
    dt = time.time() - t0
    print("%-58s %7.3f s" % (label, dt), flush=True)
    return out, dt


def quiet_counts(**kw):
    c = BasicCounter(fa, k=args.k, silent=True, **kw)
    c.get_counts()
    return c


timed("warm-up (library load, first launches)", lambda: quiet_counts(mean=False, std=False, log2="Log2.none"))
raw, t_raw = timed("raw counts (FASTA -> host f32 matrix)", lambda: quiet_counts(mean=False, std=False, log2="Log2.none"))
print("    -> %.0f Mbases/s file to host array" % (bases / 1e6 / t_raw))
nv, _ = timed("1. norm vectors (Log2.post, mean/std computed)", lambda: quiet_counts(mean=True, std=True, log2="Log2.post"))
mean_vec, std_vec = np.asarray(nv.mean), np.asarray(nv.std)
cn, _ = timed("2. counts normalised with those vectors", lambda: quiet_counts(mean=mean_vec, std=std_vec, log2="Log2.post"))
r, t_r = timed("3. pearson(counts, counts) -> host [N,N] f32", lambda: pearson(cn.counts, cn.counts))
print("    -> %.2f G pairs/s including the %.2f GB device-to-host copy" % (args.rows ** 2 / t_r / 1e9, r.nbytes / 1e9))

# ---- parity against the oracle ----
want_raw = orc.raw_counts(seqs_prefix, args.k)
got_raw = np.asarray(raw.counts)[: args.check_prefix]
assert np.array_equal(got_raw.view(np.uint32), want_raw.view(np.uint32)), "raw per-kb counts differ on the prefix"
x = np.asarray(raw.counts)
want_nv = orc.normalize(x.copy(), True, True, "Log2.post")
want_counts, want_mean, want_std = want_nv if isinstance(want_nv, tuple) else (want_nv, None, None)
if want_mean is not None:
    assert np.array_equal(mean_vec.view(np.uint32), np.asarray(want_mean).view(np.uint32)), "mean vector differs"
    assert np.array_equal(std_vec.view(np.uint32), np.asarray(want_std).view(np.uint32)), "std vector differs"
np.testing.assert_allclose(np.asarray(nv.counts), want_counts, rtol=1e-5, atol=2e-6)
want_cn = orc.normalize(x.copy(), mean_vec, std_vec, "Log2.post")
want_cn = want_cn[0] if isinstance(want_cn, tuple) else want_cn
np.testing.assert_allclose(np.asarray(cn.counts), want_cn, rtol=1e-5, atol=2e-6)
# Pearson: STRICT parity, device against the reference's float32 numpy path (oracle.pearson = pearson.py:35-41) cell by
# cell, |got - ref| <= 2e-6 + 1e-5 |ref| — no slack for the reference's own error — on the leading diagonal block, the
# trailing one and an off-diagonal one (rows of the longest and of the shortest transcripts included)
cnt = np.asarray(cn.counts)
nb = min(3000, args.rows)
order = np.argsort(lengths)
picks = {"first": np.arange(nb), "last": np.arange(args.rows - nb, args.rows),
         "longest": np.sort(order[-nb // 2:]), "shortest": np.sort(order[:nb // 2])}
worst = 0.0
for (na, ia), (nb_, ib) in ((("first", picks["first"]), ("first", picks["first"])), (("last", picks["last"]), ("last", picks["last"])),
                            (("first", picks["first"]), ("last", picks["last"])), (("longest", picks["longest"]), ("shortest", picks["shortest"]))):
    want_r = orc.pearson(cnt[ia], cnt[ib])
    got_r = r[np.ix_(ia, ib)]
    ratio = float(np.max(np.abs(got_r - want_r) / (2e-6 + 1e-5 * np.abs(want_r))))
    worst = max(worst, ratio)
    assert ratio <= 1.0, ("Pearson block %s x %s: %.3f of the bar against the float32 reference path" % (na, nb_, ratio))
print("parity: raw bit-exact on %d transcripts; vectors bit-exact; normalised within 1e-5; "
      "Pearson strict against the float32 reference path on 4 blocks: worst %.3f of the bar" % (args.check_prefix, worst))
print("cfg3 pipeline ok rows=%d" % args.rows)
