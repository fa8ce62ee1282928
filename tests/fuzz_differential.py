"""Differential fuzzing of the drop-in API against the oracle on the GPU box: random sequence sets
(ragged lengths around word and sweep boundaries, N runs, lower case, homopolymers), random k,
alphabet order, log2 mode and mean/std handling.  Raw counts, statistics and Log2.none pipelines
must match bit for bit; log2 pipelines and Pearson r within the parity bars."""
import io, os, sys, contextlib, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root: oracle/, seekr_amd/
from oracle import seekr_oracle as orc
from seekr_amd.kmer_counts import BasicCounter
from seekr_amd.pearson import pearson

LETTERS = np.array(list("ACGT"))
rng = None


def random_seq(k):
    kind = rng.integers(0, 10)
    if kind == 0:
        L = int(rng.choice([0, 1, k - 2, k, k + 1, 15, 16, 17, 31, 32, 33, 47, 48, 49]))
    elif kind == 1:
        L = int(rng.choice([2047, 2048, 2049, 2048 + k - 1, 4095, 4096, 4097, 4111, 6000]))
    elif kind == 2 and rng.integers(0, 3) == 0:
        # around the 8 192-window tiles of the long-sequence path (count.hip: kItemWindows)
        L = int(rng.choice([8190, 8191 + k - 1, 8192 + k - 1, 8192 + k, 16383 + k, 16384 + k - 1, 16384 + k, 20011, 40000]))
    else:
        L = int(rng.integers(k, 3000))
    L = max(L, 0)
    if L == k - 1:
        L += 1  # ZeroDivisionError in the reference, tested separately
    s = LETTERS[rng.integers(0, 4, L)]
    r = rng.integers(0, 8)
    if r == 0 and L:
        s[:] = LETTERS[rng.integers(0, 4)]                   # homopolymer
    elif r == 1 and L > 4:
        a = int(rng.integers(0, L - 1)); b = int(min(L, a + rng.integers(1, 40)))
        s[a:b] = "N"
    elif r == 2 and L > 4:
        s[rng.integers(0, L, max(1, L // 50))] = rng.choice(list("NnRYacgt"))
    return "".join(s)


def run(seqs, **kw):
    c = BasicCounter(silent=True, **kw)
    c.seqs = list(seqs)
    with contextlib.redirect_stdout(io.StringIO()):
        c.get_counts()
    return c


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def dump_case(seed, n_cases, seqs, tag):
    """Keep a failing case for replay (python tools/replay_case.py file)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "fuzz_fail_%d_%d.npz" % (seed, n_cases))
    np.savez(path, seqs=np.array(seqs, dtype=object), tag=np.array([repr(tag)], dtype=object))
    print("failing case saved to", path)


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from strict_tally import StrictTally  # noqa: E402
import parity_rule  # noqa: E402

TALLY = StrictTally()

def gen_case(r):
    """One random pipeline case: (seqs, k, alphabet, log2, mean, std, tag)."""
    global rng
    rng = r
    k = int(rng.integers(1, 8))
    alphabet = "".join(rng.permutation(list("AGTC")))
    if rng.integers(0, 4) == 0:  # an alphabet the 2-bit path does not cover: the general counting kernel
        alphabet = str(rng.choice(["ACGTN", "AT", "AGTA", "GCA", "ACGTRYN", "NNA", "T"]))
        while len(alphabet) ** k > 4096:
            k -= 1
    n = int(rng.integers(2, 60))
    seqs = [random_seq(k) for _ in range(n)]
    for i in range(1, n):  # mutated copies: pairs with r close to 1, where the bar is relative
        if rng.integers(0, 4) == 0 and len(seqs[i - 1]) > k + 2:
            parent = np.array(list(seqs[int(rng.integers(0, i))]))
            if len(parent) > k + 2:
                hits = rng.random(len(parent)) < rng.choice([0.0, 0.01, 0.05, 0.2])
                parent[hits] = LETTERS[rng.integers(0, 4, int(hits.sum()))]
                seqs[i] = "".join(parent[: int(rng.integers(k + 2, len(parent) + 1))])
    log2 = str(rng.choice(["Log2.none", "Log2.pre", "Log2.post"]))
    mean, std = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    tag = dict(k=k, alphabet=alphabet, n=n, log2=log2, mean=mean, std=std)
    return seqs, k, alphabet, log2, mean, std, tag


def fuzz(seed, budget_s=60.0, max_cases=10 ** 9):
  rng_ = np.random.default_rng(seed)
  t0, n_cases = time.time(), 0
  while time.time() - t0 < budget_s and n_cases < max_cases:
   try:
        seqs, k, alphabet, log2, mean, std, tag = gen_case(rng_)
        n = len(seqs)
        raw = orc.raw_counts(seqs, k, alphabet=alphabet)
        got = run(seqs, k=k, alphabet=alphabet, mean=False, std=False, log2="Log2.none").counts
        assert np.array_equal(bits(got), bits(raw)), ("raw", tag)
        with np.errstate(all="ignore"):
            ref, rmean, rstd = orc.normalize(raw, mean=mean, std=std, log2=log2)
        c = run(seqs, k=k, alphabet=alphabet, mean=mean, std=std, log2=log2)
        same_nan = np.array_equal(np.isnan(c.counts), np.isnan(ref))
        assert same_nan, ("nan pattern", tag)
        if log2 == "Log2.none":
            assert np.array_equal(bits(np.nan_to_num(c.counts)), bits(np.nan_to_num(ref))), ("normalised", tag)
        elif log2 == "Log2.post":
            assert np.allclose(c.counts, ref, rtol=1e-5, atol=2e-6, equal_nan=True), ("normalised log2", tag)
        else:
            # Log2.pre takes column statistics of log2 outputs, which agree with numpy's only to 1 ulp.  After
            # (x - mean) / std that ulp weighs ulp(x) / std(column) in absolute terms — 7e-6 for a column of
            # values around 6 with a spread of 0.07 — for numpy as for the device, so the absolute part of
            # the bar is widened by two such units per column.
            with np.errstate(all="ignore"):
                pre = np.log2(raw + np.float32(1))
                col_std = np.std(pre.astype(np.float64), axis=0) if std else np.ones(pre.shape[1])
                unit = np.spacing(np.abs(pre).max(axis=0).astype(np.float32)).astype(np.float64) / np.maximum(col_std, 1e-30)
            # ... and the std itself moves by ulp / std relatively, which scales every value of the column
            ulp = np.spacing(np.abs(pre).max(axis=0).astype(np.float32)).astype(np.float64)
            rel = (4.0 * ulp / np.maximum(col_std, 1e-30)) if std else np.zeros(pre.shape[1])
            tol = 2e-6 + 2.0 * unit[None, :] + (1e-5 + rel[None, :]) * np.abs(ref)
            with np.errstate(all="ignore"):
                okc = (np.abs(c.counts.astype(np.float64) - ref) <= tol) | (np.isnan(c.counts) & np.isnan(ref)) | (col_std[None, :] < 1e-6)
            if not okc.all():
                i, j = np.argwhere(~okc)[0]
                print("Log2.pre mismatch at", i, j, "ours", c.counts[i, j], "oracle", ref[i, j], "\nraw column", raw[:, j],
                      "\npre column", pre[:, j], "\nours column", c.counts[:, j], "\noracle column", ref[:, j],
                      "\nlens", [len(q) for q in seqs])
            assert okc.all(), ("normalised Log2.pre", tag)
        if log2 != "Log2.pre":
            if mean:
                assert np.array_equal(bits(c.mean), bits(rmean)), ("mean", tag)
            if std:
                assert np.array_equal(bits(np.nan_to_num(c.std)), bits(np.nan_to_num(rstd))), ("std", tag)
        # Pearson is checked on OUR normalised counts: log2 outputs agree with numpy's only to 1 ulp, and on
        # rows of 4 or 16 near-equal values (k = 1, 2) row standardisation amplifies that ulp to 1e-5 in r
        ref = np.array(c.counts, dtype=np.float32)
        with np.errstate(all="ignore"):
            want = orc.pearson(ref, ref)
        r = pearson(c.counts, c.counts)
        # a constant row has no correlation: numpy returns NaN or +-inf for it depending on how its
        # pairwise row sum happens to round (mean off by an ulp -> -2e-7 / 0 = -inf); treat both alike
        with np.errstate(all="ignore"):
            r = np.where(np.isinf(r), np.nan, r)
            want = np.where(np.isinf(want), np.nan, want)
        if not np.array_equal(np.isnan(r), np.isnan(want)):
            i, j = np.argwhere(np.isnan(r) != np.isnan(want))[0]
            print("NaN mismatch at", i, j, "ours", r[i, j], "numpy", want[i, j], "\nrow i", ref[i][:24], "\nrow j", ref[j][:24])
        assert np.array_equal(np.isnan(r), np.isnan(want)), ("pearson nan", tag)
        # Strict against the reference's float32 result; a cell may leave it only where the float32 inner product is
        # order-sensitive — an input property (tests/parity_rule.py) — and must then be within the bar of float64.
        with np.errstate(all="ignore"):
            truth = orc.pearson_f64_truth(ref, ref)
        ok = ~np.isnan(want) & ~np.isnan(truth)
        verdict = parity_rule.judge(r, want, truth, ok, ref, ref)
        TALLY.add(verdict, tag)
        if verdict["failures"]:
            i, j, why, numbers = verdict["failures"][0]
            print(why, "at", i, j, numbers)
            print("row i", ref[i][:16], "\nrow j", ref[j][:16])
        assert not verdict["failures"], ("pearson", tag, verdict["failures"][:3])
        n_cases += 1
   except AssertionError:
      dump_case(seed, n_cases, seqs, tag)
      raise
  return n_cases


if __name__ == "__main__":
    n = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("fuzz ok: %d cases" % n)
