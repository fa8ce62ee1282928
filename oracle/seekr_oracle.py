"""CPU oracle for the SEEKR k-mer-count + Pearson hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy / pure-Python *restatement* of the reference algorithm
(CalabreseLab/seekr @ 2024-11-01, v2.0.2); citations are `file:line` relative to
the reference checkout.  It exists so that the HIP path can be checked on a box
where the reference itself is absent.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it; the product package
(`seekr_amd/`) never does and fails loudly when the HIP library is missing.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here
against (a) the reference's own fixtures and known-answer literals
(`seekr/tests/data/*`, `seekr/tests/test_kmer_counts.py`, `test_pearson.py`,
`test_console_scripts.py`), re-typed / re-encoded under `tests/golden/`, and
(b) vectors produced by importing the reference in the build container with
`tests/golden/make_golden.py` (script committed next to the vectors).

Everything here is written for clarity and exactness, not speed.  The two
functions whose *structure* mirrors the reference's hot loops (`occurrences_py`
and `pearson`) are also what `bench.py` times as the "port" CPU baseline.
"""

from collections import defaultdict
from itertools import product

import numpy as np

LOG2_MODES = ("Log2.post", "Log2.pre", "Log2.none")


# --------------------------------------------------------------------------
# FASTA reading — fasta_reader.py:41-78
# --------------------------------------------------------------------------
def read_fasta(path):
    """Return (headers, seqs) with the reference reader's semantics.

    fasta_reader.py:41-45 strips every line; :47-63 treats a line whose first
    character is '>' as a header, concatenates the other lines up to the next
    header and upper-cases them.  Errors reproduced: a blank line raises
    IndexError (`line[0]` on an empty string, :53); a header that directly
    follows another header anywhere but at line 0 raises AssertionError (:58).
    """
    with open(path) as handle:
        lines = [ln.strip() for ln in handle]
    merged = []
    pending = ""
    for lineno, text in enumerate(lines):
        if text[0] == ">":  # IndexError on blank lines, as the reference
            if pending:
                merged.append(pending.upper())
                pending = ""
            elif lineno != 0:
                raise AssertionError(
                    "There may be a header without a sequence at line {}.".format(lineno)
                )
            merged.append(text)
        else:
            pending += text
    merged.append(pending.upper())
    return merged[0::2], merged[1::2]


# --------------------------------------------------------------------------
# Vocabulary — kmer_counts.py:120-122
# --------------------------------------------------------------------------
def kmer_vocabulary(k, alphabet="AGTC"):
    """List of k-mers in itertools.product order and the kmer -> column map."""
    words = ["".join(t) for t in product(alphabet, repeat=k)]
    return words, {w: i for i, w in enumerate(words)}


# --------------------------------------------------------------------------
# Counting — kmer_counts.py:140-151
# --------------------------------------------------------------------------
def occurrences_py(row, seq, k, column_of):
    """Structure-faithful restatement of BasicCounter.occurrences (:140-151).

    Every one of the W = len(seq)-k+1 windows adds 1000/W (Python float64) to a
    dict keyed by the substring; afterwards only substrings found in the
    vocabulary are *assigned* into `row` (so windows holding a non-alphabet
    character are dropped but still counted in W, and bins that never occur
    keep whatever `row` held).  len(seq) == k-1 raises ZeroDivisionError.
    """
    tally = defaultdict(int)
    n_windows = len(seq) - k + 1
    step = 1000 / n_windows
    for start in range(n_windows):
        tally[seq[start:start + k]] += step
    for word, value in tally.items():
        col = column_of.get(word)
        if col is not None:
            row[col] = value
    return row


def count_kmers_u32(seqs, k, alphabet="AGTC"):
    """Integer surface: n[i, j] = number of windows of seqs[i] equal to k-mer j.

    Same definition as :142-150 (index = sum code(c_p) * A^(k-1-p), code =
    position in `alphabet`; a window with any non-alphabet character is
    skipped).  Vectorised with a rolling index + bincount; used for sizes where
    `occurrences_py` is too slow, and checked against it in the tests.
    """
    a = len(alphabet)
    n_cols = a ** k
    lut = np.full(256, -1, dtype=np.int64)
    for code, ch in enumerate(alphabet):
        lut[ord(ch)] = code  # later duplicates win, like the dict in :122
    out = np.zeros((len(seqs), n_cols), dtype=np.uint32)
    weights = a ** np.arange(k - 1, -1, -1, dtype=np.int64)
    for i, seq in enumerate(seqs):
        n_windows = len(seq) - k + 1
        if n_windows <= 0:
            continue
        raw = np.frombuffer(seq.encode("latin-1", "replace"), dtype=np.uint8)
        codes = lut[raw]
        win = np.lib.stride_tricks.sliding_window_view(codes, k)
        ok = (win >= 0).all(axis=1)
        idx = (win[ok] * weights).sum(axis=1)
        out[i] = np.bincount(idx, minlength=n_cols).astype(np.uint32)
    return out


def per_kb_from_counts(counts_u32, lengths, k, dtype=np.float32):
    """Per-kb matrix from integer counts exactly as :144-150 produce it.

    The reference adds `1000/W` (float64) to a dict entry n times and then
    stores the float64 sum into a float32 row.  We replay the n sequential
    float64 additions (table[n] = table[n-1] + step) and round once.
    len == k-1 -> ZeroDivisionError; len < k-1 -> all-zero row.
    """
    counts_u32 = np.asarray(counts_u32)
    out = np.zeros(counts_u32.shape, dtype=dtype)
    for i, length in enumerate(lengths):
        n_windows = int(length) - k + 1
        if n_windows == 0:
            raise ZeroDivisionError("division by zero")
        if n_windows < 0:
            continue
        step = 1000 / n_windows
        top = int(counts_u32[i].max()) if counts_u32.shape[1] else 0
        table = np.zeros(top + 1, dtype=np.float64)
        acc = 0
        for n in range(1, top + 1):
            acc += step
            table[n] = acc
        out[i] = table[counts_u32[i]].astype(dtype)
    return out


def raw_counts(seqs, k, alphabet="AGTC"):
    """float32 [N, A^k] raw per-kb matrix == get_counts() with mean=std=False, Log2.none."""
    n = count_kmers_u32(seqs, k, alphabet)
    return per_kb_from_counts(n, [len(s) for s in seqs], k)


def raw_counts_py(seqs, k, alphabet="AGTC"):
    """Same as `raw_counts` through the structure-faithful loop (:194-200)."""
    _, column_of = kmer_vocabulary(k, alphabet)
    out = np.zeros([len(seqs), len(alphabet) ** k], dtype=np.float32)
    for i, seq in enumerate(seqs):
        out[i] = occurrences_py(out[i], seq, k, column_of)
    return out


# --------------------------------------------------------------------------
# Column statistics — kmer_counts.py:165-187 (numpy axis-0 reduce order)
# --------------------------------------------------------------------------
def seqsum_f32(x):
    """Column sums of a float32 matrix with rows added strictly in index order.

    This is what `np.add.reduce(X, axis=0)` does for a C-contiguous float32
    matrix in the numpy build the reference runs on: one
    float32 accumulator per column, `acc = fl32(acc + X[i, j])` for i = 0..N-1.
    """
    x = np.asarray(x, dtype=np.float32)
    acc = np.zeros(x.shape[1], dtype=np.float32)
    for i in range(x.shape[0]):
        acc = acc + x[i]  # float32 + float32 -> float32, one rounding per row
    return acc


def column_mean_f32(x):
    """np.mean(x, axis=0) restated: seqsum / N in float32 (:168)."""
    return (seqsum_f32(x) / np.float32(x.shape[0])).astype(np.float32)


def column_std_f32(x):
    """np.std(x, axis=0) restated (:174): population std, every step float32."""
    n = np.float32(x.shape[0])
    m = (seqsum_f32(x) / n).astype(np.float32)
    d = (x - m).astype(np.float32)
    d = (d * d).astype(np.float32)
    v = (seqsum_f32(d) / n).astype(np.float32)
    return np.sqrt(v).astype(np.float32)


def center(x, mean=True):
    """:165-169.  Returns (x_centred float32, mean vector used)."""
    if mean is True:
        mean = column_mean_f32(x)
    y = np.array(x, dtype=np.float32, copy=True)
    y -= mean  # f64/int vectors: computed in the promoted type, rounded to f32
    return y, mean


def standardize(x, std=True):
    """:171-175.  Returns (x_scaled float32, std vector used)."""
    if std is True:
        std = column_std_f32(x)
    y = np.array(x, dtype=np.float32, copy=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        y /= std
    return y, std


def log2_plus_one(x):
    """:189-192: counts += 1; counts = log2(counts)."""
    y = np.array(x, dtype=np.float32, copy=True)
    y += 1
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log2(y)


def normalize(raw, mean=True, std=True, log2="Log2.post"):
    """The pipeline of get_counts after the counting loop (:201-209).

    Returns (counts, mean, std) where mean/std are what the reference leaves in
    `counter.mean` / `counter.std`.
    """
    if log2 not in LOG2_MODES:
        raise ValueError("log2 must be one of ['Log2.pre', 'Log2.post', 'Log2.none']")
    x = np.array(raw, dtype=np.float32, copy=True)
    if log2 == "Log2.pre":
        x = log2_plus_one(x)
    if mean is not False:
        x, mean = center(x, mean)
    if std is not False:
        x, std = standardize(x, std)
    if log2 == "Log2.post":
        x += np.abs(np.min(x))  # NaN-propagating global minimum (:208)
        x = log2_plus_one(x)
    return x, mean, std


# --------------------------------------------------------------------------
# The same three methods on a hand-assigned matrix that is NOT float32 —
# kmer_counts.py:165-192 act on whatever dtype `self.counts` has
# (test_kmer_counts.py:44-90 assigns it by hand).  Restated step by step in the
# order numpy 2.x's `_methods._mean` / `_var` take for each dtype; pinned to
# the reference by tests/golden/g11_other_dtypes (test_oracle_golden.py).
# --------------------------------------------------------------------------
def seqsum_any(x, acc_dtype):
    """Rows added one after the other into one accumulator per column of type
    `acc_dtype` (np.add.reduce along axis 0 of a C-contiguous matrix), every
    addition rounded to that type."""
    acc = np.zeros(x.shape[1], dtype=acc_dtype)
    for i in range(x.shape[0]):
        acc = (acc + x[i].astype(acc_dtype)).astype(acc_dtype)
    return acc


def pairwise_sum_any(a, dt):
    """One call of numpy's float add loop over a contiguous run (numpy 2.2,
    _core/src/umath/loops_utils.h.src: *_pairwise_sum — numpy is the
    reference's dependency, not vendored in it): fewer than 8 values one after
    the other from 0; up to 128 in eight strided accumulators folded
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the leftovers; longer runs split
    at n/2 rounded down to a multiple of 8."""
    n = len(a)
    if n < 8:
        res = dt(0)
        for v in a:
            res = dt(res + v)
        return res
    if n <= 128:
        r = [dt(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = dt(r[j] + a[i + j])
            i += 8
        res = dt(dt(dt(r[0] + r[1]) + dt(r[2] + r[3])) + dt(dt(r[4] + r[5]) + dt(r[6] + r[7])))
        while i < n:
            res = dt(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return dt(pairwise_sum_any(a[:n2], dt) + pairwise_sum_any(a[n2:], dt))


NP_BUFSIZE = 8192  # np.getbufsize(): the reduction hands its inner loop at most this many elements per call


def reduces_column_by_column(x):
    """Does np.add.reduce(x, axis=0) make axis 0 its INNER loop?  When axis 0 is
    the faster axis (column-major: DataFrame.values of a read CSV) or there is
    one column."""
    return x.shape[0] >= 2 and (x.shape[1] == 1 or abs(x.strides[0]) < abs(x.strides[1]))


def colsum_any(x, acc_dtype):
    """np.add.reduce(x, axis=0, dtype=acc_dtype) in numpy's own order for x's
    LAYOUT: row after row (seqsum_any) or, reduces_column_by_column, every
    column in pieces of NP_BUFSIZE, each piece cast to the loop's type and added
    pairwise: out = acc(out + pairwise(piece)).  The half loop (HALF_add) keeps
    float32 accumulators inside a piece and rounds to half once per piece."""
    if not reduces_column_by_column(x):
        return seqsum_any(x, acc_dtype)
    acc = np.dtype(acc_dtype).type
    work = np.float32 if acc is np.float16 else acc
    with np.errstate(all="ignore"):
        out = np.empty(x.shape[1], dtype=acc)
        for j in range(x.shape[1]):
            col = np.ascontiguousarray(x[:, j]).astype(work)
            res = acc(0)
            for i0 in range(0, len(col), NP_BUFSIZE):
                res = acc(work(res) + pairwise_sum_any(col[i0:i0 + NP_BUFSIZE], work))
            out[j] = res
    return out


def column_mean_any(x):
    """np.mean(x, axis=0) (:168): float16 sums in float32, the quotient is taken
    in float64 (float32 array / intp scalar), stored as float32 and then cast
    to float16; integers are converted to float64; float64 is itself."""
    n = x.shape[0]
    with np.errstate(all="ignore"):
        if x.dtype == np.float16:
            s = colsum_any(x, np.float32)
            return (s.astype(np.float64) / n).astype(np.float32).astype(np.float16)
        if x.dtype == np.float32:  # float32 array / intp scalar: the quotient in float64, stored as float32
            return (colsum_any(x, np.float32).astype(np.float64) / n).astype(np.float32)
        return colsum_any(x, np.float64) / n


def column_std_any(x):
    """np.std(x, axis=0) (:174), numpy's `_var` step by step: float16 stays
    float16 at every step (each quotient by N taken in float64 and rounded once
    to half), float32 stays float32, everything else runs in float64.  The
    deviations keep x's layout ('K' order), so a column-major matrix has its
    squares added column by column too."""
    n = x.shape[0]
    dt = x.dtype.type if x.dtype in (np.float16, np.float32) else np.float64
    with np.errstate(all="ignore"):
        m = (colsum_any(x, dt).astype(np.float64) / n).astype(dt)
        d = (x - m).astype(dt)
        d = (d * d).astype(dt)
        v = (colsum_any(d, dt).astype(np.float64) / n).astype(dt)
        return np.sqrt(v)


def host_center(counts, mean=True):
    """:165-169 on `counts` IN PLACE; raises what numpy raises (float statistics
    do not cast into an integer matrix).  Returns the mean that was used — it
    replaces the attribute BEFORE the subtraction, so callers that want the
    reference's state after an exception pass a holder: see the tests."""
    if mean is True:
        mean = column_mean_any(counts)
    return mean, (lambda: np.subtract(counts, mean, out=counts))


def host_standardize(counts, std=True):
    """:171-175, same contract as host_center."""
    if std is True:
        std = column_std_any(counts)
    return std, (lambda: np.true_divide(counts, std, out=counts))


def host_log2_norm(counts):
    """:189-192: `counts += 1` in place (integers wrap), then a NEW array
    np.log2(counts) whose dtype follows numpy's loops."""
    np.add(counts, 1, out=counts)
    with np.errstate(all="ignore"):
        return np.log2(counts)


def get_counts(seqs, k=6, mean=True, std=True, log2="Log2.post", alphabet="AGTC"):
    """BasicCounter(...).get_counts() end to end (:194-209)."""
    if len(seqs) == 1 and std is True:
        raise ValueError("You cannot standardize a single sequence.")
    return normalize(raw_counts(seqs, k, alphabet), mean, std, log2)


# --------------------------------------------------------------------------
# Pearson — pearson.py:32-44
# --------------------------------------------------------------------------
def pearson(counts1, counts2, row_standardize=True):
    """pearson.py:35-41: row-standardise both operands, inner product / K."""
    c1 = np.asarray(counts1)
    c2 = np.asarray(counts2)
    with np.errstate(divide="ignore", invalid="ignore"):
        if row_standardize:
            c1 = (c1.T - np.mean(c1, axis=1)).T
            c1 = (c1.T / np.std(c1, axis=1)).T
            c2 = (c2.T - np.mean(c2, axis=1)).T
            c2 = (c2.T / np.std(c2, axis=1)).T
        return np.inner(c1, c2) / c1.shape[1]


def pearson_f64_truth(counts1, counts2, row_standardize=True):
    """Same formula evaluated in float64 — the yardstick for rounding error."""
    return pearson(np.asarray(counts1, dtype=np.float64),
                   np.asarray(counts2, dtype=np.float64), row_standardize)


# --------------------------------------------------------------------------
# Synthetic workloads — SURVEY §8(d)
# --------------------------------------------------------------------------
CHUNK = 10_000


def synthetic_codes(seed, n_seqs, length, start=0):
    """uint8 [n, L] base codes 0..3 (map through b"ACGT") for rows start..start+n.

    Chunked seeding (10 000 sequences per chunk, SeedSequence([seed, chunk])) so
    any prefix can be regenerated without generating the whole set.
    """
    out = np.empty((n_seqs, length), dtype=np.uint8)
    row = start
    filled = 0
    while filled < n_seqs:
        chunk = row // CHUNK
        first = chunk * CHUNK
        rng = np.random.default_rng(np.random.SeedSequence([seed, chunk]))
        block = rng.integers(0, 4, size=(CHUNK, length), dtype=np.uint8)
        take = min(CHUNK - (row - first), n_seqs - filled)
        out[filled:filled + take] = block[row - first:row - first + take]
        filled += take
        row += take
    return out


def codes_to_seqs(codes):
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    ascii_rows = letters[codes]
    return [r.tobytes().decode("ascii") for r in ascii_rows]
