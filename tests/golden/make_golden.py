#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run only in the build container, where the reference checkout is mounted
read-only at /root/reference:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -W ignore /root/repo/tests/golden/make_golden.py

It imports `seekr.kmer_counts.BasicCounter`, `seekr.pearson.pearson` and
`seekr.fasta_reader.Reader` from the reference, feeds them inputs that this
script builds itself (seeded, reproducible — the test-suite regenerates the same
inputs from the same seeds), and stores inputs + outputs as small .npz/.json
files.  The reference's own *data* fixtures for the path
(seekr/tests/data/example*.npy, example_3mers_raw.csv) are stored as arrays in
`reference_fixtures.npz`.  No reference source text is copied.  The script is a
no-op when the reference is absent (GPU box).
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


sys.path.insert(0, HERE)
from inputs import EXAMPLE_FA, skewed_set, synth_2000, big_count_matrix, write_fasta  # noqa: E402


def main():
    if not os.path.isdir(os.path.join(REF, "seekr")):
        print("reference not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from seekr.kmer_counts import BasicCounter  # noqa: E402
    from seekr.pearson import pearson  # noqa: E402
    from seekr.fasta_reader import Reader  # noqa: E402

    tmp = tempfile.mkdtemp(prefix="golden_")

    def counter_for(seqs, **kw):
        c = BasicCounter(silent=True, **kw)
        c.seqs = list(seqs)
        return c

    def run(seqs, **kw):
        c = counter_for(seqs, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            c.get_counts()
        return c

    meta = {"numpy": np.__version__}

    # ---------------- reference data fixtures (data files of the reference's tests) ------
    d = os.path.join(REF, "seekr", "tests", "data")
    fx = {
        "example_2mers_counts": np.load(os.path.join(d, "example_2mers_counts.npy")),
        "example_2mers_count": np.load(os.path.join(d, "example_2mers_count.npy")),
        "example_mean": np.load(os.path.join(d, "example_mean.npy")),
        "example_std": np.load(os.path.join(d, "example_std.npy")),
        "example_3mers_raw_csv": np.loadtxt(os.path.join(d, "example_3mers_raw.csv"), delimiter=","),
    }
    with open(os.path.join(d, "example.fa")) as fh:
        assert fh.read() == EXAMPLE_FA, "example.fa text drifted"
    np.savez_compressed(os.path.join(HERE, "reference_fixtures.npz"), **fx)

    # ---------------- G1: example.fa through the reference -------------------------------
    fa = os.path.join(tmp, "example.fa")
    with open(fa, "w") as fh:
        fh.write(EXAMPLE_FA)
    ex_seqs = Reader(fa).get_seqs()
    g1 = {}
    for k in (1, 2, 3):
        c = run(ex_seqs, k=k, mean=False, std=False, log2="Log2.none")
        g1["raw_k{}".format(k)] = c.counts
        meta["example_raw_k{}_sha".format(k)] = sha16(c.counts)
    for k in (1, 2):
        stats = run(ex_seqs, k=k, log2="Log2.none")
        g1["mean_none_k{}".format(k)] = stats.mean
        g1["std_none_k{}".format(k)] = stats.std
        for mode in ("Log2.post", "Log2.pre", "Log2.none"):
            tag = mode.split(".")[1]
            c = run(ex_seqs, k=k, log2=mode)
            g1["full_{}_k{}".format(tag, k)] = c.counts
            g1["full_{}_k{}_mean".format(tag, k)] = c.mean
            g1["full_{}_k{}_std".format(tag, k)] = c.std
            c = run(ex_seqs, k=k, log2=mode, mean=stats.mean.copy(), std=stats.std.copy())
            g1["vec_{}_k{}".format(tag, k)] = c.counts
            c = run(ex_seqs, k=k, log2=mode, mean=True, std=False)
            g1["meanonly_{}_k{}".format(tag, k)] = c.counts
    np.savez_compressed(os.path.join(HERE, "g1_example.npz"), **g1)

    # ---------------- G3: realistic-length skewed sets, counts + pearson ------------------
    s1 = skewed_set(101, 111)
    s2 = skewed_set(202, 151)
    g3 = {}
    for k in (4, 5, 6):
        nv = run(s1, k=k, log2="Log2.post")
        c1 = run(s1, k=k, log2="Log2.post", mean=nv.mean.copy(), std=nv.std.copy())
        c2 = run(s2, k=k, log2="Log2.post", mean=nv.mean.copy(), std=nv.std.copy())
        with np.errstate(all="ignore"):
            r = pearson(c1.counts, c2.counts)
        if k == 4:
            g3["s1_raw_k4"] = run(s1, k=4, mean=False, std=False, log2="Log2.none").counts
            g3["s1_counts_k4"] = c1.counts
            g3["s1_self_default_k4"] = nv.counts
        if k < 6:
            g3["mean_k{}".format(k)] = nv.mean
            g3["std_k{}".format(k)] = nv.std
            g3["pearson_k{}".format(k)] = r
        else:
            meta["g3_k6_all_nan"] = bool(np.isnan(r).all())
            meta["g3_k6_counts_all_nan"] = bool(np.isnan(c1.counts).all())
    # row_standardize=False and mixed dtypes
    a = g3["s1_counts_k4"][:7]
    b = g3["s1_counts_k4"][7:12]
    g3["pearson_nostd"] = pearson(a, b, row_standardize=False)
    g3["pearson_f64"] = pearson(a.astype(np.float64), b.astype(np.float64))
    g3["pearson_mixed"] = pearson(a, b.astype(np.float64))
    meta["pearson_dtypes"] = {
        "f32": str(pearson(a, b).dtype), "f64": str(g3["pearson_f64"].dtype),
        "mixed": str(g3["pearson_mixed"].dtype),
        "int": str(pearson(np.arange(12).reshape(3, 4) % 5, np.arange(8).reshape(2, 4) % 3).dtype),
    }
    np.savez_compressed(os.path.join(HERE, "g3_skewed.npz"), **g3)

    # ---------------- G4: synthetic 2000 x 2 kb, k=6 ---------------------------------------
    syn = synth_2000()
    g4 = {}
    raw = run(syn, k=6, mean=False, std=False, log2="Log2.none").counts
    w = 2000 - 6 + 1
    n_int = np.rint(raw.astype(np.float64) * w / 1000.0).astype(np.uint32)
    meta["g4_raw_sha"] = sha16(raw)
    meta["g4_u32_sha"] = sha16(n_int)
    meta["g4_u32_sum"] = int(n_int.sum())
    for mode in ("Log2.post", "Log2.pre", "Log2.none"):
        tag = mode.split(".")[1]
        c = run(syn, k=6, log2=mode)
        g4["mean_" + tag] = c.mean
        g4["std_" + tag] = c.std
        g4["counts_{}_head".format(tag)] = c.counts[:8].copy()
        meta["g4_counts_{}_sha".format(tag)] = sha16(c.counts)
        g4["pearson256_" + tag] = pearson(c.counts[:256], c.counts[:256])
    np.savez_compressed(os.path.join(HERE, "g4_synth2000.npz"), **g4)

    # ---------------- G5: large-N float32 drift pin ----------------------------------------
    big = big_count_matrix()
    c = counter_for(["ACGT"] * 2, k=6)
    c.counts = big.copy()
    c.center()
    mean_big = c.mean.copy()
    c.standardize()
    std_big = c.std.copy()
    g5 = {"mean": mean_big, "std": std_big, "z_row0": c.counts[0].copy(),
          "z_rowlast": c.counts[-1].copy(), "z_min": np.min(c.counts)}
    truth_mean = big.astype(np.float64).mean(axis=0)
    meta["g5_mean_drift_vs_f64"] = float(np.max(np.abs(mean_big - truth_mean) / truth_mean))
    np.savez_compressed(os.path.join(HERE, "g5_bigN.npz"), **g5)
    del big, c

    # ---------------- G6: edge cases ---------------------------------------------------------
    g6 = {}
    edge = {}
    # non-alphabet chars, lower case assigned directly, U, short sequences
    edge_seqs = ["ACGTNACGT", "acgtACGT", "ACGUACGU", "ACG", "ACGTACGTACNNNNNNACGTTTGA", "A", ""]
    c = run(edge_seqs, k=3, mean=False, std=False, log2="Log2.none")
    g6["edge_raw_k3"] = c.counts
    edge["edge_seqs"] = edge_seqs
    try:
        run(["ACGTAC", "AC"], k=3, mean=False, std=False, log2="Log2.none")
        edge["len_eq_k_minus_1"] = "no error"
    except ZeroDivisionError as e:
        edge["len_eq_k_minus_1"] = "ZeroDivisionError: " + str(e)
    # alphabet permutation
    c = run(ex_seqs, k=2, mean=False, std=False, log2="Log2.none", alphabet="ACGT")
    g6["raw_k2_ACGT"] = c.counts
    # reader: multi-line, CRLF, lower case, no trailing newline
    rs = skewed_set(303, 6, 50, 200)
    p1 = os.path.join(tmp, "ml.fa")
    write_fasta(p1, rs, width=60, crlf=True, lower=True)
    with open(p1, "rb+") as fh:  # drop the final newline
        fh.seek(-2, os.SEEK_END)
        fh.truncate()
    rd = Reader(p1)
    got = rd.get_seqs()
    edge["reader_multiline_ok"] = got == rs
    edge["reader_headers"] = Reader(p1).get_headers()
    for name, text in (("blank_line", ">a\nACGT\n\n>b\nACGT\n"), ("double_header", ">a\nACGT\n>b\n>c\nACGT\n")):
        p = os.path.join(tmp, name + ".fa")
        with open(p, "w") as fh:
            fh.write(text)
        try:
            Reader(p).get_seqs()
            edge["reader_" + name] = "no error"
        except Exception as e:  # noqa: BLE001
            edge["reader_" + name] = type(e).__name__ + ": " + str(e)
    # single sequence with std=True
    try:
        c1 = BasicCounter(fa, k=2)
        c1.seqs = ["ACGT"]
        edge["single_seq_note"] = "constructor checks only seqs read from infasta"
    except ValueError as e:
        edge["single_seq"] = str(e)
    p = os.path.join(tmp, "one.fa")
    with open(p, "w") as fh:
        fh.write(">a\nACGTACGT\n")
    try:
        BasicCounter(p, k=2)
        edge["single_seq_infasta"] = "no error"
    except ValueError as e:
        edge["single_seq_infasta"] = "ValueError: " + str(e)
    try:
        BasicCounter(fa, k=2, log2="log2")
        edge["bad_log2"] = "no error"
    except ValueError as e:
        edge["bad_log2"] = "ValueError: " + str(e)
    # zero-variance column -> NaN + printed warning
    buf = io.StringIO()
    c = counter_for(ex_seqs, k=3, log2="Log2.none")
    with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
        c.get_counts()
    edge["nan_warning_text"] = buf.getvalue()
    g6["example_k3_none_with_nan"] = c.counts
    c = counter_for(ex_seqs, k=3, log2="Log2.post")
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        c.get_counts()
    g6["example_k3_post_with_nan"] = c.counts
    # float64 / int user vectors
    c = counter_for(ex_seqs, k=1, log2="Log2.none", mean=np.array([100.0, 200.5, 300.25, 50.125]),
                    std=np.array([3, 7, 11, 13]))
    c.get_counts()
    g6["user_vec_f64_int_k1"] = c.counts
    # constant row in pearson -> NaN row
    m = np.array([[1, 2, 3, 4], [5, 5, 5, 5], [4, 1, 3, 2]], dtype=np.float32)
    with np.errstate(all="ignore"):
        g6["pearson_const_row"] = pearson(m, m)
    try:
        pearson(np.zeros((2, 4), np.float32), np.zeros((2, 5), np.float32), row_standardize=False)
        edge["pearson_col_mismatch"] = "no error"
    except ValueError as e:
        edge["pearson_col_mismatch"] = "ValueError"
    # .npy suffix rule and save modes
    out = os.path.join(tmp, "counts.seekr")
    c = BasicCounter(fa, outfile=out, k=2, binary=True, label=False, silent=True)
    with contextlib.redirect_stdout(io.StringIO()):
        c.make_count_file()
    edge["npy_suffix_written"] = sorted(f for f in os.listdir(tmp) if f.startswith("counts.seekr"))
    out = os.path.join(tmp, "plain.csv")
    c = BasicCounter(fa, outfile=out, k=2, binary=False, label=False, silent=True, mean=False, std=False,
                     log2="Log2.none")
    c.make_count_file()
    with open(out) as fh:
        edge["plain_csv_first_line"] = fh.readline().strip()
    out = os.path.join(tmp, "label.csv")
    c = BasicCounter(fa, outfile=out, k=1, binary=False, label=True, silent=True, mean=False, std=False,
                     log2="Log2.none")
    c.make_count_file()
    with open(out) as fh:
        edge["label_csv_text"] = fh.read()
    try:
        BasicCounter(fa, outfile=out, k=1, binary=True, label=True, silent=True).save()
        edge["binary_and_label"] = "no error"
    except AssertionError as e:
        edge["binary_and_label"] = "AssertionError"
    # known-answer literals of the reference's own tests, evaluated through the reference
    k1 = counter_for(ex_seqs, k=1)
    g6["kat_occ_k1_seq0"] = k1.occurrences(np.zeros(4), ex_seqs[0])
    g6["kat_occ_k1_seq1"] = k1.occurrences(np.zeros(4), ex_seqs[1])
    k2 = counter_for(ex_seqs, k=2)
    g6["kat_occ_k2_seq1"] = k2.occurrences(np.zeros(16), ex_seqs[1])
    g6["kat_pearson_int"] = pearson(
        np.array([[8, 5, 6, 9, 2], [8, 3, 6, 6, 7], [7, 7, 3, 3, 7]]),
        np.array([[2, 8, -9, -1, -8], [-4, 1, 2, -1, 2], [5, -3, -7, 2, -9]]))
    np.savez_compressed(os.path.join(HERE, "g6_edges.npz"), **g6)
    meta["edge"] = edge

    with open(os.path.join(HERE, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in meta.items() if k != "edge"}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
