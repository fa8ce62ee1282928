"""Pins the CPU oracle (oracle/seekr_oracle.py) to the reference.

Sources of truth, all under tests/golden/ (see make_golden.py):
  * reference_fixtures.npz  — data files the reference's own tests hold
  * g1..g6 *.npz, meta.json — outputs of the reference run on seeded inputs
  * literals re-typed from seekr/tests/test_kmer_counts.py / test_pearson.py
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import seekr_oracle as orc
from inputs import EXAMPLE_FA, big_count_matrix, skewed_set, synth_2000, write_fasta


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def gold(golden_dir):
    out = {name: np.load(os.path.join(golden_dir, name + ".npz")) for name in
           ("reference_fixtures", "g1_example", "g3_skewed", "g4_synth2000", "g5_bigN", "g6_edges")}
    with open(os.path.join(golden_dir, "meta.json")) as fh:
        out["meta"] = json.load(fh)
    return out


@pytest.fixture(scope="module")
def example_seqs(tmp_path_factory):
    p = tmp_path_factory.mktemp("fa") / "example.fa"
    p.write_text(EXAMPLE_FA)
    headers, seqs = orc.read_fasta(str(p))
    assert headers == [">SEQ1", ">SEQ2", ">SEQ3", ">SEQ4", ">SEQ5"]
    return seqs


# ------------------------- the reference's own known-answer tests ----------------------
def test_reader_init(example_seqs):  # test_kmer_counts.py:13-16
    assert len(example_seqs) == 5 and example_seqs[0] == "AAAAAA"


def test_occurrences_literals(example_seqs, gold):  # test_kmer_counts.py:18-42
    _, m1 = orc.kmer_vocabulary(1)
    row = orc.occurrences_py(np.zeros(4), example_seqs[0], 1, m1)
    assert np.allclose(row, [1000, 0, 0, 0])
    row = orc.occurrences_py(np.zeros(4), example_seqs[1], 1, m1)
    assert np.allclose(row, [0, 500, 500, 0])
    _, m2 = orc.kmer_vocabulary(2)
    row = orc.occurrences_py(np.zeros(16), example_seqs[1], 2, m2)
    exp = np.zeros(16)
    exp[5], exp[9], exp[10] = 454.545, 90.909, 454.545
    assert np.allclose(row, exp)
    g6 = gold["g6_edges"]
    assert np.array_equal(row, g6["kat_occ_k2_seq1"])
    assert np.array_equal(orc.occurrences_py(np.zeros(4), example_seqs[1], 1, m1), g6["kat_occ_k1_seq1"])


def test_center_standardize_log2_literals():  # test_kmer_counts.py:44-90
    x = np.array([[1, 2, 3, 4], [1, -2, 5, 10]], dtype=np.float32)
    y, _ = orc.center(x)
    assert np.allclose(y, [[0, 2, -1, -3], [0, -2, 1, 3]])
    mean = np.ones(4)
    mean[3] = -1
    y, _ = orc.center(x, mean)
    assert np.allclose(y, [[0, 1, 2, 5], [0, -3, 4, 11]])
    x = np.array([[1, 2, 3, 4], [0, -2, 5, 10]], dtype=np.float32)
    y, _ = orc.standardize(x)
    assert np.allclose(y, [[2, 1, 3, 4 / 3], [0, -1, 5, 10 / 3]])
    y, _ = orc.standardize(x, np.arange(1, 5))
    assert np.allclose(y, [[1, 1, 1, 1], [0, -1, 5 / 3, 2.5]])
    assert np.allclose(orc.log2_plus_one(x + 2), np.log2(x + 3))


def test_get_counts_k1_literal(example_seqs):  # test_kmer_counts.py:92-106
    expected = np.array([[2.1798673, 0.27807194, 0.0, 0.5133058],
                         [0.6370419, 2.1100981, 2.048016, 0.5133058],
                         [1.2010899, 1.4672222, 1.3604679, 1.8107259],
                         [1.2073011, 1.3895708, 1.3721647, 1.8666755],
                         [1.318994, 1.1856667, 1.5349197, 1.6688585]], dtype=np.float32)
    got, _, _ = orc.get_counts(example_seqs, k=1)
    assert np.allclose(got, expected, rtol=1e-4, atol=1e-5)


def test_pearson_literals(gold):  # test_pearson.py:7-24
    c1 = np.array([[8, 5, 6, 9, 2], [8, 3, 6, 6, 7], [7, 7, 3, 3, 7]])
    c2 = np.array([[2, 8, -9, -1, -8], [-4, 1, 2, -1, 2], [5, -3, -7, 2, -9]])
    exp = np.array([[0.3217847, -0.71611487, 0.85110363],
                    [-0.52756992, -0.47172818, 0.22652512],
                    [0.43762719, -0.17902872, 0.01547461]])
    got = orc.pearson(c1, c2)
    assert np.allclose(got, exp)
    assert np.array_equal(got, gold["g6_edges"]["kat_pearson_int"])
    one = np.array([[1, 2, 3, 4], [2, 4, 6, 8]])
    assert np.allclose(orc.pearson(one, one), np.ones((2, 2)))


def test_reference_data_fixtures(example_seqs, gold):  # test_console_scripts.py:34-124
    fx = gold["reference_fixtures"]
    full, _, _ = orc.get_counts(example_seqs, k=2)
    assert np.allclose(full, fx["example_2mers_counts"])
    raw3 = orc.raw_counts(example_seqs, 3)
    assert np.allclose(raw3, fx["example_3mers_raw_csv"])
    # %1.6f text round trip is how the reference's CSV test compares
    txt = np.array([[float("%1.6f" % v) for v in r] for r in raw3])
    assert np.array_equal(txt, fx["example_3mers_raw_csv"])
    vec, _, _ = orc.get_counts(example_seqs, k=2, mean=fx["example_mean"], std=fx["example_std"])
    assert np.allclose(vec, fx["example_2mers_count"])
    _, mean, std = orc.get_counts(example_seqs, k=2, log2="Log2.none")
    assert np.allclose(mean, fx["example_mean"]) and np.allclose(std, fx["example_std"])


# ------------------------- vectors captured from the reference -------------------------
def test_g1_example_bitexact(example_seqs, gold):
    g1 = gold["g1_example"]
    for k in (1, 2, 3):
        raw = orc.raw_counts(example_seqs, k)
        assert np.array_equal(bits(raw), bits(g1["raw_k%d" % k])), k
        assert np.array_equal(bits(orc.raw_counts_py(example_seqs, k)), bits(raw))
        assert sha16(raw) == gold["meta"]["example_raw_k%d_sha" % k]
    assert orc.count_kmers_u32(example_seqs, 2).sum(axis=1).tolist() == [5, 11, 15, 74, 75]
    for k in (1, 2):
        for tag in ("post", "pre", "none"):
            mode = "Log2." + tag
            with np.errstate(all="ignore"):
                x, mean, std = orc.get_counts(example_seqs, k=k, log2=mode)
                assert np.array_equal(bits(mean), bits(g1["full_%s_k%d_mean" % (tag, k)]))
                assert np.array_equal(bits(std), bits(g1["full_%s_k%d_std" % (tag, k)]))
                assert np.array_equal(bits(x), bits(g1["full_%s_k%d" % (tag, k)])), (tag, k)
                v, _, _ = orc.get_counts(example_seqs, k=k, log2=mode, mean=g1["mean_none_k%d" % k],
                                         std=g1["std_none_k%d" % k])
                assert np.array_equal(bits(v), bits(g1["vec_%s_k%d" % (tag, k)]))
                mo, _, _ = orc.get_counts(example_seqs, k=k, log2=mode, mean=True, std=False)
                assert np.array_equal(bits(mo), bits(g1["meanonly_%s_k%d" % (tag, k)]))


def test_g3_skewed_sets(gold):
    g3 = gold["g3_skewed"]
    s1, s2 = skewed_set(101, 111), skewed_set(202, 151)
    assert np.array_equal(bits(orc.raw_counts(s1, 4)), bits(g3["s1_raw_k4"]))
    for k in (4, 5):
        x, mean, std = orc.get_counts(s1, k=k)
        assert np.array_equal(bits(mean), bits(g3["mean_k%d" % k]))
        assert np.array_equal(bits(std), bits(g3["std_k%d" % k]))
        c1, _, _ = orc.get_counts(s1, k=k, mean=mean, std=std)
        c2, _, _ = orc.get_counts(s2, k=k, mean=mean, std=std)
        if k == 4:
            assert np.array_equal(bits(x), bits(g3["s1_self_default_k4"]))
            assert np.array_equal(bits(c1), bits(g3["s1_counts_k4"]))
        r = orc.pearson(c1, c2)
        assert r.dtype == np.float32 and r.shape == (111, 151)
        assert np.allclose(r, g3["pearson_k%d" % k], rtol=1e-5, atol=2e-6)
    with np.errstate(all="ignore"):
        x6, m6, s6 = orc.get_counts(s1, k=6)
        c6, _, _ = orc.get_counts(s2, k=6, mean=m6, std=s6)
        assert np.isnan(x6).all() and np.isnan(c6).all() and np.isnan(orc.pearson(x6, c6)).all()
    a, b = g3["s1_counts_k4"][:7], g3["s1_counts_k4"][7:12]
    assert np.allclose(orc.pearson(a, b, row_standardize=False), g3["pearson_nostd"], rtol=1e-5, atol=2e-6)
    assert np.allclose(orc.pearson(a.astype(np.float64), b.astype(np.float64)), g3["pearson_f64"],
                       rtol=1e-12, atol=1e-14)
    assert orc.pearson(a, b.astype(np.float64)).dtype == np.float64


def test_g4_synthetic_2000(gold):
    g4, meta = gold["g4_synth2000"], gold["meta"]
    seqs = synth_2000()
    n = orc.count_kmers_u32(seqs, 6)
    assert sha16(n) == meta["g4_u32_sha"] and int(n.sum()) == meta["g4_u32_sum"] == 2000 * 1995
    raw = orc.per_kb_from_counts(n, [2000] * 2000, 6)
    assert sha16(raw) == meta["g4_raw_sha"]
    # structure-faithful python loop on a prefix (the slow path)
    assert np.array_equal(bits(orc.raw_counts_py(seqs[:40], 6)), bits(raw[:40]))
    for tag in ("post", "pre", "none"):
        x, mean, std = orc.normalize(raw, log2="Log2." + tag)
        assert np.array_equal(bits(mean), bits(g4["mean_" + tag])), tag
        assert np.array_equal(bits(std), bits(g4["std_" + tag])), tag
        assert np.array_equal(bits(x[:8]), bits(g4["counts_%s_head" % tag])), tag
        assert sha16(x) == meta["g4_counts_%s_sha" % tag]
        assert np.allclose(orc.pearson(x[:256], x[:256]), g4["pearson256_" + tag], rtol=1e-5, atol=2e-6)


def test_g5_large_n_rowsequential_drift(gold):
    """np.mean/np.std(axis=0) in the reference are strictly row-sequential float32 sums."""
    g5 = gold["g5_bigN"]
    big = big_count_matrix()
    mean = orc.column_mean_f32(big)
    assert np.array_equal(bits(mean), bits(g5["mean"]))
    big -= mean
    std = orc.column_std_f32(big)
    assert np.array_equal(bits(std), bits(g5["std"]))
    assert np.array_equal(bits(big[0] / std), bits(g5["z_row0"]))
    assert np.array_equal(bits(big[-1] / std), bits(g5["z_rowlast"]))
    # the same numpy build is on this box: the restatement must equal numpy's own reduce
    assert np.array_equal(bits(np.std(big, axis=0)), bits(std))
    # monotone-rounding shortcut used on the GPU: global min of z from per-column minima
    with np.errstate(all="ignore"):
        zmin = np.min((big.min(axis=0) / std).astype(np.float32))
    assert np.float32(zmin) == g5["z_min"]


def test_g6_edges(example_seqs, gold, tmp_path):
    g6, edge = gold["g6_edges"], gold["meta"]["edge"]
    raw = orc.raw_counts(edge["edge_seqs"], 3)
    assert np.array_equal(bits(raw), bits(g6["edge_raw_k3"]))
    assert np.array_equal(bits(orc.raw_counts_py(edge["edge_seqs"], 3)), bits(raw))
    with pytest.raises(ZeroDivisionError):
        orc.raw_counts(["ACGTAC", "AC"], 3)
    with pytest.raises(ZeroDivisionError):
        orc.raw_counts_py(["ACGTAC", "AC"], 3)
    assert np.array_equal(bits(orc.raw_counts(example_seqs, 2, "ACGT")), bits(g6["raw_k2_ACGT"]))
    # reader semantics
    rs = skewed_set(303, 6, 50, 200)
    p = str(tmp_path / "ml.fa")
    write_fasta(p, rs, width=60, crlf=True, lower=True)
    headers, seqs = orc.read_fasta(p)
    assert seqs == rs and headers == edge["reader_headers"]
    for name, text, exc in (("blank_line", ">a\nACGT\n\n>b\nACGT\n", IndexError),
                            ("double_header", ">a\nACGT\n>b\n>c\nACGT\n", AssertionError)):
        q = tmp_path / (name + ".fa")
        q.write_text(text)
        with pytest.raises(exc) as info:
            orc.read_fasta(str(q))
        assert edge["reader_" + name] == type(info.value).__name__ + ": " + str(info.value)
    with np.errstate(all="ignore"):
        x, _, _ = orc.get_counts(example_seqs, k=3, log2="Log2.none")
        assert np.array_equal(bits(x), bits(g6["example_k3_none_with_nan"]))
        x, _, _ = orc.get_counts(example_seqs, k=3, log2="Log2.post")
        assert np.isnan(x).all() and np.isnan(g6["example_k3_post_with_nan"]).all()
        x, _, _ = orc.get_counts(example_seqs, k=1, log2="Log2.none",
                                 mean=np.array([100.0, 200.5, 300.25, 50.125]), std=np.array([3, 7, 11, 13]))
        assert np.array_equal(bits(x), bits(g6["user_vec_f64_int_k1"]))
        m = np.array([[1, 2, 3, 4], [5, 5, 5, 5], [4, 1, 3, 2]], dtype=np.float32)
        r = orc.pearson(m, m)
        assert np.array_equal(np.isnan(r), np.isnan(g6["pearson_const_row"]))
        assert np.allclose(r, g6["pearson_const_row"], equal_nan=True)
    with pytest.raises(ValueError):
        orc.pearson(np.zeros((2, 4), np.float32), np.zeros((2, 5), np.float32), row_standardize=False)
    with pytest.raises(ValueError):
        orc.normalize(np.zeros((2, 4), np.float32), log2="log2")


def test_c_restatement_matches_numpy_oracle_and_golden(gold):
    """oracle/seekr_oracle.c (used for full-size checks) is pinned the same way."""
    from oracle import c_oracle as co
    meta = gold["meta"]
    seqs = synth_2000()
    blob, off = co.seqs_to_blob(seqs)
    n = co.count_u32(blob, off, 6)
    assert sha16(n) == meta["g4_u32_sha"]
    raw = co.per_kb_f32(n, np.diff(off), 6)
    assert sha16(raw) == meta["g4_raw_sha"]
    edge = meta["edge"]["edge_seqs"]
    blob, off = co.seqs_to_blob(edge)
    n3 = co.count_u32(blob, off, 3)
    assert np.array_equal(n3, orc.count_kmers_u32(edge, 3))
    assert np.array_equal(bits(co.per_kb_f32(n3, np.diff(off), 3)), bits(gold["g6_edges"]["edge_raw_k3"]))
    with pytest.raises(ZeroDivisionError):
        co.per_kb_f32(n3, np.diff(off) * 0 + 2, 3)
    big = big_count_matrix(seed=9, n=5000, k_cols=512)
    assert np.array_equal(bits(co.colsum_seq_f32(big)), bits(orc.seqsum_f32(big)))
    assert np.array_equal(bits(co.colsum_seq_f32(big) / np.float32(5000)), bits(np.mean(big, axis=0)))


def test_synthetic_prefix_property():
    a = orc.synthetic_codes(2, 25, 100, start=9_990)
    b = orc.synthetic_codes(2, 10_015, 100)
    assert np.array_equal(a, b[9_990:])


def test_g7_alphabets_other_than_four_letters(golden_dir):
    """The oracle on alphabets of 1, 2, 5 and 20 letters and with a repeated letter, against the reference's
    own output (tests/golden/make_golden_g7.py)."""
    from make_golden_g7 import CASES, sequences
    g7 = np.load(os.path.join(golden_dir, "g7_alphabets.npz"))
    for name, alphabet, k, letters in CASES:
        seqs = sequences(name, letters)
        raw = orc.raw_counts_py(seqs, k, alphabet)
        assert np.array_equal(raw.view(np.uint32), g7[name + "_raw"].view(np.uint32)), name
        assert np.array_equal(orc.raw_counts(seqs, k, alphabet).view(np.uint32), raw.view(np.uint32)), name
        with np.errstate(all="ignore"):
            pre = orc.normalize(raw, True, False, "Log2.pre")[0]
            post = orc.normalize(raw, True, True, "Log2.post")[0]
        np.testing.assert_array_equal(pre, g7[name + "_pre"])
        np.testing.assert_array_equal(post, g7[name + "_post"])


def test_g9_wide_rows_of_other_alphabets(golden_dir):
    """G9 (make_golden_g9.py): 10 letters at k = 4 (10 000 columns), ACGTN at k = 6 (15 625), 7 letters at k = 5 (16 807) through
    the REFERENCE — the oracle's raw counts bit-exact, its mean-centred Log2.post matrix on 64 seeded cells and in its sum,
    its Pearson of both within the north_star's bar of the reference's."""
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden_g9 as mk
    g9 = np.load(os.path.join(golden_dir, "g9_wide_alphabets.npz"))
    for name, alphabet, k in mk.CASES:
        seqs = mk.sequences(name, alphabet)
        raw = orc.raw_counts(seqs, k, alphabet)
        assert np.array_equal(raw.view(np.uint32), g9[name + "_raw"].view(np.uint32)), name
        post = orc.normalize(raw, mean=True, std=False, log2="Log2.post")[0]
        rows, cols = mk.sampled_cells(name, post.shape)
        np.testing.assert_allclose(post[rows, cols], g9[name + "_post_cells"], rtol=1e-6, atol=1e-7)
        assert abs(post.astype(np.float64).sum() - float(g9[name + "_post_sum"])) <= 1e-6 * abs(float(g9[name + "_post_sum"]))
        for tag, m in (("raw", raw), ("post", post)):
            want = g9["%s_r_%s" % (name, tag)].astype(np.float64)
            got = orc.pearson(m, m).astype(np.float64)
            assert (np.abs(got - want) <= 2e-6 + 1e-5 * np.abs(want)).all(), (name, tag)


def test_g8_file_whose_first_line_is_not_a_header(golden_dir, tmp_path):
    """fasta_reader.py:47-63 keeps entries in encounter order and :70-78 slice them as they stand: the oracle and the
    package's Reader must return the reference's (odd-looking) lists for such a file."""
    import json
    from seekr_amd.fasta_reader import Reader
    g = json.load(open(os.path.join(golden_dir, "g8_headerless.json")))
    path = str(tmp_path / "h.fa")
    with open(path, "w") as fh:
        fh.write(g["text"])
    assert orc.read_fasta(path) == (g["headers"], g["seqs"])
    assert Reader(path).get_lines() == g["lines"] and Reader(path).get_headers() == g["headers"]
    assert Reader(path).get_seqs() == g["seqs"] == g["counter_seqs"]
    raw = orc.raw_counts(g["seqs"], 2)
    assert np.array_equal(raw.view(np.uint32), np.array(g["raw_k2_bits"], dtype=np.uint32))


def g10_cases(golden_dir):
    """The non-ASCII FASTA files of tests/golden/g10_non_ascii.json (made by the reference under UTF-8 text decoding)."""
    import json
    import locale
    g = json.load(open(os.path.join(golden_dir, "g10_non_ascii.json")))
    if locale.getpreferredencoding(False).lower().replace("-", "") != g["text_encoding"].lower().replace("-", ""):
        pytest.skip("text-mode open() decodes with {} here, the fixture was made with {}".format(
            locale.getpreferredencoding(False), g["text_encoding"]))
    return g["cases"]


def test_g10_files_that_are_not_ascii(golden_dir, tmp_path):
    """fasta_reader.py:44 opens in text mode: strip / upper / len work on decoded characters.  The oracle's reader and
    the package's Reader return the reference's lists for 19 files with NBSP / NEL / U+2028 / U+3000 at line ends and
    inside lines, letters whose upper() is longer, astral characters, a BOM, CRLF; the structure-faithful counting loop
    on those strings gives the reference's raw k = 2 bits; and the files the reference fails on fail the same way."""
    from seekr_amd.fasta_reader import Reader
    builtin = {"IndexError": IndexError, "AssertionError": AssertionError, "ZeroDivisionError": ZeroDivisionError,
               "UnicodeDecodeError": UnicodeDecodeError}
    n_ok = n_err = 0
    for case in g10_cases(golden_dir):
        path = str(tmp_path / (case["name"] + ".fa"))
        with open(path, "wb") as fh:
            fh.write(bytes.fromhex(case["hex"]))
        if "exception" in case:
            for read in (lambda: orc.raw_counts_py(orc.read_fasta(path)[1], 2), lambda: orc.raw_counts_py(Reader(path).get_seqs(), 2)):
                with pytest.raises(builtin[case["exception"]]) as err:
                    read()
                assert str(err.value) == case["message"], case["name"]
            n_err += 1
            continue
        assert orc.read_fasta(path) == (case["headers"], case["seqs"]), case["name"]
        assert Reader(path).get_headers() == case["headers"] and Reader(path).get_seqs() == case["seqs"], case["name"]
        assert [len(s) for s in case["seqs"]] == case["lengths"]
        raw = orc.raw_counts_py(case["seqs"], 2)
        assert np.array_equal(raw.view(np.uint32), np.array(case["raw_k2_bits"], dtype=np.uint32)), case["name"]
        n_ok += 1
    assert n_ok >= 13 and n_err >= 6


def test_g10_native_parser_declines_every_file_with_a_high_byte(golden_dir, tmp_path):
    """skr_fasta_open (CPU code of the library) answers SKR_ERR_FASTA_TEXT for each G10 file — before any of its other
    checks — and still parses the ASCII twin of the file; BasicCounter turns the answer into the text-mode Reader."""
    from seekr_amd import _lib
    for case in g10_cases(golden_dir):
        data = bytes.fromhex(case["hex"])
        path = str(tmp_path / (case["name"] + ".fa"))
        with open(path, "wb") as fh:
            fh.write(data)
        with pytest.raises(_lib.FastaNeedsText):
            _lib.FastaFile(path)
    # the high byte far into a file cut into pieces, in the last piece only
    os.environ["SEEKR_FASTA_PIECE_BYTES"] = "64"
    try:
        body = "".join(">s%d\nACGTACGTAC\nGGTTAACC\n" % i for i in range(200))
        path = str(tmp_path / "late.fa")
        with open(path, "wb") as fh:
            fh.write(body.encode() + b">last\nACGT\xc2\xa0\n")
        with pytest.raises(_lib.FastaNeedsText):
            _lib.FastaFile(path)
        with open(path, "wb") as fh:
            fh.write(body.encode() + b">last\nACGT\x7f\n")
        fa = _lib.FastaFile(path)
        assert fa.n == 201 and int(fa.lengths()[-1]) == 5
    finally:
        del os.environ["SEEKR_FASTA_PIECE_BYTES"]


def test_g11_normalisation_methods_on_matrices_that_are_not_float32(golden_dir):
    """kmer_counts.py:165-192 on hand-assigned float64 / float16 / integer / bool matrices (7 x 5 and 3 000 x 256): the
    oracle's step-by-step restatement of numpy's `_mean` / `_var` per dtype (row-sequential sums; float16 mean in float32,
    float16 std in half steps; integers through float64) leaves the reference's bytes — mean, std, centred / scaled matrices
    exact, log2 within the bar — and numpy's exception where the reference gets one (with the attribute already replaced)."""
    import g11_cases
    assert g11_cases.check_all(golden_dir, g11_cases.run_oracle(orc)) == 9 * 2 * 16


def test_g12_column_major_matrices_as_the_reference_leaves_them(golden_dir):
    """kmer_counts.py:165-175 on column-major / strided column-major / single-column matrices of five dtypes (300 x 7 and
    9 001 x 3: past numpy's 8 192-element buffer piece): the oracle's restatement of numpy's column-by-column pairwise order
    leaves the reference's bytes for mean, std and the centred-then-standardised matrix, its dtypes, and numpy's exception
    (with the attribute already replaced) for integer matrices."""
    import g12_cases
    assert g12_cases.check_all(golden_dir, g12_cases.oracle_counter(orc)) == 5 * 2 * 3


def test_oracle_column_statistics_follow_numpys_order_for_every_layout():
    """np.mean / np.std(axis=0) of kmer_counts.py:168,174 depend on the matrix's LAYOUT: row after row for C order, every
    column pairwise in 8 192-element pieces for column-major matrices and single columns (float16: float32 accumulators
    within a piece, rounded to half once per piece; integers: cast to float64 piece by piece).  The oracle's restatement
    (colsum_any, pairwise_sum_any) against numpy itself — the reference's dependency — bit for bit, over the lengths where
    the pairwise tree changes shape and the buffer boundary, five layouts, eight dtypes."""
    rng = np.random.default_rng(2)
    n = 0
    for dt in (np.float16, np.float32, np.float64, np.int32, np.uint8, np.int64, np.bool_, np.uint64):
        for rows in (1, 2, 7, 8, 9, 24, 129, 1000, 8192, 8193, 20000):
            for cols in (1, 3):
                if np.dtype(dt).kind == "f":
                    base = (rng.poisson(2.0, size=(rows, cols)) * 0.5 + rng.standard_normal((rows, cols)) * 0.1).astype(dt)
                else:
                    base = rng.integers(0, 2 if dt is np.bool_ else 200, size=(rows, cols)).astype(dt)
                layouts = (base.copy(), np.asfortranarray(base), np.ascontiguousarray(base.T).T,
                           np.asfortranarray(np.repeat(base, 2, axis=0))[::2], np.repeat(base, 2, axis=1)[:, ::2])
                for x in layouts:
                    with np.errstate(all="ignore"):
                        assert orc.column_mean_any(x).tobytes() == np.mean(x, axis=0).tobytes(), (dt, rows, cols, x.strides)
                        assert orc.column_std_any(x).tobytes() == np.std(x, axis=0).tobytes(), (dt, rows, cols, x.strides)
                    n += 1
    assert n == 8 * 11 * 2 * 5


def test_numpy_adds_a_row_in_the_pairwise_order_the_fill_kernel_reproduces():
    """The order seekr_amd/csrc/operand.hip: np_pairwise_sum reproduces on the device, restated in Python and pinned
    against np.add.reduce itself (float32, every length up to 300 and the widths the product meets): fewer than 8 values
    one after the other; up to 128 in eight strided accumulators folded ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)) plus
    leftovers; longer rows split at n/2 rounded down to a multiple of 8.  And with it np.mean / np.std(axis=1) of
    pearson.py:35-38, step by step."""
    f32 = np.float32

    def pw(a):
        n = len(a)
        if n < 8:
            res = f32(0.0)
            for v in a:
                res = f32(res + v)
            return res
        if n <= 128:
            r = [f32(a[j]) for j in range(8)]
            i = 8
            while i < n - (n % 8):
                for j in range(8):
                    r[j] = f32(r[j] + a[i + j])
                i += 8
            res = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
            while i < n:
                res = f32(res + a[i])
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return f32(pw(a[:n2]) + pw(a[n2:]))

    rng = np.random.default_rng(0)
    for n in list(range(1, 300)) + [625, 729, 1000, 1023, 1024, 1025, 2048, 3125, 4096, 4100]:
        a = (rng.standard_normal(n) * rng.choice([1, 1e3, 1e-3]) + rng.choice([0, 5, 100])).astype(f32)
        m = np.stack([a, a[::-1]])
        want = np.add.reduce(m, axis=1)
        assert pw(m[0]).view(np.uint32) == want[0].view(np.uint32) and pw(m[1]).view(np.uint32) == want[1].view(np.uint32), n
    for K in (1, 2, 3, 4, 5, 16, 31, 64, 100, 255, 257, 625, 1024):
        x = (rng.binomial(30, 0.1, size=(5, K)) * f32(0.5) + rng.choice([0, 3])).astype(f32)
        with np.errstate(all="ignore"):
            c = (x.T - np.mean(x, axis=1)).T
            z = (c.T / np.std(c, axis=1)).T
            mine = np.empty_like(x)
            for i, row in enumerate(x):
                mean = f32(pw(row) / f32(K))
                cc = (row - mean).astype(f32)
                m2 = f32(pw(cc) / f32(K))
                d = (cc - m2).astype(f32)
                sd = np.sqrt(f32(pw((d * d).astype(f32)) / f32(K)), dtype=f32)
                mine[i] = (cc / sd).astype(f32)
        assert np.array_equal(np.nan_to_num(z, nan=7).view(np.uint32), np.nan_to_num(mine, nan=7).view(np.uint32)), K


def test_where_the_reference_disagrees_with_itself_the_cell_is_order_sensitive(golden_dir):
    """tests/golden/refspread.json (make_golden_refspread.py: the imported reference under four OpenBLAS core types x
    three thread counts x four operand arrangements): on 18 of its 32 inputs the REFERENCE moves by a bar or more
    between two such runs — "within 1e-5 of the reference" names no single number there.  The fuzzers' rule
    (tests/parity_rule.py) lets a cell leave strict parity only where the input predicate `order_sensitivity >= TAU`
    holds; this pins the predicate to the measurement: every case regenerates from its stored generator state, every
    cell on which the reference's runs are a bar or more apart had sensitivity >= 1.2 (6 x TAU) when the fixture was
    made, and the worst cell of each such case is order-sensitive when recomputed here."""
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden_refspread as mk
    import parity_rule
    with open(os.path.join(golden_dir, "refspread.json")) as fh:
        fx = json.load(fh)
    assert fx["tau"] == parity_rule.TAU and len(fx["cases"]) >= 24
    apart = [c for c in fx["cases"] if c["ref_vs_ref_bars"] >= 1.0]
    assert len(apart) >= 16 and max(c["ref_vs_ref_bars"] for c in apart) > 2.0
    for c in apart:
        assert c["cells_a_bar_or_more_apart"] >= 1 and c["least_order_sensitivity_among_them"] >= 1.2, c["tag"]
        a, b, tag = mk.case_input(c["fuzzer"], c["rng_state"])
        assert tag == c["tag"] and [a.shape[0], b.shape[0], a.shape[1]] == c["shape"]
        with np.errstate(all="ignore"):
            truth = orc.pearson_f64_truth(a, b)
        i, j = c["worst_cell"]
        assert abs(truth[i, j] - c["r_f64_at_worst_cell"]) <= 1e-12
        s = parity_rule.order_sensitivity(a, b, [(i, j)], truth)[0]
        assert s >= parity_rule.TAU and abs(s - c["order_sensitivity_at_worst_cell"]) <= 0.05 * s + 0.01, (c["tag"], s)
        # the reference's distance from float64 depends on the host: by more than a bar between the configurations measured
        flat = [v for vs in c["ref_vs_f64_bars_by_config_and_variant"].values() for v in vs]
        assert max(flat) - min(flat) >= 0.3


def test_the_parity_rule_judges_cells_as_documented():
    """parity_rule.judge: strict cells pass; a cell outside the strict bar passes only if order-sensitive AND within the
    bar of float64; a well-conditioned cell outside the strict bar is a failure whatever the reference's own error."""
    import parity_rule
    rng = np.random.default_rng(3)
    a = rng.standard_normal((6, 512)).astype(np.float32)
    with np.errstate(all="ignore"):
        ref, truth = orc.pearson(a, a).astype(np.float64), orc.pearson_f64_truth(a, a)
    ok = np.ones_like(truth, bool)
    assert parity_rule.judge(ref, ref, truth, ok, a, a)["failures"] == []
    got = ref.copy()
    got[1, 2] += 3e-5  # far outside the bar on a well-conditioned (gaussian) pair
    v = parity_rule.judge(got, ref, truth, ok, a, a)
    assert v["n_strict_fail"] == 1 and len(v["failures"]) == 1 and "NOT order-sensitive" in v["failures"][0][2]
    # an order-sensitive pair: two copies of a row with one dominant column and a constant floor (K = 4 100)
    x = np.full((2, 4100), 0.5, np.float32)
    x[:, 7] = 11.0
    x[1] *= np.float32(0.37)
    with np.errstate(all="ignore"):
        ref, truth = orc.pearson(x, x).astype(np.float64), orc.pearson_f64_truth(x, x)
    ok = np.ones_like(truth, bool)
    s = parity_rule.order_sensitivity(x, x, [(0, 1)], truth)[0]
    assert s >= 1.0, s
    fake_ref = truth + 3.0 * parity_rule.bar_of(truth)              # a reference three bars from float64 ...
    assert parity_rule.judge(truth, fake_ref, truth, ok, x, x)["failures"] == []   # ... a float64-exact device passes
    v = parity_rule.judge(truth + 1.5 * parity_rule.bar_of(truth), fake_ref, truth, ok, x, x)
    assert v["failures"] and all("not within the bar of float64" in f[2] for f in v["failures"])


def test_native_parser_keeps_a_nul_byte_inside_a_header(tmp_path):
    """Found by tools/fuzz_reader_vs_reference.py (round 6): headers travel as one '\\n'-joined buffer, and ctypes' `.value`
    cut it at the first NUL byte — every header behind it was lost.  str.strip() does not remove NUL, so the reference
    keeps it (fasta_reader.py:45,59)."""
    from seekr_amd import _lib
    from seekr_amd.fasta_reader import Reader
    path = str(tmp_path / "nul.fa")
    with open(path, "wb") as fh:
        fh.write(b">a\x00b c\nACGT\n>second\nAC\x00GT\n>third\nTTTT\n")
    fa = _lib.FastaFile(path)
    assert fa.headers() == Reader(path).get_headers() == orc.read_fasta(path)[0] == [">a\x00b c", ">second", ">third"]
    assert list(fa.lengths()) == [len(s) for s in Reader(path).get_seqs()] == [4, 5, 4]
