"""The native writers against numpy's own files, byte for byte.  The host variants need no GPU,
so the formatting logic (exact "%1.6f" rounding with integer arithmetic, .npy header padding,
threaded row order) is pinned on CPU."""
import filecmp
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from seekr_amd import _lib
    return _lib


def hard_values(rng, n, dtype):
    """Values that stress '%1.6f': exact ties at the 7th decimal, tiny negatives, big and small
    magnitudes, subnormals, per-kb count values, NaN and infinities."""
    parts = [
        rng.standard_normal(n) * 3,
        rng.integers(0, 40, n) * (1000.0 / rng.integers(1, 2000, n)),          # per-kb counts
        (rng.integers(-2_000_000, 2_000_000, n) + 0.5) / 1e6,                   # decimal ties (inexact in binary)
        rng.integers(-2 ** 20, 2 ** 20, n) / 2.0 ** 7,                          # binary fractions: exact ties for %f
        np.ldexp(rng.integers(1, 2 ** 23, n).astype(np.float64), rng.integers(-160, 40, n)),
        np.array([0.0, -0.0, 1e-7, -1e-7, 4.9999995e-7, 5e-7, -5e-7, 0.9999995, 0.99999949, 123456.7890125,
                  1e15, -1e15, 3.4e38, -3.4e38, 1e-45, np.nan, -np.nan, np.inf, -np.inf, 2.5e-7, 0.0000005, 1.0000005]),
    ]
    v = np.concatenate(parts)
    with np.errstate(over="ignore"):
        return v.astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("fmt", ["%1.6f", "%.18e"])
def test_csv_bytes_equal_numpy_savetxt(dtype, fmt, L, tmp_path):
    rng = np.random.default_rng(7)
    v = hard_values(rng, 3000, dtype)
    cols = 37
    a = np.resize(v, (len(v) // cols, cols)).astype(dtype)
    want, got = str(tmp_path / "want.csv"), str(tmp_path / "got.csv")
    np.savetxt(want, a, delimiter=",", fmt=fmt)
    for threads in (1, 5):
        L.save_csv(got, a, L.FMT_FIXED6 if fmt == "%1.6f" else L.FMT_SCI18, threads=threads)
        assert filecmp.cmp(want, got, shallow=False), (dtype, fmt, threads)


def test_csv_large_random_float32(L, tmp_path):
    rng = np.random.default_rng(1)
    a = (rng.binomial(30, 0.1, size=(700, 1024)) * np.float32(1000 / 1995)).astype(np.float32)
    a = np.log2(a + 1).astype(np.float32) - np.float32(0.731)
    want, got = str(tmp_path / "want.csv"), str(tmp_path / "got.csv")
    np.savetxt(want, a, delimiter=",", fmt="%1.6f")
    L.save_csv(got, a)
    assert filecmp.cmp(want, got, shallow=False)


@pytest.mark.parametrize("shape,dtype", [((5, 16), np.float32), ((1, 4096), np.float32), ((4096,), np.float32),
                                         ((3, 3), np.float64), ((7, 1), np.uint32), ((0, 16), np.float32),
                                         ((123, 4567), np.float32), ((12345678,), np.float32)])
def test_npy_bytes_equal_numpy_save(shape, dtype, L, tmp_path):
    rng = np.random.default_rng(3)
    a = (rng.standard_normal(shape) * 100).astype(dtype)
    want, got = str(tmp_path / "want.npy"), str(tmp_path / "got")  # np.save appends .npy
    np.save(want, a)
    L.save_npy(got, a)
    assert os.path.exists(got + ".npy") and not os.path.exists(got)
    assert filecmp.cmp(want, got + ".npy", shallow=False)
    back = np.load(got + ".npy")
    assert back.dtype == a.dtype and back.shape == a.shape


def test_writer_errors(L, tmp_path):
    with pytest.raises(OSError):
        L.save_csv(str(tmp_path / "no" / "such" / "dir.csv"), np.zeros((2, 2), np.float32))


@pytest.mark.gpu
def test_device_matrices_stream_to_identical_files(L, tmp_path):
    """skr_mat_save_npy / skr_mat_save_csv stream a device matrix through pinned double buffers;
    sizes chosen to cross several chunk boundaries."""
    ctx = L.default_context()
    rng = np.random.default_rng(11)
    a = (rng.standard_normal((9001, 4096)) * 3).astype(np.float32)
    dev = ctx.from_numpy(a)
    want, got = str(tmp_path / "want.npy"), str(tmp_path / "got.npy")
    np.save(want, a)
    L.save_npy(got, dev)
    assert filecmp.cmp(want, got, shallow=False)
    small = a[:2500]
    np.savetxt(str(tmp_path / "want.csv"), small, delimiter=",", fmt="%1.6f")
    L.save_csv(str(tmp_path / "got.csv"), dev.view(0, 2500))
    assert filecmp.cmp(str(tmp_path / "want.csv"), str(tmp_path / "got.csv"), shallow=False)
    d64 = ctx.from_numpy(a[:50].astype(np.float64))
    np.savetxt(str(tmp_path / "want64.csv"), a[:50].astype(np.float64), delimiter=",")
    L.save_csv(str(tmp_path / "got64.csv"), d64, L.FMT_SCI18)
    assert filecmp.cmp(str(tmp_path / "want64.csv"), str(tmp_path / "got64.csv"), shallow=False)


def repr_values(rng, n, dtype):
    """Every regime of numpy's float str(): powers of two (asymmetric rounding interval), powers of
    ten, the 1e-4 / 1e16 switch points, subnormals, short and long digit strings."""
    info = np.finfo(dtype)
    parts = [
        hard_values(rng, n, dtype).astype(np.float64),
        np.ldexp(1.0, np.arange(info.minexp - info.nmant, info.maxexp)),                    # all powers of two
        np.ldexp(1.0, np.arange(info.minexp - info.nmant, info.maxexp)) * (1 + info.eps),   # and their successors
        10.0 ** np.arange(-45 if dtype == np.float32 else -320, 39 if dtype == np.float32 else 308),
        np.array([1e-4, 9.9999e-5, 1.00001e-4, 1e16, 9.999999e15, 1.0000001e16, 0.1, 0.3, 2.5, 1 / 3, 1e15, 123456789.0]),
        rng.integers(1, 10 ** 6, n) / 10.0 ** rng.integers(0, 8, n),                        # short decimals
        np.exp(rng.uniform(-100, 80, n)),
    ]
    with np.errstate(over="ignore", under="ignore"):
        v = np.concatenate(parts).astype(dtype)
    return np.concatenate([v, -v])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_repr_cells_equal_numpy_str(dtype, L, tmp_path):
    """fmt_mode 2 == str(numpy scalar) for every value, NaN -> empty (pandas na_rep)."""
    rng = np.random.default_rng(17)
    v = repr_values(rng, 20000, dtype)
    if dtype == np.float32:  # plus raw bit patterns
        raw = rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32)
        v = np.concatenate([v, raw])
    a = np.ascontiguousarray(v.reshape(1, -1))
    got = str(tmp_path / "got.csv")
    L.save_csv(got, a, L.FMT_REPR, threads=3)
    cells = open(got).read().rstrip("\n").split(",")
    want = ["" if np.isnan(x) else str(x) for x in v]
    assert len(cells) == len(want)
    bad = [(i, c, w) for i, (c, w) in enumerate(zip(cells, want)) if c != w]
    assert not bad, bad[:10]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_labelled_csv_bytes_equal_pandas(dtype, L, tmp_path):
    import pandas as pd
    rng = np.random.default_rng(5)
    a = (rng.binomial(30, 0.1, size=(57, 64)) * np.float32(1000 / 1995)).astype(np.float32)
    a = (np.log2(a + 1) - np.float32(0.731)).astype(dtype)
    a[3, 4] = np.nan
    a[5, 6] = np.inf
    a[7, 8] = 1e-7
    index = [">ENST%05d|gene,%d|\"q\" x" % (i, i) if i % 7 == 0 else ">ENST%05d.1|GENE%d" % (i, i) for i in range(57)]
    columns = ["".join(p) for p in __import__("itertools").product("AGTC", repeat=3)]
    want, got = str(tmp_path / "want.csv"), str(tmp_path / "got.csv")
    pd.DataFrame(data=a, index=index, columns=columns).to_csv(want)
    L.save_csv_labelled(got, a, index, columns, threads=4)
    assert filecmp.cmp(want, got, shallow=False)
    # RangeIndex-like labels (seekr_pearson with binary input writes names None -> 0..n-1)
    pd.DataFrame(a[:, :57], None, None).to_csv(want)
    L.save_csv_labelled(got, np.ascontiguousarray(a[:, :57]), range(57), range(57))
    assert filecmp.cmp(want, got, shallow=False)


def _frames_equal(native, df):
    values, rows, cols = native
    want = df.values.astype(np.float64)
    assert values.shape == want.shape and values.dtype == np.float64
    assert np.array_equal(values.view(np.uint64)[~np.isnan(want)], want.view(np.uint64)[~np.isnan(want)])  # bit for bit
    assert np.array_equal(np.isnan(values), np.isnan(want))
    assert rows == [str(x) for x in df.index] and cols == [str(x) for x in df.columns]


def test_csv_reader_equals_pandas_on_count_files(L, tmp_path):
    """pd.read_csv(path, index_col=0) on the files seekr_kmer_counts writes (labelled, float32 shortest
    repr) and on %1.6f-style cells: values bit for bit, labels as text."""
    import itertools
    import pandas as pd
    rng = np.random.default_rng(9)
    a = (rng.binomial(30, 0.1, size=(211, 256)) * np.float32(1000 / 1995)).astype(np.float32)
    a = (np.log2(a + 1) - np.float32(0.731)).astype(np.float32)
    a[3, 4] = np.nan
    a[5, 6] = np.inf
    a[6, 6] = -np.inf
    a[7, 8] = 1.5e-5
    a[9, 9] = -0.0
    index = [">ENST%05d|gene,%d|\"q\" x" % (i, i) if i % 7 == 0 else ">ENST%05d.1|GENE%d" % (i, i) for i in range(211)]
    columns = ["".join(p) for p in itertools.product("AGTC", repeat=4)]
    path = str(tmp_path / "counts.csv")
    pd.DataFrame(a, index, columns).to_csv(path)
    for threads in (1, 6):
        native = L.load_csv_labelled(path, threads=threads)
        assert native is not None
        _frames_equal(native, pd.read_csv(path, index_col=0))
    # CRLF line ends, no trailing newline, blank lines, integer-looking cells
    text = open(path).read().replace("\n", "\r\n").rstrip("\r\n") + "\r\n\r\n"
    text = text.replace(",0.0,", ",0,", 5)
    crlf = str(tmp_path / "crlf.csv")
    open(crlf, "w", newline="").write(text)
    _frames_equal(L.load_csv_labelled(crlf), pd.read_csv(crlf, index_col=0))


def test_csv_reader_declines_what_pandas_may_round_differently(L, tmp_path):
    """Fields outside the exactly reproducible subset are not parsed natively (the caller falls back to
    pandas), and inside it the reader equals pandas even where pandas is not correctly rounded elsewhere."""
    import pandas as pd
    head = ",c0,c1\n"
    cases = {"17 digits": ">a,0.12345678901234567,1\n", "big exponent": ">a,1.5e-30,1\n", "text": ">a,hello,1\n",
             "ragged": ">a,1\n", "18 digit chars": ">a,0.000000000000123456,1\n", "numeric index": "7,1.0,2.0\n"}
    for name, row in cases.items():
        p = str(tmp_path / "x.csv")
        open(p, "w").write(head + row)
        assert L.load_csv_labelled(p) is None, name
    rng = np.random.default_rng(2)
    cells = ["%.*e" % (int(rng.integers(0, 14)), x) for x in np.exp(rng.uniform(-60, 60, 20000))] + \
            ["%.*f" % (int(rng.integers(0, 9)), x) for x in rng.standard_normal(20000) * 1000]
    p = str(tmp_path / "mixed.csv")
    with open(p, "w") as fh:
        fh.write(",v\n")
        for i, c in enumerate(cells):
            fh.write(">r%d,%s\n" % (i, c))
    native = L.load_csv_labelled(p)
    if native is not None:  # every cell happened to be in the subset
        _frames_equal(native, pd.read_csv(p, index_col=0))
    ok = bad = 0
    for c in cells[::40]:
        q = _one(tmp_path, c)
        nat = L.load_csv_labelled(q)
        if nat is None:
            bad += 1
            continue
        ok += 1
        assert nat[0][0, 0].tobytes() == np.float64(pd.read_csv(q, index_col=0).values[0, 0]).tobytes(), c
    assert ok > 300 and bad > 20


def _one(tmp_path, cell):
    q = str(tmp_path / "one.csv")
    with open(q, "w") as fh:
        fh.write(",v\n>r,%s\n" % cell)
    return q
