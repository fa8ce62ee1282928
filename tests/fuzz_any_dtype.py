"""Differential fuzzing of BasicCounter.center / standardize / log2_norm on hand-assigned matrices of random dtype, shape,
layout (contiguous, strided view, transposed) and user vectors against what kmer_counts.py:165-192 do — three numpy
statements each, restated here verbatim — on the same input: bytes of the result, of the replaced mean / std, the dtype of
everything, whether the caller's own array was written, and the exception's type and text where numpy refuses."""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekr_amd.kmer_counts import BasicCounter  # noqa: E402

DTYPES = ["float64", "float16", "float32", "int64", "int32", "int16", "int8", "uint8", "uint16", "uint32", "uint64", "bool"]


def reference_methods(counts, mean, std, method):
    """kmer_counts.py:165-192 on (counts, mean, std); returns (counts, mean, std)."""
    if method == "center":
        if mean is True:
            mean = np.mean(counts, axis=0)
        counts -= mean
    elif method == "standardize":
        if std is True:
            std = np.std(counts, axis=0)
        counts /= std
    else:
        counts += 1
        counts = np.log2(counts)
    return counts, mean, std


def random_vector(rng, cols):
    kind = rng.integers(0, 8)
    base = rng.uniform(0.2, 3.0, cols)
    if kind == 0:
        return True
    if kind == 1:
        return float(rng.uniform(0.3, 2.0))
    if kind == 2:
        return int(rng.integers(1, 4))
    dt = ["float32", "float64", "float16", "int64", "int16", "uint8"][int(rng.integers(0, 6))]
    vec = (base * (3 if "int" in dt else 1)).astype(dt) if "int" not in dt else rng.integers(1, 5, cols).astype(dt)
    shape = rng.integers(0, 12)  # numpy's broadcasting: a row vector as a matrix is fine, other shapes get numpy's sentence
    if shape == 0:
        return vec.reshape(1, cols)
    if shape == 1:
        return np.concatenate([vec, vec[:1]])      # one value too many
    if shape == 2:
        return vec.reshape(1, 1, cols)              # broadcasts to 3-D: does not fit the output
    if shape == 3:
        return vec[:1]                              # one value for every column
    return vec


def fuzz(seed, budget_s=30.0, max_cases=10 ** 9):
    rng = np.random.default_rng(seed)
    t0, n = time.time(), 0
    while time.time() - t0 < budget_s and n < max_cases:
        dtype = np.dtype(DTYPES[int(rng.integers(0, len(DTYPES)))])
        rows, cols = int(rng.choice([1, 2, 3, 7, 64, 300, 1025])), int(rng.choice([1, 3, 5, 16, 33, 256, 300]))
        if rng.integers(0, 60) == 0:  # past numpy's 8 192-element buffer piece (column-major layouts add in such pieces)
            rows, cols = int(rng.choice([8193, 9000, 16400])), int(rng.choice([1, 2, 5]))
        raw = rng.poisson(1.1, size=(rows * 2, cols * 2))
        if rng.integers(0, 4) == 0:
            raw[:, rng.integers(0, cols * 2)] = 0
        base = (raw * (0.5 if dtype.kind == "f" else 1)).astype(dtype) if dtype.kind != "b" else raw > 0
        layout = int(rng.integers(0, 4))

        def view(a):
            if layout == 0:
                return np.ascontiguousarray(a[:rows, :cols])
            if layout == 1:
                return a[:rows, :cols]               # rows with a gap behind them
            if layout == 2:
                return a[::2, ::2]                   # strided both ways
            return np.ascontiguousarray(a[:cols, :rows]).T  # transposed: column-major
        method = ["center", "standardize", "log2_norm"][int(rng.integers(0, 3))]
        vec = random_vector(rng, view(base).shape[1])
        mine_base, ref_base = base.copy(), base.copy()
        target, ref_target = view(mine_base), view(ref_base)
        c = BasicCounter(k=1, silent=True, mean=vec if method == "center" else True, std=vec if method == "standardize" else True)
        c.counts = target
        out = io.StringIO()
        try:
            with contextlib.redirect_stdout(out), np.errstate(all="ignore"):
                getattr(c, method)()
            got_exc = None
        except Exception as e:  # noqa: BLE001
            got_exc = e
        try:
            with np.errstate(all="ignore"):
                want, wmean, wstd = reference_methods(ref_target, vec if method == "center" else True, vec if method == "standardize" else True, method)
            want_exc = None
        except Exception as e:  # noqa: BLE001
            want_exc = e
        tag = (seed, n, dtype.name, (rows, cols), layout, method, type(vec).__name__ if not isinstance(vec, np.ndarray) else vec.dtype.name)
        # numpy adds the ROWS one after the other when axis 1 is the faster one (any C-like layout: what get_counts() makes);
        # a column-major matrix — or a single column — it reduces column by column in its PAIRWISE order, in 8 192-element
        # pieces: both reproduced on the device for every dtype (skr_host_colstat / skr_host_colstat_colmajor), bit for bit.

        def same(a, b):
            return np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes()
        assert type(got_exc) is type(want_exc) and str(got_exc) == str(want_exc), (tag, repr(got_exc), repr(want_exc))
        # the attribute the reference replaces BEFORE the in-place operation (also when that then raises)
        for attr, w in (("mean", wmean if want_exc is None else None), ("std", wstd if want_exc is None else None)):
            if w is not None and isinstance(w, np.ndarray) and not isinstance(vec, np.ndarray):
                g = getattr(c, attr)
                if isinstance(g, np.ndarray):
                    assert g.dtype == w.dtype and same(g, w), (tag, attr)
        if want_exc is None:
            got = np.asarray(c.counts)
            assert got.dtype == want.dtype and got.shape == want.shape, (tag, got.dtype, want.dtype)
            if method == "log2_norm":
                tol = 2e-3 if got.dtype == np.float16 else 1e-6
                with np.errstate(all="ignore"):
                    ok = (got == want) | (np.isnan(got.astype(np.float64)) & np.isnan(want.astype(np.float64))) | \
                         (np.abs(got.astype(np.float64) - want.astype(np.float64)) <= tol * (1 + np.abs(want.astype(np.float64))))
                assert ok.all(), (tag, got[~ok][:3], want[~ok][:3])
                assert c.counts is not target
            else:
                assert c.counts is target, tag  # in place, like `counts -= mean`
                assert same(got, want), (tag, got.ravel()[:4], want.ravel()[:4])
            # what the operation left in the caller's own memory (the `+= 1` of log2_norm included), gaps untouched
            assert same(mine_base, ref_base), tag
            assert ("WARNING: You have `np.nan` values" in out.getvalue()) == (method == "standardize" and bool(np.isnan(want.astype(np.float64)).any())), tag
        n += 1
    return n


if __name__ == "__main__":
    k = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
    print("any-dtype fuzz ok: %d cases" % k)
