#!/usr/bin/env python3
"""Golden vectors G11 — center() / standardize() / log2_norm() on hand-assigned count matrices that are NOT float32 — by
RUNNING THE REFERENCE (same conventions as make_golden.py; build container only, no-op without /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -W ignore /root/repo/tests/golden/make_golden_g11.py

kmer_counts.py:165-192 operate on whatever dtype `self.counts` has (test_kmer_counts.py:44-90 assigns the matrix by hand).
For float64 / float16 / integer matrices of 7 x 5 and 3 000 x 256 (inputs: `matrix()` below, seeded — the tests import it)
the fixture holds what the reference leaves behind: the mean / std vectors it computes, the matrices (small shape: whole;
large shape: SHA-256 of the bytes, 64 sampled cells and the float64 sum), the dtype of every result, and — where numpy
refuses the in-place operation (float statistics into an integer matrix, `+= 1` on bool) — the exception's type and text
together with the attribute the reference had already replaced when it was raised.
"""
import hashlib
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = [(7, 5), (3000, 256)]
DTYPES = ["float64", "float16", "int64", "int32", "int16", "int8", "uint8", "uint64", "bool"]


def matrix(dtype, shape, seed=11):
    """Per-kb-like counts: small Poisson numbers times 1000/1995, one all-zero column (std 0 -> NaN after standardize);
    integers: the Poisson numbers themselves (+ a few large / negative ones where the type has room)."""
    rng = np.random.default_rng([seed, shape[0], shape[1]])
    n = rng.poisson(0.9, size=shape).astype(np.int64)
    n[:, shape[1] // 2] = 0
    dt = np.dtype(dtype)
    if dt.kind == "f":
        return (n * (1000.0 / 1995.0)).astype(dt)
    if dt.kind == "b":
        return n > 0
    if dt.kind == "i":
        n[0, 0] = -3
        n[-1, -1] = np.iinfo(dt).max  # `+= 1` wraps here
    if dt == np.uint64:
        n = n.astype(np.uint64)
        n[1, 1] = np.uint64(2 ** 63 + 2 ** 11 + 1)  # not exact in float64
        return n
    return n.astype(dt)


def vectors(cols):
    """User-supplied vectors of several dtypes (broadcast over the rows)."""
    rng = np.random.default_rng(cols)
    base = rng.uniform(0.2, 0.8, cols)
    return {"f32": base.astype(np.float32), "f64": base, "f16": base.astype(np.float16), "i64": rng.integers(1, 4, cols),
            "i16": rng.integers(1, 4, cols).astype(np.int16), "pyfloat": 0.3}


def sample_index(shape):
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    return rng.integers(0, shape[0], 64), rng.integers(0, shape[1], 64)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def record(store, meta, key, value):
    """value: ndarray result of the reference, or an Exception."""
    if isinstance(value, Exception):
        meta[key] = {"exception": type(value).__name__, "message": str(value)}
        return
    a = np.asarray(value)
    m = {"dtype": a.dtype.name, "shape": list(a.shape), "sha256": digest(a)}
    if a.size <= 4096:
        store[key] = a
    else:
        rows, cols = sample_index(a.shape)
        store[key + "__cells"] = a[rows, cols]
        with np.errstate(all="ignore"):
            m["nansum"] = float(np.nansum(a.astype(np.float64)))
            m["n_nan"] = int(np.isnan(a.astype(np.float64)).sum())
    meta[key] = m


def scenarios(counter_cls, dtype, shape):
    """(key, callable -> (counts-or-exception, extra dict of attributes)) for one dtype and shape."""
    import contextlib
    import io
    vecs = vectors(shape[1])

    def fresh(**kw):
        c = counter_cls(silent=True, k=1, **kw)
        c.counts = matrix(dtype, shape)
        return c

    def run(method, **kw):
        c = fresh(**kw)
        out = io.StringIO()
        try:
            with contextlib.redirect_stdout(out), np.errstate(all="ignore"):
                for name in method:
                    getattr(c, name)()
            res = c.counts
        except Exception as e:  # noqa: BLE001
            res = e
        return res, c, out.getvalue()

    yield "center", run(["center"])
    yield "standardize", run(["standardize"])
    yield "center_standardize", run(["center", "standardize"])
    yield "log2_norm", run(["log2_norm"])
    for tag, v in vecs.items():
        yield "center_vec_" + tag, run(["center"], mean=v)
        yield "standardize_vec_" + tag, run(["standardize"], std=v)


def main():
    if not os.path.isdir(os.path.join(REF, "seekr")):
        print("reference not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from seekr.kmer_counts import BasicCounter
    store, meta = {}, {"numpy": np.__version__}
    for dtype in DTYPES:
        for shape in SHAPES:
            for name, (res, c, printed) in scenarios(BasicCounter, dtype, shape):
                key = "%s_%dx%d_%s" % (dtype, shape[0], shape[1], name)
                record(store, meta, key, res)
                meta[key]["warned"] = "WARNING: You have `np.nan` values" in printed
                for attr in ("mean", "std"):
                    val = getattr(c, attr)
                    if isinstance(val, np.ndarray) and name in ("center", "standardize", "center_standardize"):
                        record(store, meta, key + "__" + attr, val)
            print(dtype, shape, "done")
    np.savez_compressed(os.path.join(HERE, "g11_other_dtypes.npz"), **store)
    with open(os.path.join(HERE, "g11_other_dtypes.json"), "w") as fh:
        json.dump(meta, fh, indent=0, sort_keys=True)
    print("wrote g11_other_dtypes.npz (%d arrays) / .json (%d records)" % (len(store), len(meta)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
