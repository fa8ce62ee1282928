"""Golden set G11 (tests/golden/make_golden_g11.py): center() / standardize() / log2_norm() on hand-assigned count matrices
of every dtype but float32, as the reference leaves them.  One comparison for two runners: the oracle's restatement
(tests/test_oracle_golden.py, CPU) and seekr_amd.BasicCounter (tests/test_gpu_parity.py, on the GPU)."""
import contextlib
import hashlib
import io
import json
import os

import numpy as np

import make_golden_g11 as mk

METHODS = {"center": ["center"], "standardize": ["standardize"], "center_standardize": ["center", "standardize"],
           "log2_norm": ["log2_norm"]}


def load(golden_dir):
    with open(os.path.join(golden_dir, "g11_other_dtypes.json")) as fh:
        meta = json.load(fh)
    return meta, np.load(os.path.join(golden_dir, "g11_other_dtypes.npz"))


def scenario(name, cols):
    """-> (methods to call, constructor keywords)."""
    if name in METHODS:
        return METHODS[name], {}
    method, _, tag = name.partition("_vec_")
    return [method], {("mean" if method == "center" else "std"): mk.vectors(cols)[tag]}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def compare(meta, store, key, got, exact=True):
    want = meta[key]
    if "exception" in want:
        assert isinstance(got, Exception), (key, "the reference raises", want["exception"], "got", type(got))
        assert type(got).__name__ == want["exception"] and str(got) == want["message"], (key, repr(got), want)
        return
    assert not isinstance(got, Exception), (key, repr(got))
    got = np.asarray(got)
    assert got.dtype.name == want["dtype"] and list(got.shape) == want["shape"], (key, got.dtype, got.shape, want["dtype"], want["shape"])
    if digest(got) == want["sha256"]:
        return
    if exact:
        full = store[key] if key in store.files else None
        where = ""
        if full is not None:
            bad = np.argwhere(~((got == full) | (np.isnan(got.astype(np.float64)) & np.isnan(full.astype(np.float64)))))
            where = " first at {}: {!r} vs {!r}".format(bad[0], got[tuple(bad[0])], full[tuple(bad[0])]) if len(bad) else " (NaN payload / -0.0 only)"
        raise AssertionError("{}: bytes differ from the reference's{}".format(key, where))
    # log2 outputs: |a - b| <= atol + 1e-5 |b| (north_star's bar for normalised counts), one unit of half for float16
    rtol, atol = (2e-3, 1e-3) if got.dtype == np.float16 else (1e-5, 1e-6)
    if key in store.files:
        ref, mine = store[key].astype(np.float64), got.astype(np.float64)
    else:
        rows, cols = mk.sample_index(got.shape)
        ref, mine = store[key + "__cells"].astype(np.float64), got[rows, cols].astype(np.float64)
        with np.errstate(all="ignore"):
            assert int(np.isnan(got.astype(np.float64)).sum()) == want["n_nan"], key
            assert abs(float(np.nansum(got.astype(np.float64))) - want["nansum"]) <= 1e-6 * abs(want["nansum"]) + 1e-6 or got.dtype == np.float16, key
    with np.errstate(all="ignore"):
        same = (mine == ref) | (np.isnan(mine) & np.isnan(ref)) | (np.abs(mine - ref) <= atol + rtol * np.abs(ref))
    assert same.all(), (key, mine[~same][:4], ref[~same][:4])


def check_all(golden_dir, run, dtypes=None, shapes=None):
    """run(dtype, shape, methods, kwargs) -> (counts-or-exception, mean attribute, std attribute, printed text)."""
    meta, store = load(golden_dir)
    n = 0
    for dtype in dtypes or mk.DTYPES:
        for shape in shapes or mk.SHAPES:
            names = list(METHODS) + ["%s_vec_%s" % (m, t) for t in mk.vectors(shape[1]) for m in ("center", "standardize")]
            for name in names:
                key = "%s_%dx%d_%s" % (dtype, shape[0], shape[1], name)
                methods, kw = scenario(name, shape[1])
                got, mean, std, printed = run(dtype, shape, methods, kw)
                compare(meta, store, key, got, exact="log2" not in name)
                assert ("WARNING: You have `np.nan` values" in printed) == meta[key]["warned"], key
                if name in METHODS:
                    for attr, val in (("mean", mean), ("std", std)):
                        if key + "__" + attr in meta:
                            assert isinstance(val, np.ndarray), (key, attr, val)
                            compare(meta, store, key + "__" + attr, val)
                        else:
                            assert not isinstance(val, np.ndarray), (key, attr)
                n += 1
    return n


def run_counter(counter_cls):
    """The runner over a BasicCounter class (the product's; the generator runs the reference's the same way)."""
    def run(dtype, shape, methods, kw):
        c = counter_cls(silent=True, k=1, **kw)
        c.counts = mk.matrix(dtype, shape)
        out = io.StringIO()
        try:
            with contextlib.redirect_stdout(out), np.errstate(all="ignore"):
                for name in methods:
                    getattr(c, name)()
            res = c.counts
        except Exception as e:  # noqa: BLE001 - compared with the reference's
            res = e
        return res, c.mean, c.std, out.getvalue()
    return run


def run_oracle(orc):
    """The runner over the oracle's restatement (oracle/seekr_oracle.py: host_center / host_standardize / host_log2_norm)."""
    nan_warning = "WARNING: You have `np.nan` values"

    def run(dtype, shape, methods, kw):
        counts = mk.matrix(dtype, shape)
        state = {"mean": kw.get("mean", True), "std": kw.get("std", True), "printed": ""}
        try:
            with np.errstate(all="ignore"):
                for name in methods:
                    if name == "center":
                        state["mean"], op = orc.host_center(counts, state["mean"])
                        op()
                    elif name == "standardize":
                        state["std"], op = orc.host_standardize(counts, state["std"])
                        op()
                        if np.isnan(counts).any():
                            state["printed"] = nan_warning
                    else:
                        counts = orc.host_log2_norm(counts)
            res = counts
        except Exception as e:  # noqa: BLE001
            res = e
        return res, state["mean"], state["std"], state["printed"]
    return run
