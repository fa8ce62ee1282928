"""The fuzzers' parity rule (tests/parity_rule.py) exercised WITHOUT a GPU: both fuzzers run their own case generators and
judging code against an IDEAL device — rows standardised in float32 in numpy's own order (as the fill kernels do), the
inner product exact (float64), rounded once to float32.  Such a device is at most a rounding from float64, so every cell on
which it leaves the strict bar does so because the REFERENCE is off — and the rule must find every one of those cells
order-sensitive (an input predicate), or the predicate has a hole.  Same code path as the GPU fuzz tests, seconds instead
of a soak."""
import types

import numpy as np

from oracle import seekr_oracle as orc
import parity_rule


def ideal_pearson(a, b, row_standardize=True, outfile=None):
    a, b = np.asarray(a), np.asarray(b)
    f32 = a.dtype == np.float32 and b.dtype == np.float32
    with np.errstate(all="ignore"):
        if f32:
            za = parity_rule.f32_rows(a, row_standardize).astype(np.float64)
            zb = za if b is a else parity_rule.f32_rows(b, row_standardize).astype(np.float64)
            return (np.inner(za, zb) / a.shape[1]).astype(np.float32)
        return orc.pearson_f64_truth(a, b, row_standardize)


def test_pearson_fuzzer_under_an_ideal_device(monkeypatch):
    import fuzz_pearson
    monkeypatch.setattr(fuzz_pearson, "pearson", ideal_pearson)
    monkeypatch.setattr(fuzz_pearson, "TALLY", fuzz_pearson.StrictTally())
    n = fuzz_pearson.fuzz(2024, budget_s=20.0, max_cases=1500)
    t = fuzz_pearson.TALLY.summary()
    assert n >= 300 and t["float32_cases"] >= 100
    assert t["strict_failures_on_cells_that_are_not_order_sensitive"] == 0
    assert t["worst_device_vs_float64_on_them_over_bar"] <= 0.2  # the ideal device: float32 standardisation is all it loses


def test_pipeline_fuzzer_under_an_ideal_device(monkeypatch):
    import fuzz_differential

    def run(seqs, **kw):
        raw = orc.raw_counts(list(seqs), kw["k"], alphabet=kw["alphabet"])
        with np.errstate(all="ignore"):
            c, m, s = orc.normalize(raw, mean=kw["mean"], std=kw["std"], log2=kw["log2"])
        return types.SimpleNamespace(counts=np.array(c, np.float32), mean=m, std=s)

    monkeypatch.setattr(fuzz_differential, "pearson", ideal_pearson)
    monkeypatch.setattr(fuzz_differential, "run", run)
    monkeypatch.setattr(fuzz_differential, "TALLY", fuzz_differential.StrictTally())
    n = fuzz_differential.fuzz(2025, budget_s=20.0, max_cases=400)
    t = fuzz_differential.TALLY.summary()
    assert n >= 60 and t["strict_failures_on_cells_that_are_not_order_sensitive"] == 0
