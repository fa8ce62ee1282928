"""Tally of STRICT Pearson parity inside the differential fuzzers: per case, the largest
|got - reference| / (2e-6 + 1e-5 |reference|) with reference = oracle.pearson (the numpy float32 path), i.e. the
north_star bar with NO allowance for the reference's own error.  The fuzzers' pass/fail criterion is unchanged (it
allows a multiple of the reference's error on inputs whose float32 standardisation is ill-conditioned); this records
how many cases need that allowance at all, and how far from float64 the reference itself is in those."""
import numpy as np


class StrictTally:
    def __init__(self):
        self.cases = self.strict_ok = 0
        self.worst = 0.0
        self.not_strict = []  # (strict ratio, reference's own |ref - f64| / bar on that case, device's)
        self.tags = []        # (strict ratio, the fuzzer's description of the case) for the cases above

    def add(self, got, ref, truth, ok, tag=None):
        """float32 cases only; `ok`: cells where reference and truth are finite numbers worth judging."""
        if not ok.any():
            return
        got, ref, truth = (np.asarray(a, dtype=np.float64) for a in (got, ref, truth))
        bar = 2e-6 + 1e-5 * np.abs(np.where(ok, ref, 0.0))
        strict = float(np.max(np.where(ok, np.abs(got - ref), 0.0) / bar))
        self.cases += 1
        self.worst = max(self.worst, strict)
        if strict <= 1.0:
            self.strict_ok += 1
        else:
            e_ref = float(np.max(np.where(ok, np.abs(ref - truth), 0.0) / bar))
            e_dev = float(np.max(np.where(ok, np.abs(got - truth), 0.0) / bar))
            self.not_strict.append((round(strict, 3), round(e_ref, 3), round(e_dev, 3)))
            self.tags.append((round(strict, 3), tag))

    def summary(self):
        ns = sorted(self.not_strict, reverse=True)
        closer = sum(1 for s, e_ref, e_dev in ns if e_dev <= e_ref)
        return {"float32_cases": self.cases, "strict_ok": self.strict_ok, "needed_the_allowance": len(ns),
                "worst_strict_ratio": round(self.worst, 3),
                "of_those_device_closer_to_float64_than_reference": closer,
                "of_those_min_reference_error_over_bar": min((e for _, e, _ in ns), default=None),
                "top5 (strict, ref vs f64, device vs f64)": ns[:5],
                "which": [t for _, t in sorted(self.tags, key=lambda x: -x[0])[:12]]}
