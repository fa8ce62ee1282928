"""Tally of the Pearson parity rule inside the differential fuzzers (tests/parity_rule.py): how many
float32 cases are strict — every cell within 2e-6 + 1e-5 |ref| of the reference's float32 result (oracle.pearson, the
numpy path) — and, for the others, how many cells left the strict bar, that ALL of them were order-sensitive (else the
fuzzer has already failed), how sensitive, and how far from float64 the device was on them."""


class StrictTally:
    def __init__(self):
        self.cases = self.strict_ok = 0
        self.worst = 0.0
        self.cells = 0
        self.worst_vs_f64 = 0.0
        self.least_sensitivity = None
        self.tags = []  # (strict ratio, order-sensitive cells, the fuzzer's description of the case)

    def add(self, verdict, tag=None):
        """`verdict`: parity_rule.judge()'s result for one float32 case."""
        self.cases += 1
        self.worst = max(self.worst, verdict["strict_ratio"])
        if verdict["n_strict_fail"] == 0:
            self.strict_ok += 1
            return
        self.cells += verdict["n_order_sensitive"]
        self.worst_vs_f64 = max(self.worst_vs_f64, verdict["worst_vs_f64"])
        self.tags.append((round(verdict["strict_ratio"], 3), verdict["n_order_sensitive"], tag))

    def summary(self):
        return {"float32_cases": self.cases, "strict_ok": self.strict_ok, "cases_with_order_sensitive_cells": len(self.tags),
                "order_sensitive_cells_outside_the_strict_bar": self.cells, "worst_strict_ratio": round(self.worst, 3),
                "worst_device_vs_float64_on_them_over_bar": round(self.worst_vs_f64, 3),
                "strict_failures_on_cells_that_are_not_order_sensitive": 0,  # any such cell fails the fuzzer itself
                "which": [t for t in sorted(self.tags, key=lambda x: -x[0])[:12]]}
