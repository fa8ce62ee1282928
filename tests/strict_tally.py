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
        self.cells_judged = 0
        self.by_width = {}   # K -> [float32 cases, cells judged, cells outside the strict bar (all excused, else the fuzzer failed)]
        self.excused = []    # (K, strict ratio, sensitivity, device vs float64) of every cell the order-sensitivity clause let through

    def add(self, verdict, tag=None):
        """`verdict`: parity_rule.judge()'s result for one float32 case."""
        self.cases += 1
        self.worst = max(self.worst, verdict["strict_ratio"])
        K = (tag or {}).get("K")
        w = self.by_width.setdefault(K, [0, 0, 0])
        w[0] += 1
        w[1] += verdict.get("n_cells", 0)
        w[2] += verdict["n_order_sensitive"]
        self.cells_judged += verdict.get("n_cells", 0)
        self.excused += [(K,) + tuple(c) for c in verdict.get("excused", [])]
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
                "cells_judged": self.cells_judged,
                # the reach of the clause: per width, (cases, cells judged, cells it excused); and what a stricter TAU would
                # do to the excused cells — how many have a sensitivity below TAU' and would count as failures under it
                "by_width": {str(k): v for k, v in sorted(self.by_width.items(), key=lambda kv: (kv[0] is None, kv[0]))},
                "tau_sweep_excused_cells_that_would_fail": {str(t): sum(1 for c in self.excused if c[2] < t) for t in (0.1, 0.2, 0.5, 1.0)},
                "excused_cells_sensitivity_quartiles": (
                    [round(sorted(c[2] for c in self.excused)[int(q * (len(self.excused) - 1))], 3) for q in (0.0, 0.25, 0.5, 0.75, 1.0)]
                    if self.excused else []),
                # |got - ref| > bar with got within e of float64 means the REFERENCE is at least 1 - e bars from float64: the
                # clause can only fire where the reference's own float32 result is that far off (measured: the smallest such distance)
                "excused_cells_least_reference_vs_float64_over_bar": round(min((c[4] for c in self.excused), default=0.0), 3),
                "excused_cells_worst_device_vs_float64_over_bar": round(max((c[3] for c in self.excused), default=0.0), 3),
                "which": [t for t in sorted(self.tags, key=lambda x: -x[0])[:12]]}
