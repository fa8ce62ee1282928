"""The N x N Pearson matrix at the sizes BASELINE.json names, on one MI355X, checked against the oracle
(tools/fullsize_check.py): config 2 (50 000 rows, the size bench.py times) and config 4's 200 000 rows
(r = 160 GB of the 288 GB; 782 x 782 tiles in the persistent queue; byte offsets past 2^37)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rows", [50_000, 200_000])
def test_full_size_self_pearson_against_oracle(rows):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fullsize_check.py"), "--rows", str(rows)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "fullsize ok rows=%d" % rows in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_bench_line_carries_the_measurement_contract():
    """bench.py at N = 1 on a reduced workload: one JSON line with the contract's fields — roofline (MFMA bound, frac =
    achieved / peak), roofline_count (HBM bound), cpu_baseline (kind "port", repeats stated), the post-timing correctness
    probe and the PCIe-inclusive e2e block — and a verified = true r."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "6000", "--length", "600", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "roofline_count", "cpu_baseline", "verified", "e2e"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert d["roofline_count"]["bound"] == "hbm" and 0 < d["roofline_count"]["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and "median of 3" in cb["sample"] and cb["value"] > 0 and cb["cores"] >= 1
    assert d["verified"] is True and d["verified_detail"]["worst_error_over_bar"] <= 1.0
    assert d["e2e"]["fasta_to_host_counts_mbases_per_s"] > 0 and d["e2e"]["host_to_host_pearson_mpairs_per_s"] > 0
    # round 4: the opt-in two-product-unit contraction measured after the timed region, verified like the headline
    arm = d["f16f8_arm"]
    assert arm["operand_kind"] == 3 and arm["verified"] is True and arm["worst_error_over_bar"] <= 1.0 and arm["value"] > 0
    assert "never part of `value`" in arm["note"] and d["config"]["precision"] == "f16x3"


def test_bench_line_prices_an_output_bound_contraction_against_hbm():
    """At k = 4 (256 columns) writing r (4 B per pair) takes longer at 8 TB/s than 512 flop per pair take at the dense
    matrix-core peak: `roofline.bound` says "hbm", achieved / peak are GB/s, the matrix-core view is kept next to it."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--k", "4", "--rows", "6000", "--length", "600", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-target-200k", "--no-f16f8-arm"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["mfma_view"]["peak_tflops"] == 2500.0 and 0 < rf["mfma_view"]["frac"] < rf["frac"] and d["verified"] is True
