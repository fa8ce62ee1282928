"""STRICT Pearson parity (VERDICT r2 #1b): |device - oracle.pearson| <= 2e-6 + 1e-5 |ref| cell by cell — the reference's
float32 numpy path itself as the yardstick, no allowance for its own distance from float64 — on the grid of
tools/strict_parity.py: sparse raw counts / binomial raw counts / 0-1 rows / gaussian x {raw values row-standardised by
pearson(), Log2.post-normalised} x K in {256, 4 096, 16 384}, default precision (f16x3) and the fp32 kernel."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_strict_parity_grid(tmp_path):
    out_json = str(tmp_path / "strict.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "strict_parity.py"), "--rows", "768", "--json", out_json],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "-> ok" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    grid = json.load(open(out_json))["grid"]
    assert len(grid) >= 20
    for cell in grid:
        assert cell["f16x3"]["strict"] <= 1.0 and cell["fp32"]["strict"] <= 1.0, cell
        # the device is never further from float64 than the bar allows on its own, whatever the reference does
        assert cell["f16x3"]["vs_f64"] <= 1.0, cell
