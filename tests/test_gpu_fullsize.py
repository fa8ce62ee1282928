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
