"""Soak of the structured-row class of tests/fuzz_pearson.py alone: every second case a
structured one, at EVERY width the split-fp16 contraction serves (64 ... 65 536 columns), the second operand of a cross
comparison drawn from another structure class than the first.  The rule is the fuzzers' (tests/parity_rule.py): strict
against the reference's float32 result, float64 as the yardstick only on order-sensitive cells.

    gpurun --timeout 2400 -- 'python tools/soak_structured.py 1800 > gpurun_out/soak.log'      # seconds; default precision
"""
import os
import sys
import time

os.environ.setdefault("SEEKR_FUZZ_LAYOUT_EVERY", "2")
os.environ.setdefault("SEEKR_FUZZ_LAYOUT_WIDTHS", "64,256,1024,4096,16384,65536")
os.environ.setdefault("SEEKR_FUZZ_LAYOUT_CROSS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import fuzz_pearson  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 100000
t0 = time.time()
n = fuzz_pearson.fuzz(seed, budget_s=budget)
print("precision {}, structured rows every {}nd case at widths {} (cross operands of different structure: {}), seed {}: "
      "{} cases in {:.0f} s, no failure".format(os.environ.get("SEEKR_PRECISION", "f16x3 (default)"), fuzz_pearson.LAYOUT_EVERY,
                                                 fuzz_pearson.LAYOUT_WIDTHS, fuzz_pearson.LAYOUT_CROSS_STRUCTURE, seed, n,
                                                 time.time() - t0))
print(fuzz_pearson.TALLY.summary())
