"""Long differential soak on the GPU box: every fuzzer of tests/ with fresh seeds for a stated number of seconds each.
    gpurun --timeout 2700 -- 'python tools/soak.py 900 600 300 240'
(differential BasicCounter/pearson pipeline, pearson API, consumers, FASTA reader).  Round 2, after the fill-kernel
rewrite and the small-k routing change: 72 968 + 268 583 + 49 210 + 166 300 cases, no failure."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzz_consumers, fuzz_differential, fuzz_fasta, fuzz_pearson  # noqa: E402

budgets = [float(a) for a in sys.argv[1:5]] + [60.0] * 4
seed = int(time.time()) % 100000
out = {"seed": seed}
out["differential"] = fuzz_differential.fuzz(seed + 11, budget_s=budgets[0])
out["pearson"] = fuzz_pearson.fuzz(seed + 12, budget_s=budgets[1])
out["consumers"] = fuzz_consumers.fuzz(seed + 13, budget_s=budgets[2])
out["fasta"] = fuzz_fasta.fuzz(seed + 14, budget_s=budgets[3])
print("soak ok:", out)
# strict parity (no allowance for the reference's own error): how many float32 cases met |got - ref| <= 2e-6 + 1e-5 |ref|
print("strict parity, pipeline fuzzer:", fuzz_differential.TALLY.summary())
print("strict parity, pearson fuzzer: ", fuzz_pearson.TALLY.summary())
