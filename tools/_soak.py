import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import fuzz_pearson, fuzz_differential, fuzz_consumers, fuzz_fasta
t = int(time.time())
out = {}
out["differential"] = fuzz_differential.fuzz(t % 100000 + 11, budget_s=900.0)
out["pearson"] = fuzz_pearson.fuzz(t % 100000 + 12, budget_s=600.0)
out["consumers"] = fuzz_consumers.fuzz(t % 100000 + 13, budget_s=300.0)
out["fasta"] = fuzz_fasta.fuzz(t % 100000 + 14, budget_s=240.0)
print("soak ok:", out)
