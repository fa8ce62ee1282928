"""Replay a case the pipeline fuzzer saved (gpurun_out/fuzz_fail_*.npz: sequences + parameters) and print where the
device's r stands: against the reference's float32 result, against float64, and the cell's order-sensitivity.

    python tools/replay_case.py gpurun_out/fuzz_fail_44635_24598.npz [--lib seekr_amd/libseekr_hip_r3.so]
"""
import argparse
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("case")
ap.add_argument("--lib", default="")
args = ap.parse_args()
from seekr_amd import _lib  # noqa: E402
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from oracle import seekr_oracle as orc  # noqa: E402
import parity_rule  # noqa: E402
from seekr_amd.kmer_counts import BasicCounter  # noqa: E402
from seekr_amd.pearson import pearson  # noqa: E402

d = np.load(args.case, allow_pickle=True)
seqs, tag = list(d["seqs"]), eval(str(d["tag"][0]))
print(tag, "lengths", [len(s) for s in seqs][:20])
c = BasicCounter(silent=True, k=tag["k"], alphabet=tag["alphabet"], mean=tag["mean"], std=tag["std"], log2=tag["log2"])
c.seqs = seqs
with contextlib.redirect_stdout(io.StringIO()):
    c.get_counts()
x = np.array(c.counts, dtype=np.float32)
with np.errstate(all="ignore"):
    ref, truth = orc.pearson(x, x).astype(np.float64), orc.pearson_f64_truth(x, x)
got = pearson(x, x).astype(np.float64)
ok = np.isfinite(ref) & np.isfinite(truth) & np.isfinite(got)
strict = np.where(ok, np.abs(got - ref) / parity_rule.bar_of(np.where(ok, ref, 0)), 0)
vs64 = np.where(ok, np.abs(got - truth) / parity_rule.bar_of(np.where(ok, truth, 0)), 0)
r64 = np.where(ok, np.abs(ref - truth) / parity_rule.bar_of(np.where(ok, truth, 0)), 0)
i, j = np.unravel_index(np.argmax(vs64), vs64.shape)
print("columns", x.shape[1], "worst device vs float64: %.3f bars at (%d, %d): got %.9f ref %.9f f64 %.9f; strict there %.3f; reference vs float64 there %.3f"
      % (vs64[i, j], i, j, got[i, j], ref[i, j], truth[i, j], strict[i, j], r64[i, j]))
print("order-sensitivity of that cell: %.3f" % parity_rule.order_sensitivity(x, x, [(i, j)], truth)[0])
z = parity_rule.f32_rows(x)
for r in (i, j):
    vals, cnt = np.unique(x[r], return_counts=True)
    print("row %d: %d distinct values, most common %.6g x %d of %d; max z^2 / K = %.4f" % (r, len(vals), vals[np.argmax(cnt)], cnt.max(), x.shape[1], float(np.nanmax(z[r] ** 2)) / x.shape[1]))
print("cells outside the strict bar: %d; outside the bar of float64: %d" % (int((strict > 1).sum()), int((vs64 > 1).sum())))
