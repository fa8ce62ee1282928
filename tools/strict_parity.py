"""STRICT Pearson parity: the device against the reference's own float32 path, cell by cell, with no slack for the
reference's error (VERDICT r2 #1b).  north_star: "Pearson r matching reference within 1e-5"; the bar used throughout is
|got - ref| <= 2e-6 + 1e-5 |ref| with ref = oracle.pearson = the numpy float32 restatement of pearson.py:35-41 (the
absolute term is the float32 BLAS result's own distance from float64 on r ~ 0).

Grid: the four data classes of tools/adversarial.py x {raw values row-standardised by pearson() itself (`-uc -us -l
Log2.none` counts are a legitimate input, pearson.py:32), the same matrix after the Log2.post normalisation of
kmer_counts.py:203-209} x K in {256, 4 096, 16 384}, through the drop-in seekr_amd.pearson.pearson (default precision
f16x3) and, for comparison, the fp32 kernel.  Prints max |got - ref| / bar (must be <= 1) next to the reference's and
the device's distance from float64 truth in the same unit.

    python tools/strict_parity.py [--rows 1024] [--json out.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import seekr_oracle as orc  # noqa: E402  (checker only)
from seekr_amd import _lib  # noqa: E402


def data_classes(n, k, rng):
    sparse = np.zeros((n, k), np.float32)
    for i in range(n):  # 3..200 non-zero k-mers per row: raw per-kb counts of a short sequence
        nnz = int(rng.integers(3, min(200, k // 2)))
        cols = rng.choice(k, nnz, replace=False)
        sparse[i, cols] = rng.integers(1, 4, nnz) * np.float32(1000.0 / rng.integers(50, 900))
    yield "sparse raw counts", sparse
    yield "binomial raw counts", (rng.binomial(50, 0.05, size=(n, k)) * np.float32(2.5)).astype(np.float32)
    yield "0/1 rows", np.where(rng.random((n, k)) < 0.5, np.float32(1.0), np.float32(0.0)).astype(np.float32)
    yield "gaussian", rng.standard_normal((n, k)).astype(np.float32)


def ratios(got, ref, truth):
    bar = 2e-6 + 1e-5 * np.abs(ref)
    ok = np.isfinite(ref) & np.isfinite(truth)
    strict = float(np.max(np.where(ok, np.abs(got.astype(np.float64) - ref) / bar, 0.0)))
    e_dev = float(np.max(np.where(ok, np.abs(got.astype(np.float64) - truth) / bar, 0.0)))
    e_ref = float(np.max(np.where(ok, np.abs(ref.astype(np.float64) - truth) / bar, 0.0)))
    return strict, e_dev, e_ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--json", default="")
    ap.add_argument("--cols", type=int, nargs="*", default=[256, 4096, 16384])
    ap.add_argument("--f16f8", action="store_true", help="also run the opt-in two-product-unit precision (SKR_PREC_F16F8) and print "
                    "the storage kind its operand ended with (3 = H / X lines, 2 = degraded to the three-product split, 0 = fp32)")
    args = ap.parse_args()
    ctx = _lib.default_context()
    table = []
    print("%-22s %-9s %6s | %-28s | %-28s | ref vs f64" % ("data", "input", "K", "f16x3: strict   (vs f64)", "fp32: strict   (vs f64)"))
    for k in args.cols:
        rng = np.random.default_rng(k)
        for name, x in data_classes(args.rows, k, rng):
            for form in ("raw", "Log2.post"):
                a = x
                if form == "Log2.post":
                    with np.errstate(all="ignore"):
                        a = orc.normalize(x.copy(), True, True, "Log2.post")
                    a = np.ascontiguousarray(a[0] if isinstance(a, tuple) else a, dtype=np.float32)
                    if not np.isfinite(a).all():  # a zero-variance column: the whole matrix is NaN in the reference too
                        continue
                with np.errstate(all="ignore"):
                    ref = orc.pearson(a, a)
                    truth = orc.pearson_f64_truth(a, a)
                dev = ctx.from_numpy(a)
                row = {"data": name, "input": form, "K": k, "rows": args.rows}
                for prec in ("f16x3", "fp32"):
                    got = _lib.pearson(ctx, dev, dev, True, _lib.PRECISIONS[prec]).to_numpy()
                    strict, e_dev, e_ref = ratios(got, ref, truth)
                    row[prec] = {"strict": round(strict, 3), "vs_f64": round(e_dev, 3)}
                    row["ref_vs_f64"] = round(e_ref, 3)
                if args.f16f8:
                    op, _ = _lib.operand_fill(ctx, dev, precision=_lib.PREC_F16F8, row_standardize=True)
                    r8 = ctx.empty(dev.rows, dev.rows)
                    _lib.pearson_gemm_op(ctx, op, op, r8, symmetric=True)
                    strict, e_dev, _ = ratios(r8.to_numpy(), ref, truth)
                    row["f16f8"] = {"strict": round(strict, 3), "vs_f64": round(e_dev, 3), "operand_kind": op.kind}
                    r8.free()
                    op.free()
                dev.free()
                table.append(row)
                print("%-22s %-9s %6d | %8.3f        (%6.3f)      | %8.3f        (%6.3f)      | %6.3f"
                      % (name, form, k, row["f16x3"]["strict"], row["f16x3"]["vs_f64"], row["fp32"]["strict"], row["fp32"]["vs_f64"],
                         row["ref_vs_f64"])
                      + ("   | f16f8: strict %.3f (vs f64 %.3f), operand kind %d" % (row["f16f8"]["strict"], row["f16f8"]["vs_f64"],
                                                                                  row["f16f8"]["operand_kind"]) if args.f16f8 else ""), flush=True)
    worst = max(r["f16x3"]["strict"] for r in table)
    print("strict parity: worst f16x3 cell %.3f of the bar over %d grid points -> %s" % (worst, len(table), "ok" if worst <= 1 else "EXCEEDED"))
    if args.f16f8:
        w8 = max(r["f16f8"]["strict"] for r in table)
        print("strict parity, f16f8 (routing included): worst cell %.3f of the bar -> %s" % (w8, "ok" if w8 <= 1 else "EXCEEDED"))
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"bar": "|got - oracle.pearson| <= 2e-6 + 1e-5 |ref|", "grid": table, "worst_f16x3": worst}, fh, indent=1)
    return 0 if worst <= 1 else 1


if __name__ == "__main__":
    sys.exit(main())
