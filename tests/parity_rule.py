"""The Pearson parity rule of the differential fuzzers — test infrastructure.

north_star: "Pearson r matching reference within 1e-5".  The reference is `np.inner` of float32 rows
(`seekr/pearson.py:35-41`), i.e. whatever float32 summation order the host's BLAS kernel happens to use; on some
inputs that order alone moves r by several bars (`tests/golden/refspread.json`: the imported reference against ITSELF
under OPENBLAS_CORETYPE / thread count / operand order, up to 4.7 bars apart on one input).  The rule, per cell (i, j):

  strict      |got - ref| <= 2e-6 + 1e-5 |ref|          ref = oracle.pearson (numpy float32) — no allowance;
  otherwise   the cell must be ORDER-SENSITIVE, a property of the INPUT alone: evaluated on the reference's own float32
              standardised rows, at least one of four plain float32 summation orders of sum_k z_ik z_jk (sequential,
              sequential reversed, numpy's pairwise tree, 16 strided accumulators as a SIMD kernel keeps them) lies
              TAU = 0.2 bars or more from the float64 value — a float32 `np.inner` is then not a number but a
              range, and the yardstick is float64:   |got - f64| <= 2e-6 + 1e-5 |f64|.

A cell that fails strict and is not order-sensitive is a FAILURE (no slack proportional to the reference's error any
more).  `order_sensitivity` is only evaluated for cells that fail strict, so the fuzzers keep their throughput.
"""
import numpy as np

TAU = 0.2


def bar_of(x, unit=1.0):
    return (2e-6 + 1e-5 * np.abs(x)) * unit


def f32_rows(x, row_standardize=True):
    """The reference's own float32 row standardisation (pearson.py:35-38)."""
    x = np.asarray(x, np.float32)
    if not row_standardize:
        return x
    with np.errstate(all="ignore"):
        x = (x.T - np.mean(x, axis=1)).T
        x = (x.T / np.std(x, axis=1)).T
    return x


def alt_order_values(za, zb, i, js):
    """float32 evaluations of sum_k za[i,k] zb[j,k] / K for j in js in four summation orders -> float64 [4, len(js)]."""
    K = za.shape[1]
    with np.errstate(all="ignore"):
        p = (za[i][None, :] * zb[js]).astype(np.float32)               # rounded products
        res = [np.cumsum(p, axis=1, dtype=np.float32)[:, -1],             # one accumulator, k ascending
               np.cumsum(p[:, ::-1], axis=1, dtype=np.float32)[:, -1],    # one accumulator, k descending
               np.add.reduce(p, axis=1, dtype=np.float32)]                # numpy's pairwise tree
        q = np.pad(p, ((0, 0), (0, (-K) % 16))).reshape(len(js), -1, 16)
        lanes = np.cumsum(q, axis=1, dtype=np.float32)[:, -1, :]          # 16 strided accumulators, folded at the end
        res.append(np.add.reduce(lanes, axis=1, dtype=np.float32))
        return np.stack(res).astype(np.float64) / K


def order_sensitivity(a, b, cells, truth, row_standardize=True, unit=1.0):
    """For each (i, j) of `cells`: the largest distance of the four float32 orders from the float64 value, in bars."""
    za = f32_rows(a, row_standardize)
    zb = za if b is a else f32_rows(b, row_standardize)
    cells = np.asarray(cells).reshape(-1, 2)
    out = np.zeros(len(cells))
    for i in np.unique(cells[:, 0]):
        pick = np.nonzero(cells[:, 0] == i)[0]
        js = cells[pick, 1]
        alt = alt_order_values(za, zb, int(i), js)
        t = truth[int(i), js]
        with np.errstate(all="ignore"):
            out[pick] = np.nanmax(np.abs(alt - t[None, :]) / bar_of(t, unit)[None, :], axis=0)
    return out


def judge(got, ref, truth, ok, a, b, row_standardize=True, unit=1.0):
    """Apply the rule to the cells `ok`.  Returns a dict:
    strict_ratio (worst |got - ref| / bar over ok cells), n_strict_fail, n_order_sensitive (of those),
    failures: list of (i, j, reason, numbers) — empty when the case passes."""
    got, ref, truth = (np.asarray(x, dtype=np.float64) for x in (got, ref, truth))
    with np.errstate(all="ignore"):
        strict = np.where(ok, np.abs(got - ref) / bar_of(np.where(ok, ref, 0.0), unit), 0.0)
    res = {"strict_ratio": float(strict.max()) if strict.size else 0.0, "n_strict_fail": 0, "n_order_sensitive": 0,
           "failures": [], "worst_sensitivity": 0.0, "worst_vs_f64": 0.0}
    cells = np.argwhere(strict > 1.0)
    if not len(cells):
        return res
    res["n_strict_fail"] = int(len(cells))
    sens = order_sensitivity(a, b, cells, truth, row_standardize, unit)
    for (i, j), s in zip(cells, sens):
        e64 = abs(got[i, j] - truth[i, j]) / bar_of(truth[i, j], unit)
        if s >= TAU:
            res["n_order_sensitive"] += 1
            res["worst_sensitivity"] = max(res["worst_sensitivity"], float(s))
            res["worst_vs_f64"] = max(res["worst_vs_f64"], float(e64))
            if not e64 <= 1.0:
                res["failures"].append((int(i), int(j), "order-sensitive cell, but not within the bar of float64",
                                        dict(got=got[i, j], ref=ref[i, j], f64=truth[i, j], vs_f64=e64, sensitivity=float(s))))
        else:
            res["failures"].append((int(i), int(j), "strict parity fails on a cell that is NOT order-sensitive",
                                    dict(got=got[i, j], ref=ref[i, j], f64=truth[i, j], strict=float(strict[i, j]), vs_f64=e64,
                                         sensitivity=float(s))))
    return res
