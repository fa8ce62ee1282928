"""The Pearson parity rule of the differential fuzzers — test infrastructure.

north_star: "Pearson r matching reference within 1e-5".  The reference is `np.inner` of float32 rows
(`seekr/pearson.py:35-41`), i.e. whatever float32 summation order the host's BLAS kernel happens to use; on some
inputs that order alone moves r by several bars (`tests/golden/refspread.json`: the imported reference against ITSELF
under OPENBLAS_CORETYPE / thread count / operand order, up to 2.3 bars apart on one input).  The rule, per cell (i, j):

  strict      |got - ref| <= 2e-6 + 1e-5 |ref|          ref = oracle.pearson (numpy float32) — no allowance;
  otherwise   the cell must be ORDER-SENSITIVE, a property of the INPUT alone: evaluated on the reference's own float32
              standardised rows, at least one of nine plain float32 evaluations of sum_k z_ik z_jk lies TAU = 0.2 bars or
              more from the float64 value — four with separately rounded products (one accumulator with k ascending, with
              k descending, numpy's pairwise tree, 16 strided accumulators as a SIMD reduction keeps them) and five shaped
              like a BLAS sgemm micro-kernel: ONE accumulator per cell fed by fused multiply-adds in k order, restarted
              every Q = 128 / 256 / 384 / 512 columns (or never) with the block sums added to C — OpenBLAS's GEMM_Q
              blocking, which differs by core type.  A float32 `np.inner` is then not a number but a range, and the
              yardstick is float64:   |got - f64| <= 2e-6 + 1e-5 |f64|.
              (The BLAS-shaped orders were added after the first long soak: on a K = 3 125 cell the reference on the GPU
              box was 1.3 bars from float64 — 0.47 / 1.30 / 1.19 / 0.61 bars under OPENBLAS_CORETYPE = Haswell / SkylakeX /
              Sandybridge / Nehalem on the build host — the device 0.11, and none of the four rounded-product orders
              moved by more than 0.195; the FMA chain with Q = 384 reproduces 1.2.)

A cell that fails strict and is not order-sensitive is a FAILURE (no slack proportional to the reference's error any
more).  `order_sensitivity` is only evaluated for cells that fail strict, so the fuzzers keep their throughput.

The reach of the clause (measured numbers in DESIGN.md section 2): it is evaluated ONLY on
cells that already failed strict parity, so what matters is not how many cells of an input would qualify (0 % on
config-2 normalised counts, about a third on raw counts, nearly all wide r ~ 1 cells) but how many strict failures it
excuses and what a stricter TAU would do to them: every excused cell is kept with its sensitivity (judge()["excused"]),
the tally (tests/strict_tally.py) reports per width the share of strict-failing cells that used the clause and, for
TAU' = 0.1 / 0.2 / 0.5 / 1.0, how many of them would turn into failures — the TAU sweep of a recorded soak.
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


FMA_BLOCKS = (0, 128, 256, 384, 512)  # 0: one chain over all of K


def _fma_chain(pe, block):
    """One float32 accumulator per cell, fused multiply-adds in k order (pe: the EXACT products, float64 [cells, K] —
    two float32 factors have 48 significant bits), restarted every `block` columns, block sums added to C in float32."""
    n, K = pe.shape
    block = block or K
    C = np.zeros(n, np.float32)
    for k0 in range(0, K, block):
        acc = np.zeros(n, np.float32)
        for k in range(k0, min(K, k0 + block)):
            acc = (acc.astype(np.float64) + pe[:, k]).astype(np.float32)
        C = (C.astype(np.float64) + acc).astype(np.float32) if k0 else acc
    return C


def alt_order_values(za, zb, i, js):
    """float32 evaluations of sum_k za[i,k] zb[j,k] / K for j in js in nine summation orders -> float64 [9, len(js)]."""
    K = za.shape[1]
    with np.errstate(all="ignore"):
        pe = za[i][None, :].astype(np.float64) * zb[js].astype(np.float64)  # exact products
        p = pe.astype(np.float32)                                          # rounded products
        res = [np.cumsum(p, axis=1, dtype=np.float32)[:, -1],             # one accumulator, k ascending
               np.cumsum(p[:, ::-1], axis=1, dtype=np.float32)[:, -1],    # one accumulator, k descending
               np.add.reduce(p, axis=1, dtype=np.float32)]                # numpy's pairwise tree
        q = np.pad(p, ((0, 0), (0, (-K) % 16))).reshape(len(js), -1, 16)
        lanes = np.cumsum(q, axis=1, dtype=np.float32)[:, -1, :]          # 16 strided accumulators, folded at the end
        res.append(np.add.reduce(lanes, axis=1, dtype=np.float32))
        res += [_fma_chain(pe, b) for b in FMA_BLOCKS if b < K or b == 0]  # BLAS-shaped: FMA chain, K blocked by Q
        return np.stack(res).astype(np.float64) / K


def order_sensitivity(a, b, cells, truth, row_standardize=True, unit=1.0):
    """For each (i, j) of `cells`: the largest distance of the nine float32 orders from the float64 value, in bars."""
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
           "failures": [], "worst_sensitivity": 0.0, "worst_vs_f64": 0.0, "n_cells": int(np.count_nonzero(ok)),
           "excused": []}  # (strict ratio, sensitivity, device vs float64, REFERENCE vs float64) of every cell the clause let through
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
            res["excused"].append((float(strict[i, j]), float(s), float(e64),
                                   float(abs(ref[i, j] - truth[i, j]) / bar_of(truth[i, j], unit))))
            if not e64 <= 1.0:
                res["failures"].append((int(i), int(j), "order-sensitive cell, but not within the bar of float64",
                                        dict(got=got[i, j], ref=ref[i, j], f64=truth[i, j], vs_f64=e64, sensitivity=float(s))))
        else:
            res["failures"].append((int(i), int(j), "strict parity fails on a cell that is NOT order-sensitive",
                                    dict(got=got[i, j], ref=ref[i, j], f64=truth[i, j], strict=float(strict[i, j]), vs_f64=e64,
                                         sensitivity=float(s))))
    return res
