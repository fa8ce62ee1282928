"""All-pairs Pearson correlation of k-mer profiles on MI355X — drop-in for `seekr.pearson`
(pearson.py:32-44): same signature, same dtype promotion, same NaN behaviour.

float32 inputs run on the matrix cores in the arithmetic named by `SEEKR_PRECISION`:
  f16x3 (default)   split-fp16, 3 products/k: two 11-bit halves carry 22 significand bits, so
                    the operands are float32-grade; error vs float64 ~2e-7 off the diagonal
                    (below numpy's own float32 noise), ~3e-6 on r ~ 1.  Row-standardised rows
                    only (|z| <= sqrt(K) fits fp16); without row standardisation -> fp32.
  bf16x3            split-bf16, 3 products/k: ~5 % faster, but two 8-bit halves carry 16 bits
                    and the residual adds up on rows with few distinct values (raw counts of
                    short sequences): up to ~2e-6 off the diagonal, 1e-5 on it
  fp32              f32-input MFMA, blocked accumulation: 5e-7 on smooth data, ~6x slower
(tools/adversarial.py prints the errors of the three on worst-case inputs; bf16x3 is kept as the CONTROL arm of that
choice — the accuracy tables of DESIGN section 2 are measured against it — not as a recommendation; bf16x4 was retired in round 5; the bf16 choices use
the fp32 kernel below 1024 columns, f16x3 below 64).  Anything else (float64, integers, DataFrames read from
CSV) is promoted to float64 like numpy does and runs on the f64 MFMA.  A float16 matrix — the reference would standardise
and multiply it in half precision (1e-3 noise) — is evaluated in float64 and the result ROUNDED to the dtype the reference
returns (float16 with float16: float16; with float32: float32): the reference's dtype, better than its bits.
"""
import os

import numpy as np

from seekr_amd import _lib, multi


def _as_matrix(counts):
    arr = counts.values if hasattr(counts, "values") and not isinstance(counts, np.ndarray) else counts
    arr = np.asarray(arr)
    if arr.ndim != 2:
        raise ValueError("pearson expects 2-D count matrices, got shape {}".format(arr.shape))
    return arr


def _precision_for(dtype, row_standardize=True):
    if dtype == np.float64:
        return _lib.PREC_F64
    name = os.environ.get("SEEKR_PRECISION", "f16x3").lower()
    if name not in ("fp32", "bf16x3", "f16x3", "f16f8"):
        raise ValueError("SEEKR_PRECISION must be fp32, bf16x3, f16x3 or f16f8, got {!r}".format(name))
    if name in ("f16x3", "f16f8") and not row_standardize:
        name = "fp32"  # arbitrary magnitudes: outside fp16's range / inside its subnormals
    return _lib.PRECISIONS[name]


def _result_dtype(c1, c2):
    """What np.inner of the two row-standardised operands has in the reference: every float dtype keeps itself through
    np.mean / np.std, integers and bool become float64."""
    kinds = [c.dtype if c.dtype.kind == "f" else np.dtype(np.float64) for c in (c1, c2)]
    return np.result_type(*kinds)


def _operands(counts1, counts2):
    c1, c2 = _as_matrix(counts1), _as_matrix(counts2)
    if c1.shape[1] != c2.shape[1]:
        raise ValueError("shapes {} and {} not aligned: {} (dim 1) != {} (dim 1)".format(
            c1.shape, c2.shape, c1.shape[1], c2.shape[1]))
    # numpy promotion of the reference: f32 with f32 stays f32, everything else becomes f64
    # (np.mean of an integer matrix is float64)
    w1 = np.float32 if c1.dtype == np.float32 else np.float64
    w2 = np.float32 if c2.dtype == np.float32 else np.float64
    same = c1 is c2 or (c1.shape == c2.shape and c1.dtype == c2.dtype and c1.ctypes.data == c2.ctypes.data
                        and c1.strides == c2.strides)
    return c1, c2, w1, w2, same


def _fits_one_block(ctx, c1, c2, same, w1, w2):
    """Whether r, the inputs and their prepared operands fit the GPU at once (then: one upload, one launch, one download —
    the path of rounds 1-4); otherwise r is produced in row stripes (multi.run_pearson)."""
    out_item = 4 if (w1 == np.float32 and w2 == np.float32) else 8
    rows_in = c1.shape[0] + (0 if same else c2.shape[0])
    need = c1.shape[0] * c2.shape[0] * out_item + 2 * rows_in * (c1.shape[1] + 32) * out_item
    return need <= 0.8 * ctx.mem_info()[0]


def _striped(c1, c2, w1, w2, same, row_standardize, devices, outfile_only=None):
    precision = _precision_for(np.dtype(np.float32 if (w1 == np.float32 and w2 == np.float32) else np.float64), row_standardize)
    return multi.run_pearson(c1, None if same else c2, w1, w2, row_standardize, precision, devices, outfile=outfile_only)


@_lib.api_call
def pearson(counts1, counts2, row_standardize=True, outfile=None):
    """Calculates a column standardized Pearson correlation matrix (pearson.py:32-44).

    r[i, j] = <z1_i, z2_j> / K with z the row-standardised counts (population std, computed
    on the centred row) when `row_standardize`, else the raw inner product / K.

    SEEKR_DEVICES naming several GPUs: the rows of counts1 are cut into one range per GPU, every GPU uploads its range,
    the prepared rows of counts2 are all-gathered, and every GPU writes its row block of r into the result over its own
    PCIe link (seekr_amd.multi) — the same bits as on one GPU.  A result that does not fit the GPU's memory is produced
    in row stripes (one GPU or several); SEEKR_PEARSON_STRIPE_ROWS forces a stripe height.
    """
    c1, c2, w1, w2, same = _operands(counts1, counts2)
    devices = multi.requested_devices()
    if devices or multi.forced_stripe_rows() or not _fits_one_block(_lib.default_context(), c1, c2, same, w1, w2):
        dist = _striped(c1, c2, w1, w2, same, row_standardize, devices)
        if dist.dtype != _result_dtype(c1, c2):
            dist = dist.astype(_result_dtype(c1, c2))
        if outfile:
            _lib.save_npy(outfile, dist)
        return dist
    ctx = _lib.default_context()
    d1 = ctx.from_numpy(c1.astype(w1, copy=False))
    d2 = d1 if same else ctx.from_numpy(c2.astype(w2, copy=False))
    if w1 == w2:
        r = _lib.pearson(ctx, d1, d2, row_standardize=row_standardize, precision=_precision_for(np.dtype(w1), row_standardize))
    else:
        # mixed float32 / float64: the reference standardises each operand in its own dtype and
        # only the inner product promotes (pearson.py:35-41)
        if row_standardize:
            d1, d2 = _lib.row_standardize(ctx, d1), _lib.row_standardize(ctx, d2)
        if w1 == np.float32:
            d1 = ctx.from_numpy(d1.to_numpy().astype(np.float64))
        else:
            d2 = ctx.from_numpy(d2.to_numpy().astype(np.float64))
        r = _lib.pearson(ctx, d1, d2, row_standardize=False, precision=_lib.PREC_F64)
    dist = r.to_numpy()
    if dist.dtype != _result_dtype(c1, c2):  # a float16 operand (module docstring)
        dist = dist.astype(_result_dtype(c1, c2))
    if outfile:
        _lib.save_npy(outfile, dist)
    return dist


@_lib.api_call
def pearson_to_file(counts1, counts2, outfile, row_standardize=True):
    """pearson(counts1, counts2, outfile=outfile) for a caller that does not want the matrix back (`seekr_pearson -bo`,
    console_scripts.py:632-633): r goes from the GPU(s) to its place in the .npy file stripe by stripe and is never held in
    host memory — neither once nor, as np.save of a returned array would, twice."""
    c1, c2, w1, w2, same = _operands(counts1, counts2)
    if _result_dtype(c1, c2).itemsize < (4 if (w1 == np.float32 and w2 == np.float32) else 8):  # a float16 operand: through the host
        pearson(counts1, counts2, row_standardize=row_standardize, outfile=outfile)
        return
    _striped(c1, c2, w1, w2, same, row_standardize, multi.requested_devices(), outfile_only=outfile)
