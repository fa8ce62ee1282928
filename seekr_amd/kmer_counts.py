"""k-mer count matrices on MI355X — drop-in for `seekr.kmer_counts` (kmer_counts.py:48-262).

`BasicCounter` keeps the reference's constructor, public attributes and methods; the work is
done by hand-written HIP kernels behind `libseekr_hip.so`:

    sequences --pack 2 bit/base--> HBM --count_kmers (LDS histogram + per-kb scale)-->
    float32 [N, 4^k] in HBM --colsum_seq / elementwise (numpy-order normalisation)--> host

There is no CPU fallback: without the library or without a gfx950 device every method that
computes raises.
"""
from itertools import product

import numpy as np

from seekr_amd import _lib, multi
from seekr_amd.fasta_reader import Reader
from seekr_amd.my_tqdm import my_tqdm

_LOG2_CHOICES = ["Log2.pre", "Log2.post", "Log2.none"]

NAN_WARNING = (
    "\nWARNING: You have `np.nan` values in your counts "
    "after standardization. This is likely due to "
    "a kmer not appearing in any of your sequences. "
    "Try: \n1) using a smaller kmer size, \n2) beginning "
    "with a larger set of sequences, \n3) passing "
    "precomputed normalization vectors from a larger "
    "data set (e.g. GENCODE)."
)


def _shape_text(shape):
    return "(" + ",".join(str(d) for d in shape) + ("," if len(shape) == 1 else "") + ")"


def _column_vector(arr, n_rows, n_cols):
    """The user's mean / std as the K values `counts -= vec` subtracts from every row (kmer_counts.py:169,175), by numpy's
    broadcasting rules: a scalar, (K,), (1,), (1, K) ... all spread over the rows; what numpy refuses is refused in numpy's
    words (the shapes of the in-place operation: (N,K) vec (N,K)).  An operand that varies along the ROWS — (N, 1), (N, K):
    legal numpy, no use of the reference's — is not supported on the device."""
    if n_rows is None:
        return np.ascontiguousarray(np.broadcast_to(arr, (n_cols,)))
    shape = (int(n_rows), int(n_cols))
    try:
        together = np.broadcast_shapes(shape, arr.shape)
    except ValueError:
        raise ValueError("operands could not be broadcast together with shapes {} {} {} ".format(
            _shape_text(shape), _shape_text(arr.shape), _shape_text(shape))) from None
    if together != shape:
        raise ValueError("non-broadcastable output operand with shape {} doesn't match the broadcast shape {}".format(
            _shape_text(shape), _shape_text(together)))
    if arr.ndim >= 2 and arr.shape[-2] != 1:
        raise NotImplementedError("a mean / std operand of shape {} varies along the rows; the device path takes one value "
                                  "per column (a vector of length {})".format(_shape_text(arr.shape), n_cols))
    return np.ascontiguousarray(np.broadcast_to(arr, (1, shape[1]))[0])


def _as_device_vector(ctx, vec, n_cols, n_rows=None):
    """User mean/std vector -> 1 x K device vector; float32 stays float32, everything else is
    evaluated in float64 and rounded once, like numpy's in-place `counts -= vec`."""
    if isinstance(vec, (bool, int, float)):
        arr = np.asarray(vec, dtype=np.float32)  # a Python scalar is weak (NEP 50): it takes the matrix's type BEFORE the operation
    else:
        arr = np.asarray(vec)
    if arr.dtype == np.float32 or arr.dtype == np.float16:
        arr = arr.astype(np.float32, copy=False)
    else:
        arr = arr.astype(np.float64)
    return ctx.from_numpy(_column_vector(arr, n_rows, n_cols))


class BasicCounter:
    """Generates overlapping k-mer counts for a fasta file (reference: kmer_counts.py:48-135).

    Parameters
    ----------
    infasta : str, optional        path of the FASTA file to count
    outfile : str, optional        where `save` writes
    k : int                        k-mer length (1..7: histogram in LDS; 8..12: in the output row in HBM)
    binary : bool                  save as .npy if True, else csv
    mean, std : bool | ndarray | str
        True = compute from the data, False = skip, str = np.load(path), array = use as given
    log2 : 'Log2.post' | 'Log2.pre' | 'Log2.none'
    leave, silent : progress-bar options (cosmetic; counting is one kernel launch)
    label : bool                   label rows/columns of a csv
    alphabet : str                 column order follows `itertools.product`; 4 distinct letters take the
                                   2-bit kernels, any other string the general counting kernel
    """

    @_lib.api_call
    def __init__(self, infasta=None, outfile=None, k=6, binary=True, mean=True, std=True, log2="Log2.post",
                 leave=True, silent=False, label=False, alphabet="AGTC"):
        self.infasta = infasta
        self._seqs = None
        self._packed = None  # PackedSeqs resident in HBM (native FASTA path)
        self._fasta = None   # SEEKR_DEVICES names several GPUs: the file parsed into host memory, packed range by range later
        self.alphabet = alphabet
        # 4 distinct letters: 2 bits per base and the tuned kernels; any other alphabet string the
        # reference accepts (kmer_counts.py:120-122) goes through the general counting kernel
        self._two_bit = len(alphabet) == 4 and len(set(alphabet)) == 4
        if infasta is not None:
            # kmer_counts.py:103-105 reads the file here; errors of the reader surface here too
            if self._two_bit:
                try:
                    if multi.requested_devices():
                        self._fasta = _lib.FastaFile(infasta)  # the reader's errors surface here, as in the reference
                    else:
                        self._packed = _lib.default_context().pack_fasta(infasta, alphabet)
                except _lib.FastaNeedsText:
                    # a byte >= 0x80 somewhere in the file: the reference decodes the text before strip() / upper() /
                    # len() (fasta_reader.py:44, kmer_counts.py:143-144), so NBSP at a line end is stripped, 'ß' becomes
                    # 'SS', a two-byte letter is one position of W = len(seq) - k + 1, an undecodable byte raises
                    # UnicodeDecodeError.  The text-mode Reader does all of that by construction.
                    self._seqs = Reader(infasta).get_seqs()
                except ValueError as e:
                    # a file whose first line is not a header: the native reader refuses it, the reference slices its
                    # entry list anyway (Reader.get_seqs then returns what stands at the odd positions).  Same result
                    # here: the strings of the Python Reader are packed instead.
                    if "does not start with a '>' header line" not in str(e):
                        raise
                    self._seqs = Reader(infasta).get_seqs()
            else:
                self._seqs = Reader(infasta).get_seqs()
        self.outfile = outfile
        self.k = k
        self.binary = binary
        self.mean = np.load(mean) if isinstance(mean, str) else mean
        self.std = np.load(std) if isinstance(std, str) else std
        self.log2 = log2
        self.leave = leave
        self.silent = silent
        self.label = label
        self.counts = None
        self.alpha_len = len(alphabet)
        self.kmers = ["".join(t) for t in product(alphabet, repeat=k)]
        self.map = {kmer: col for col, kmer in enumerate(self.kmers)}

        n_seqs = self._n_seqs()
        if n_seqs == 1 and self.std is True:
            raise ValueError(
                "You cannot standardize a single sequence. "
                "Please pass the path to an std. dev. array, "
                "or use raw counts by setting std=False."
            )
        if self.log2 not in _LOG2_CHOICES:
            raise ValueError("log2 must be one of ['Log2.pre', 'Log2.post', 'Log2.none']")

    # ---- `seqs` keeps the reference's public attribute (list of str), materialised lazily ----
    @property
    def seqs(self):
        if self._seqs is None and self.infasta is not None:
            self._seqs = Reader(self.infasta).get_seqs()
        return self._seqs

    @seqs.setter
    def seqs(self, value):
        self._seqs = value
        self._packed = None  # caller-assigned sequences win over the packed file
        self._fasta = None

    def _ctx(self):
        return _lib.default_context()

    def _n_seqs(self):
        for held in (self._packed, self._fasta):
            if held is not None:
                return held.n
        return len(self._seqs) if self._seqs is not None else None

    def _packed_seqs(self):
        if self._packed is not None:
            return self._packed
        if self._fasta is not None:  # read under SEEKR_DEVICES, counted on one GPU after all
            return self._fasta.pack(self._ctx(), alphabet=self.alphabet)
        if self._seqs is None:
            raise TypeError("BasicCounter has no sequences: pass infasta or assign `seqs`")
        return self._ctx().pack(self._seqs, self.alphabet)

    def _check_k(self):
        if not isinstance(self.k, (int, np.integer)) or self.k < 1:
            raise ValueError("k must be a positive integer")

    # ---- kmer_counts.py:140-151 -----------------------------------------------------------
    @_lib.api_call
    def occurrences(self, row, seq):
        """Counts k-mers of one sequence on a per-kilobase scale into `row` (any float dtype).

        Bins of k-mers that do not occur keep their previous content, exactly like the
        reference's dict-driven assignment."""
        self._check_k()
        ctx = self._ctx()
        if self._two_bit:
            packed = ctx.pack([seq], self.alphabet)
            n = _lib.count_u32(ctx, packed, self.k).to_numpy()[0]
            if self.k <= 7:
                vals = _lib.count_per_kb(ctx, packed, self.k, dtype=np.float64).to_numpy()[0]
            else:
                # the float64 flush exists for k <= 7; above that the (sparse) row is rebuilt from the integer counts:
                # n sequential float64 additions of 1000/W, the reference's own arithmetic (kmer_counts.py:144-150)
                windows = len(seq) - self.k + 1
                if windows == 0:
                    raise ZeroDivisionError("division by zero")
                inc = 1000.0 / windows
                vals = np.zeros(n.shape, dtype=np.float64)
                hit = np.nonzero(n)[0]
                top = int(n[hit].max()) if len(hit) else 0
                partial = np.zeros(top + 1, dtype=np.float64)
                acc = 0.0
                for j in range(1, top + 1):
                    acc += inc
                    partial[j] = acc
                vals[hit] = partial[n[hit]]
        else:
            n = _lib.count_generic(ctx, [seq], self.alphabet, self.k, np.uint32).to_numpy()[0]
            vals = _lib.count_generic(ctx, [seq], self.alphabet, self.k, np.float64).to_numpy()[0]
        present = n > 0
        row[present] = vals[present]
        return row

    def _progress(self):
        """The reference iterates sequences under a tqdm bar (:153-163); counting here is one
        launch, so the bar is advanced once by the number of sequences."""
        if self.silent:
            return None
        total = self._n_seqs()
        if not self.leave:
            return my_tqdm()(total=total, desc="Kmers", leave=False)
        return my_tqdm()(total=total)

    # ---- host-array versions of the normalisation steps (kmer_counts.py:165-192) -----------
    def _device_counts(self):
        return self._ctx().from_numpy(np.asarray(self.counts))

    def _other_dtype(self):
        """`self.counts` when it is NOT float32 — a matrix assigned by hand (test_kmer_counts.py:44-90 does; a float64
        matrix read from a CSV, an integer one): the reference's methods act on whatever dtype it has, so do these
        (skr_host_colstat / skr_host_apply: numpy's arithmetic for that dtype, on the device).  None for float32, the
        dtype get_counts() produces and the tuned kernels take."""
        counts = np.asarray(self.counts)
        if counts.ndim != 2:
            raise ValueError("counts must be a 2-D matrix")
        if counts.dtype == np.float32:
            return None
        if counts.dtype not in _lib.NP_CODES:
            raise TypeError("count matrices of dtype {} are not supported on the device (float16/32/64, integers and "
                            "bool are)".format(counts.dtype))
        return counts

    @staticmethod
    def _replay(ufunc, counts, operand):
        """numpy's own verdict on the in-place `counts <op>= operand`, asked on a ZERO-row slice: type resolution, the
        same_kind casting rule, broadcasting and the read-only check all run, no cell is computed.  What the reference
        raises — UFuncTypeError for float statistics into an integer matrix (`counts -= mean`, kmer_counts.py:169,175),
        ValueError for a vector of the wrong length — is raised here as numpy words it."""
        probe = counts[:0]
        try:
            ufunc(probe, operand, out=probe)
        except ValueError as e:  # broadcasting: numpy names the shapes — of the real matrix, not of the probe
            zero, real = _shape_text(probe.shape), _shape_text(counts.shape)
            text = str(e).replace(zero, real)
            if text.startswith("non-broadcastable output operand"):  # ... and the broadcast shape it did not fit
                together = np.broadcast_shapes(counts.shape, np.shape(operand))
                text = "non-broadcastable output operand with shape {} doesn't match the broadcast shape {}".format(real, _shape_text(together))
            raise ValueError(text) from None

    @staticmethod
    def _vector_for(counts, operand):
        """The column vector of `counts <op>= operand` in the type numpy evaluates the operation in: a Python scalar is
        weak (it takes the matrix's type), an array promotes — float64 matrix: float64; float16 matrix: float32 when the
        promoted type is float16 / float32 (half arithmetic IS float32 arithmetic rounded to half), else float64."""
        cols = counts.shape[1]
        if isinstance(operand, (bool, int, float)):
            vec = np.asarray(operand, dtype=counts.dtype)
        else:
            vec = np.asarray(operand)
        column = _column_vector(vec, counts.shape[0], cols)  # (K,), (1, K), a scalar ...: one value per column
        if counts.dtype.kind != "f":
            return column.astype(np.int64)  # legal integer case only (see _replay)
        wide = counts.dtype == np.float64 or np.result_type(counts.dtype, vec.dtype).itemsize > 4
        return column.astype(np.float64 if wide else np.float32)

    def _finish_host(self, counts, work):
        """`work` (C-contiguous, holding the result) back where the reference's in-place operation leaves it."""
        if work is counts:
            return  # the device result was downloaded straight into the caller's matrix
        if isinstance(self.counts, np.ndarray):
            self.counts[...] = work
        else:
            self.counts = work

    def _in_place_any(self, counts, attr, what, ufunc, op):
        ctx = self._ctx()
        work = np.ascontiguousarray(counts)
        if getattr(self, attr) is True:  # (:168,174: the attribute is replaced first)
            if _lib.column_major_like(counts):
                setattr(self, attr, _lib.host_colstat_colmajor(ctx, counts, what))  # numpy adds such columns pairwise, whatever the dtype
            else:
                setattr(self, attr, _lib.host_colstat(ctx, work, what))
        operand = getattr(self, attr)
        self._replay(ufunc, counts, operand)
        vec = self._vector_for(counts, operand)
        has_nan = _lib.host_apply(ctx, work, op if counts.dtype.kind == "f" else "isub", vec)
        self._finish_host(counts, work)
        return has_nan

    def _store(self, dev):
        if isinstance(self.counts, np.ndarray) and self.counts.flags.writeable:  # in place, like `self.counts -= ...`
            if self.counts.flags.c_contiguous:
                dev.to_numpy(out=self.counts)
            else:
                self.counts[...] = dev.to_numpy()  # a strided view of the caller's: its cells, not its gaps
        else:
            self.counts = dev.to_numpy()

    @_lib.api_call
    def center(self):
        """Mean center counts by column (:165-169)."""
        other = self._other_dtype()
        if other is not None:
            self._in_place_any(other, "mean", "mean", np.subtract, "sub")
            return
        ctx = self._ctx()
        if self.mean is True and _lib.column_major_like(np.asarray(self.counts)):
            # a column-major float32 matrix (or a single column): numpy reduces it column by column, pairwise (_lib.column_major_like)
            self.mean = _lib.host_colstat_colmajor(ctx, np.asarray(self.counts), "mean")
        dev = self._device_counts()
        if self.mean is True:
            acc = ctx.zeros(1, dev.cols)
            _lib.colsum_seq(ctx, dev, acc)
            _lib.vec_finish(ctx, acc, dev.rows)
            self.mean = acc.vector()
            mean_dev = acc
        else:
            mean_dev = _as_device_vector(ctx, self.mean, dev.cols, dev.rows)
        _lib.apply(ctx, dev, center=mean_dev)
        self._store(dev)

    @_lib.api_call
    def standardize(self):
        """Divide out the standard deviations from columns of the count matrix (:171-187)."""
        other = self._other_dtype()
        if other is not None:
            if self._in_place_any(other, "std", "std", np.true_divide, "div"):
                print(NAN_WARNING)
            return
        ctx = self._ctx()
        if self.std is True and _lib.column_major_like(np.asarray(self.counts)):
            self.std = _lib.host_colstat_colmajor(ctx, np.asarray(self.counts), "std")
        dev = self._device_counts()
        if self.std is True:
            mprime = ctx.zeros(1, dev.cols)
            _lib.colsum_seq(ctx, dev, mprime)
            _lib.vec_finish(ctx, mprime, dev.rows)
            var = ctx.zeros(1, dev.cols)
            _lib.colsum_seq(ctx, dev, var, center2=mprime, square=True)
            _lib.vec_finish(ctx, var, dev.rows, take_sqrt=True)
            self.std = var.vector()
            std_dev = var
        else:
            std_dev = _as_device_vector(ctx, self.std, dev.cols, dev.rows)
        _, has_nan = _lib.apply(ctx, dev, scale=std_dev, want_nan=True)
        self._store(dev)
        if has_nan:
            print(NAN_WARNING)

    @_lib.api_call
    def log2_norm(self):
        """Apply a log2 transform to the count matrix (:189-192): counts += 1; log2."""
        other = self._other_dtype()
        if other is not None:
            self._replay(np.add, other, 1)  # `counts += 1`: numpy refuses it on bool
            work = np.ascontiguousarray(other)
            out = np.empty(work.shape, dtype=np.log2(other[:0]).dtype)  # numpy's result type (no cell computed)
            _lib.host_apply(self._ctx(), work, "log2p1", out=out)
            self._finish_host(other, work)  # the reference's `+= 1` is in place, the log2 a new array
            self.counts = out
            return
        ctx = self._ctx()
        dev = self._device_counts()
        if isinstance(self.counts, np.ndarray) and self.counts.flags.writeable:
            # the reference's `self.counts += 1` happens IN the caller's array before np.log2 binds a new one (:191-192): x + 1
            # on the device as x - (-1) — the same float32 rounding — and back into that array (a view keeps its gaps)
            plus = ctx.empty(dev.rows, dev.cols)
            _lib.apply(ctx, dev, y=plus, center=ctx.from_numpy(np.full((1, dev.cols), -1.0, np.float32)))
            if self.counts.flags.c_contiguous:
                plus.to_numpy(out=self.counts)
            else:
                self.counts[...] = plus.to_numpy()
        _lib.apply(ctx, dev, pre=True)
        self.counts = dev.to_numpy()

    # ---- kmer_counts.py:194-209 -------------------------------------------------------------
    @_lib.api_call
    def get_counts(self):
        """Generates k-mer counts for the sequences: count -> Log2.pre -> centre ->
        standardise -> Log2.post, all on the GPU; `self.counts` receives the float32 result."""
        self._check_k()
        devices = multi.requested_devices()
        if devices:
            # SEEKR_DEVICES: the rows in contiguous ranges, one per GPU; statistics, counts and warning as on one GPU
            if self._fasta is None and self._seqs is None and self._packed is not None and self.infasta is not None:
                self._fasta = _lib.FastaFile(self.infasta)  # read before SEEKR_DEVICES was set: onto the host again
            if self._n_seqs() is None:
                raise TypeError("BasicCounter has no sequences: pass infasta or assign `seqs`")
            bar = self._progress()
            has_nan = multi.counter_get_counts(self, devices)
            if bar is not None:
                bar.update(len(self.counts))
                bar.close()
            if has_nan:
                print(NAN_WARNING)
            return
        ctx = self._ctx()
        bar = self._progress()
        if self._two_bit:
            packed = self._packed_seqs()
            dev = _lib.count_per_kb(ctx, packed, self.k, log2_pre=(self.log2 == "Log2.pre"))
            n_counted = packed.n
        else:
            if self._seqs is None:
                raise TypeError("BasicCounter has no sequences: pass infasta or assign `seqs`")
            dev = _lib.count_generic(ctx, self._seqs, self.alphabet, self.k, np.float32,
                                     log2_pre=(self.log2 == "Log2.pre"))
            n_counted = len(self._seqs)
        if bar is not None:
            bar.update(n_counted)
            bar.close()
        mean_mode, mean_vec = 0, None
        if self.mean is True:
            mean_mode = 1
        elif self.mean is not False:
            mean_mode, mean_vec = 2, _as_device_vector(ctx, self.mean, dev.cols, dev.rows)
        std_mode, std_vec = 0, None
        if self.std is True:
            std_mode = 1
        elif self.std is not False:
            std_mode, std_vec = 2, _as_device_vector(ctx, self.std, dev.cols, dev.rows)
        # Log2.pre was fused into the counting flush, so the normaliser sees 'none' for it
        log2 = "Log2.post" if self.log2 == "Log2.post" else "Log2.none"
        mean_out, std_out, has_nan = _lib.normalize(ctx, dev, log2, mean_mode, mean_vec, std_mode, std_vec)
        if mean_out is not None:
            self.mean = mean_out.vector()
        if std_out is not None:
            self.std = std_out.vector()
        self.counts = dev.to_numpy()
        if has_nan:
            print(NAN_WARNING)

    # ---- kmer_counts.py:211-241 -------------------------------------------------------------
    def save(self, names=None):
        """Saves the counts: .npy (binary), labelled csv (label) or plain csv (%1.6f)."""
        err_msg = (
            "You cannot label a binary file. "
            'Set only one of "binary" or "label" as True. '
            "If you used `-b` from the command line, "
            "try also using `-rl`."
        )
        assert not (self.binary and self.label), err_msg
        assert self.outfile is not None, "Please provide an outfile location."
        if self.binary:
            _lib.save_npy(self.outfile, self.counts)  # np.save, streamed natively
        elif self.label:
            if names is None:
                held = self._packed if self._packed is not None else self._fasta
                names = held.headers() if held is not None else Reader(self.infasta).get_headers()
            # DataFrame(data=self.counts, index=names, columns=self.kmers).to_csv(self.outfile), natively
            _lib.save_csv_labelled(self.outfile, self.counts, names, self.kmers)
        else:
            _lib.save_csv(self.outfile, self.counts)  # np.savetxt(..., delimiter=",", fmt="%1.6f"), threaded

    # ---- kmer_counts.py:243-262 -------------------------------------------------------------
    def make_count_file(self, names=None):
        """get_counts() then save() when an outfile was given; returns the count matrix."""
        self.get_counts()
        if self.outfile is not None:
            self.save(names)
        return self.counts
