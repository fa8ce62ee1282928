"""Differential fuzz of the FASTA reading path against the IMPORTED reference (build container only: needs /root/reference;
no GPU).  Random files of tests/fuzz_fasta.py — one in three with code points above U+007F, one in 200 undecodable — with
odd ASCII bytes sprinkled in (VT, FF, FS-US, CR, NUL, '>' inside lines).  For every file: a byte >= 0x80 -> the native parser
must decline (FastaNeedsText) and the package's text-mode Reader must give the reference's headers / sequences or its
exception, text included; a headerless file -> the package's fallback gives the reference's lists or exception; otherwise
the native parser's headers and lengths are the reference's, or it raises the reference's exception with its text.

    PYTHONDONTWRITEBYTECODE=1 python tools/fuzz_reader_vs_reference.py SEED N_FILES

Round 6: 26 000 files over four seeds, no divergence — after it found that a header holding a NUL byte came back
truncated through ctypes' `.value` (fixed: `.raw`)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
from seekr.fasta_reader import Reader as RefReader
from seekr_amd import _lib
from seekr_amd.fasta_reader import Reader
import fuzz_fasta
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
d = tempfile.mkdtemp(); path = os.path.join(d, "f.fa")
ODD = [b"\x0b", b"\x0c", b"\x1c", b"\x1d", b"\x1e", b"\x1f", b"\r", b"\x00", b" >", b">", b"\t"]
n_native = n_text = n_exc = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5000):
    data = fuzz_fasta.random_fasta(rng)
    for _ in range(int(rng.integers(0, 4))):  # sprinkle odd ASCII control / marker bytes
        at = int(rng.integers(0, len(data) + 1))
        data = data[:at] + ODD[int(rng.integers(0, len(ODD)))] + data[at:]
    with open(path, "wb") as fh:
        fh.write(data)
    piece = int(rng.choice([0, 1, 7, 40, 300]))
    if piece: os.environ["SEEKR_FASTA_PIECE_BYTES"] = str(piece)
    else: os.environ.pop("SEEKR_FASTA_PIECE_BYTES", None)
    try:
        rh, rs = RefReader(path).get_headers(), RefReader(path).get_seqs(); rexc = None
    except Exception as e:
        rexc = e
    high = any(b >= 0x80 for b in data)
    try:
        fa = _lib.FastaFile(path); nexc = None
    except Exception as e:
        nexc = e
    if high:
        assert isinstance(nexc, _lib.FastaNeedsText), (case, repr(nexc))
        n_text += 1
        try:
            ph, ps = Reader(path).get_headers(), Reader(path).get_seqs(); pexc = None
        except Exception as e:
            pexc = e
        assert type(pexc) is type(rexc) and str(pexc) == str(rexc), (case, repr(pexc), repr(rexc))
        if rexc is None:
            assert ph == rh and ps == rs
        continue
    if isinstance(nexc, ValueError) and "does not start with a '>' header line" in str(nexc):
        try:  # the package's fallback for headerless files: its text-mode Reader
            ph, ps = Reader(path).get_headers(), Reader(path).get_seqs(); pexc = None
        except Exception as e:
            pexc = e
        assert type(pexc) is type(rexc) and str(pexc) == str(rexc), (case, repr(pexc), repr(rexc))
        if rexc is None:
            assert ph == rh and ps == rs, (case, data[:80])
        continue
    if rexc is not None:
        n_exc += 1
        assert type(nexc) is type(rexc) and str(nexc) == str(rexc), (case, repr(nexc), repr(rexc), data[:120])
        continue
    assert nexc is None, (case, repr(nexc), data[:120])
    n_native += 1
    assert fa.headers() == rh, (case, fa.headers()[:3], rh[:3])
    assert list(fa.lengths()) == [len(s) for s in rs], (case, data[:120])
print("reader fuzz ok: native %d, text-mode %d, matching exceptions %d" % (n_native, n_text, n_exc))
