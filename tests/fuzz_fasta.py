"""Differential fuzzing of the native FASTA reader / packer (skr_seqs_from_fasta) against the oracle's
restatement of fasta_reader.py:41-63: random files with CRLF, trailing blanks, lower case, N runs,
odd characters, missing final newline, blank lines and headers without sequence (which must raise the
reference's exceptions)."""
import io
import os
import sys
import time
import contextlib
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import seekr_oracle as orc  # noqa: E402
from seekr_amd.kmer_counts import BasicCounter  # noqa: E402


def random_fasta(rng):
    n = int(rng.integers(1, 12))
    eol = "\r\n" if rng.integers(0, 4) == 0 else "\n"
    lines = []
    for i in range(n):
        lines.append(">seq%d %s" % (i, "".join(rng.choice(list("abc |,;\t"), int(rng.integers(0, 6))))))
        if rng.integers(0, 25) == 0:
            continue  # header without sequence -> AssertionError (unless last: silently dropped? the oracle decides)
        L = int(rng.choice([0, 1, 2, 3, 5, 17, 64, 65, 300, 2050])) if rng.integers(0, 3) == 0 else int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTacgtNnRYU-*"), L, p=[.22, .22, .22, .22, .02, .02, .02, .02, .01, .005, .005, .005, .005, .005, .005]))
        width = int(rng.choice([10, 60, 61, 80, 10 ** 6]))
        chunks = [seq[j:j + width] for j in range(0, len(seq), width)] or [""]
        for c in chunks:
            if c == "" and rng.integers(0, 2):
                continue
            lines.append(c + (" " * int(rng.integers(0, 3)) if rng.integers(0, 6) == 0 else ""))
        if rng.integers(0, 40) == 0:
            lines.append("")  # blank line -> IndexError
    text = eol.join(lines)
    if rng.integers(0, 3):
        text += eol
    return text


def fuzz(seed, budget_s=30.0, max_cases=10 ** 9):
    rng = np.random.default_rng(seed)
    d = tempfile.mkdtemp()
    path = os.path.join(d, "f.fa")
    t0, n_cases = time.time(), 0
    while time.time() - t0 < budget_s and n_cases < max_cases:
        text = random_fasta(rng)
        with open(path, "w", newline="") as fh:
            fh.write(text)
        k = int(rng.integers(1, 4))
        # the reader parses big files in pieces on several threads and stitches them; force that on small files
        piece = int(rng.choice([0, 1, 7, 40, 300]))
        if piece:
            os.environ["SEEKR_FASTA_PIECE_BYTES"] = str(piece)
        else:
            os.environ.pop("SEEKR_FASTA_PIECE_BYTES", None)
        try:
            headers, seqs = orc.read_fasta(path)
            want_exc = None
        except Exception as e:  # noqa: BLE001
            want_exc = type(e)
        if want_exc is None and any(len(s) == k - 1 for s in seqs):
            want_exc = ZeroDivisionError
        try:
            c = BasicCounter(path, k=k, mean=False, std=False, log2="Log2.none", silent=True)
            with contextlib.redirect_stdout(io.StringIO()):
                c.get_counts()
            got_exc = None
        except Exception as e:  # noqa: BLE001
            got_exc = type(e)
        try:
            assert got_exc == want_exc, ("exception", got_exc, want_exc)
            if want_exc is None:
                assert list(c.seqs) == list(seqs), "sequences differ"
                raw = orc.raw_counts(seqs, k)
                assert np.array_equal(np.ascontiguousarray(c.counts).view(np.uint32), raw.view(np.uint32)), "counts differ"
        except AssertionError:
            out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "fuzz_fasta_fail_%d_%d.fa" % (seed, n_cases)), "w", newline="") as fh:
                fh.write(text)
            print("piece bytes", piece, repr(text[:400]))
            raise
        n_cases += 1
    os.environ.pop("SEEKR_FASTA_PIECE_BYTES", None)
    return n_cases


if __name__ == "__main__":
    n = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
    print("fasta fuzz ok: %d cases" % n)
