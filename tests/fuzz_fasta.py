"""Differential fuzzing of the native FASTA reader / packer (skr_seqs_from_fasta) against the oracle's
restatement of fasta_reader.py:41-63: random files with CRLF, trailing blanks, lower case, N runs,
odd characters, missing final newline, blank lines and headers without sequence (which must raise the
reference's exceptions).  One file in three is not ASCII: 2 % of its header and sequence characters are code points
above U+007F (Unicode white space that strip() removes, letters whose upper() is longer or lands in ASCII, astral
characters, a BOM) — such a file must take the text-mode reader (fasta_reader.py:44 decodes before strip / upper / len)
— and one file in 200 holds a byte that is not UTF-8 at all (UnicodeDecodeError, like the reference)."""
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


WIDE = ["\u00a0", "\u0085", "\u2028", "\u2029", "\u3000", "\ufeff", "\u00e9", "\u00df", "\u0131", "\ufb01", "\u0149",
        "\u03a9", "\u00ff", "\U0001d400", "\U0001f9ec", "\u1680", "\u200b"]


def sprinkle(rng, text, share=0.02):
    """`share` of the characters of `text` replaced by code points above U+007F."""
    if not text:
        return text
    chars = list(text)
    for at in np.nonzero(rng.random(len(chars)) < share)[0]:
        chars[at] = WIDE[int(rng.integers(0, len(WIDE)))]
    return "".join(chars)


def random_fasta(rng):
    """The file as BYTES (UTF-8 unless an undecodable byte was planted)."""
    wide = rng.integers(0, 3) == 0
    text = random_text(rng, wide)
    data = text.encode("utf-8")
    if rng.integers(0, 200) == 0 and data:
        at = int(rng.integers(0, len(data)))
        data = data[:at] + bytes([int(rng.choice([0xff, 0xc3, 0x80, 0xe2]))]) + data[at:]
    return data


def random_text(rng, wide):
    n = int(rng.integers(1, 12))
    eol = "\r\n" if rng.integers(0, 4) == 0 else "\n"
    lines = []
    for i in range(n):
        head = ">seq%d %s" % (i, "".join(rng.choice(list("abc |,;\t"), int(rng.integers(0, 6)))))
        lines.append(head[0] + sprinkle(rng, head[1:], 0.05) if wide else head)
        if rng.integers(0, 25) == 0:
            continue  # header without sequence -> AssertionError (unless last: silently dropped? the oracle decides)
        L = int(rng.choice([0, 1, 2, 3, 5, 17, 64, 65, 300, 2050])) if rng.integers(0, 3) == 0 else int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTacgtNnRYU-*"), L, p=[.22, .22, .22, .22, .02, .02, .02, .02, .01, .005, .005, .005, .005, .005, .005]))
        width = int(rng.choice([10, 60, 61, 80, 10 ** 6]))
        chunks = [seq[j:j + width] for j in range(0, len(seq), width)] or [""]
        for c in chunks:
            if c == "" and rng.integers(0, 2):
                continue
            if wide:
                c = sprinkle(rng, c)
            blank = "\u00a0" if wide and rng.integers(0, 3) == 0 else " "
            lines.append(c + (blank * int(rng.integers(0, 3)) if rng.integers(0, 6) == 0 else ""))
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
        with open(path, "wb") as fh:
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
            with open(os.path.join(out, "fuzz_fasta_fail_%d_%d.fa" % (seed, n_cases)), "wb") as fh:
                fh.write(text)
            print("piece bytes", piece, repr(text[:400]))
            raise
        n_cases += 1
    os.environ.pop("SEEKR_FASTA_PIECE_BYTES", None)
    return n_cases


if __name__ == "__main__":
    n = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
    print("fasta fuzz ok: %d cases" % n)
