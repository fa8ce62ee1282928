#!/usr/bin/env python3
"""Golden vectors G10 — FASTA files that are NOT pure ASCII — by RUNNING THE REFERENCE (same conventions as
make_golden.py; build container only, no-op without /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -W ignore /root/repo/tests/golden/make_golden_g10.py

The reference opens the file in text mode (fasta_reader.py:44), so the bytes are decoded (UTF-8 here: the
interpreter runs in UTF-8 mode, recorded in the fixture) before `strip()` (:45), `upper()` (:55,62) and
`len(seq)` (kmer_counts.py:143-144) see them: Unicode white space at line ends is stripped, a multi-byte
character is ONE position of W = len(seq) - k + 1, `upper()` may lengthen a sequence, an undecodable byte
raises UnicodeDecodeError in BasicCounter.__init__.  Stored per case: the file's bytes (hex), then either the
reader's headers / sequences, their lengths and the raw k = 2 count bits the reference produces, or the
exception type and text it raises.
"""
import contextlib
import io
import json
import locale
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

NBSP, NEL, LS, PS, IDSP, BOM = "\u00a0", "\u0085", "\u2028", "\u2029", "\u3000", "\ufeff"

# name -> file content (str: written as UTF-8; bytes: written as they stand)
CASES = [
    ("nbsp_at_line_end", ">a\nAC" + NBSP + "\nGT\n>b\nGGTTAA\n"),
    ("nbsp_inside_line", ">a\nAC" + NBSP + "GTAC\n>b\nGGTTAA\n"),
    ("nel_inside_and_at_end", ">a\nAC" + NEL + "GT\nTTGA" + NEL + "\n>b\nACGTT\n"),
    ("line_separator_inside_and_at_ends", ">a\n" + LS + "AC" + LS + "GT" + LS + "\n>b\nAC" + PS + "GTAC\n"),
    ("ideographic_space_at_line_start", ">a\n" + IDSP + "ACGTAC" + IDSP + "\n>b\nTTTTT\n"),
    ("sharp_s_lengthens_under_upper", ">a\nac\u00dfgtac\n>b\nACGTACGT\n"),
    ("dotless_i_and_ligature", ">a\nac\u0131gta\ufb01c\n>b\nacgt\u0149acgt\n"),
    ("accented_letter_in_sequence", ">a\nAC\u00e9GTAC\n>b\nacgtacgt\u00e0\n"),
    ("astral_character_is_one_position", ">a\nAC\U0001d400GTAC\n>b\nAAAA\U0001f9ecCCCC\n"),
    ("non_ascii_only_in_headers", ">g\u00e8ne 1 \u2014 \u03b1\nACGTAC\nGGTA\n>b" + NBSP + "\nTTGACA\n"),
    ("bom_makes_first_line_a_sequence", BOM + ">a\nACGTAC\n>b\nGGTTAA\n>c\nTTTTTT\n"),
    ("crlf_with_nbsp_ends", ">a\r\nACGT" + NBSP + "\r\nAC\r\n>b\r\nGGTT" + NBSP + NBSP + "\r\n"),
    ("line_of_nbsp_only_is_blank", ">a\nACGT\n" + NBSP + "\nAC\n"),
    ("header_then_nbsp_header", ">a\nACGT\n>b\n" + NBSP + ">c\nAC\n"),
    ("sequence_of_length_k_minus_1_in_characters", ">a\n\u00c9\n>b\nACGT\n"),
    ("invalid_utf8_byte", b">a\nAC\xffGT\n>b\nACGT\n"),
    ("latin1_file", b">g\xe9ne\nACGT\n"),
    ("truncated_multibyte_at_eof", b">a\nACGT\xe2\x80"),
    ("mixed_records", ">r1 " + NBSP + "\nacgt" + NBSP + "\nAC" + NEL + "GT\n>r2\n\u00df\u00dfacgtacgt\n>r3\n" + IDSP + "ttga" + LS + "ca\n"
                      ">r4\nnnnnacgtacgtacgtnnnn\n>r5\nACGTACGTAC\u00e9\n"),
]


def as_bytes(content):
    return content if isinstance(content, bytes) else content.encode("utf-8")


def main():
    if not os.path.isdir(os.path.join(REF, "seekr")):
        print("reference not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from seekr.fasta_reader import Reader
    from seekr.kmer_counts import BasicCounter
    out = {"text_encoding": locale.getpreferredencoding(False), "utf8_mode": sys.flags.utf8_mode, "cases": []}
    for name, content in CASES:
        path = "/tmp/g10_%s.fa" % name
        with open(path, "wb") as fh:
            fh.write(as_bytes(content))
        rec = {"name": name, "hex": as_bytes(content).hex()}
        try:
            with contextlib.redirect_stderr(io.StringIO()):
                c = BasicCounter(path, k=2, mean=False, std=False, log2="Log2.none", silent=True)
                c.get_counts()
            rec["headers"] = Reader(path).get_headers()
            rec["seqs"] = list(c.seqs)
            assert rec["seqs"] == Reader(path).get_seqs()
            rec["lengths"] = [len(s) for s in c.seqs]
            rec["raw_k2_bits"] = [[int(v) for v in row] for row in np.asarray(c.counts, np.float32).view(np.uint32)]
        except Exception as e:  # noqa: BLE001 - the exception IS the golden answer
            rec["exception"] = type(e).__name__
            rec["message"] = str(e)
        out["cases"].append(rec)
        print(name, "->", rec.get("exception") or rec["lengths"])
        os.remove(path)
    with open(os.path.join(HERE, "g10_non_ascii.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote g10_non_ascii.json:", len(out["cases"]), "cases")
    return 0


if __name__ == "__main__":
    sys.exit(main())
