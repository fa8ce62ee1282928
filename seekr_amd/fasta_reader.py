"""Host-side FASTA access with the behaviour of `seekr.fasta_reader.Reader` (fasta_reader.py:9-109).

Python strings for the callers that want them (headers, sequences, rewritten headers).  Counting does
not go through here: `skr_seqs_from_fasta` in libseekr_hip parses and packs the file natively; this
module is the small pure-Python counterpart that keeps the reference's class usable (`seqs` of a
BasicCounter, labelled CSV rows, `supply_basic_header`).
"""

HEADER_MARK = ">"


def records(stripped_lines):
    """The entries of the file in encounter order, as fasta_reader.py:47-63 builds them: a header line as it stands,
    the sequence lines up to the next header joined and upper-cased.  For a well-formed file that is
    [h0, s0, h1, s1, ...].  The reference's failure modes are kept: an empty line fails on `line[0]` (IndexError); a
    header that follows another header — anywhere but on the first line — trips the assertion; a header at the very
    end gets an empty sequence; and a file whose first line is NOT a header simply starts with a sequence entry
    ([s0, h1, s1, ...]), which `get_headers` / `get_seqs` then slice as they slice everything (the reference does not
    notice either)."""
    entries, pieces = [], []
    for index, line in enumerate(stripped_lines):
        if line[0] != HEADER_MARK:
            pieces.append(line)
            continue
        if pieces:
            entries.append("".join(pieces).upper())
            pieces = []
        elif index:
            raise AssertionError("There may be a header without a sequence at line {}.".format(index))
        entries.append(line)
    entries.append("".join(pieces).upper())
    return entries


class Reader:
    """`Reader(infasta, outfasta=None, names=None)`; `data` is filled by the `get_*` methods."""

    def __init__(self, infasta=None, outfasta=None, names=None):
        self.infasta, self.outfasta, self.names = infasta, outfasta, names
        self.data = None

    def _read_data(self):
        """`data` = the file's lines, stripped (fasta_reader.py:41-45; text mode: the locale's decoding, as there)."""
        with open(self.infasta) as fh:
            self.data = [raw.strip() for raw in fh]

    def _upper_seq_per_line(self):
        """`data` = one header entry, one upper-cased sequence entry, ... (fasta_reader.py:47-63)."""
        self.data = records(self.data)

    def _load(self):
        self._read_data()
        self._upper_seq_per_line()
        return self.data[0::2], self.data[1::2]

    def get_lines(self):
        self._load()
        return self.data

    def get_headers(self):
        self._load()
        return self.data[0::2]

    def get_seqs(self):
        self._load()
        return self.data[1::2]

    def get_data(self, tuples_only=False):
        self._load()
        headers, sequences = self.data[0::2], self.data[1::2]
        tuples = zip(headers, sequences)
        return tuples if tuples_only else (tuples, headers, sequences)

    def supply_basic_header(self):
        """GENCODE-looking headers `>||||name||length|` from `names` (default: the file's own headers),
        fasta_reader.py:90-103.  Returns the new line list; `data` itself is left alone."""
        if self.names is None:
            self.names = iter(self.get_headers())
        lines = []
        for at, line in enumerate(self.data):
            if line[0] == HEADER_MARK:
                name = next(self.names).strip(HEADER_MARK)
                line = "{0}||||{1}||{2}|".format(HEADER_MARK, name, len(self.data[at + 1]))
            lines.append(line)
        return lines

    def save(self):
        """Writes `data`, one entry per line, to `outfasta` (fasta_reader.py:105-109)."""
        with open(self.outfasta, "w") as fh:
            fh.write("".join(entry + "\n" for entry in self.data))
