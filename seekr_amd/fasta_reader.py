"""FASTA reader with the semantics of the reference `seekr.fasta_reader.Reader`
(fasta_reader.py:9-109): same constructor, same methods, same error behaviour.

This class produces Python strings for callers that want them (`get_seqs`, `get_headers`,
`get_data`, header rewriting).  The counting path itself does not go through it: the
native reader in libseekr_hip (`skr_seqs_from_fasta`) parses and packs the file directly.
"""


class Reader:
    """Normalises a FASTA file: one upper-case sequence string per header.

    Parameters mirror the reference (fasta_reader.py:34-39): `infasta` path to read,
    `outfasta` path used by `save`, `names` iterable of replacement names used by
    `supply_basic_header`.  `data` holds the header/sequence lines after a `get_*` call.
    """

    def __init__(self, infasta=None, outfasta=None, names=None):
        self.infasta = infasta
        self.outfasta = outfasta
        self.names = names
        self.data = None

    # fasta_reader.py:41-45
    def _read_data(self):
        with open(self.infasta) as handle:
            self.data = [line.strip() for line in handle]

    # fasta_reader.py:47-63
    def _upper_seq_per_line(self):
        records = []
        parts = []
        for lineno, text in enumerate(self.data):
            if text[0] == ">":  # a blank line raises IndexError here, as upstream
                if parts:
                    records.append("".join(parts).upper())
                    parts = []
                else:
                    assert lineno == 0, "There may be a header without a sequence at line {}.".format(lineno)
                records.append(text)
            elif text:
                parts.append(text)
        records.append("".join(parts).upper())
        self.data = records

    def get_lines(self):
        self._read_data()
        self._upper_seq_per_line()
        return self.data

    def get_seqs(self):
        return self.get_lines()[1::2]

    def get_headers(self):
        return self.get_lines()[0::2]

    def get_data(self, tuples_only=False):
        lines = self.get_lines()
        headers, seqs = lines[0::2], lines[1::2]
        pairs = zip(headers, seqs)
        if tuples_only:
            return pairs
        return pairs, headers, seqs

    # fasta_reader.py:90-103
    def supply_basic_header(self):
        """Rewrite headers GENCODE-style, keeping only a common name and the length."""
        if self.names is None:
            self.names = iter(self.get_headers())
        rewritten = []
        for pos, text in enumerate(self.data):
            if text[0] == ">":
                label = next(self.names).strip(">")
                rewritten.append(">||||{}||{}|".format(label, len(self.data[pos + 1])))
            else:
                rewritten.append(text)
        return rewritten

    # fasta_reader.py:105-109
    def save(self):
        with open(self.outfasta, "w") as handle:
            handle.writelines(line + "\n" for line in self.data)
