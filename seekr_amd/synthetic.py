"""Seeded synthetic transcript sets: uniform iid bases, chunked seeding so that any
row range can be generated independently (each rank makes only its own shard, the CPU baseline
only a prefix)."""
import numpy as np

CHUNK = 10_000
LETTERS = np.frombuffer(b"ACGT", dtype=np.uint8)


def synthetic_codes(seed, n_seqs, length, start=0):
    """uint8 [n_seqs, length] of base codes 0..3 for rows start .. start+n_seqs-1."""
    out = np.empty((n_seqs, length), dtype=np.uint8)
    row, filled = start, 0
    while filled < n_seqs:
        chunk = row // CHUNK
        first = chunk * CHUNK
        rng = np.random.default_rng(np.random.SeedSequence([seed, chunk]))
        block = rng.integers(0, 4, size=(CHUNK, length), dtype=np.uint8)
        take = min(CHUNK - (row - first), n_seqs - filled)
        out[filled:filled + take] = block[row - first:row - first + take]
        filled += take
        row += take
    return out


def synthetic_ascii(seed, n_seqs, length, start=0):
    """(blob uint8 [n_seqs*length] of ASCII A/C/G/T, offsets int64 [n_seqs+1])."""
    blob = LETTERS[synthetic_codes(seed, n_seqs, length, start)].reshape(-1)
    offsets = np.arange(n_seqs + 1, dtype=np.int64) * length
    return blob, offsets
