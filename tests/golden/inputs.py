"""Seeded inputs shared by make_golden.py (which feeds them to the reference) and by
the test-suite (which feeds the same inputs to the oracle and to the HIP path)."""
import numpy as np

# ---- inputs the build owns (the tests rebuild these from the same seeds) ----------
EXAMPLE_FA = (">SEQ1\nAAAAAA\n>SEQ2\nTTTTTTGGGGGG\n>SEQ3\nATGCATGCATGCATGC\n"
              ">SEQ4\nATCATGCTAGCTAGCTACTCGATGCATGCATGCATCGATCGACTGATCGATCGATCGACTGACTGACTGACTGAC\n"
              ">SEQ5\nATATATATATTAATATATATATATATTGACTGCATGCGCTGCATTAGCTATGCACCAACAGTCAGCGCTAGCCGCG\n")


def skewed_set(seed, n, lo=240, hi=670):
    """n sequences of length lo..hi with a non-uniform, per-sequence base composition."""
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(n):
        length = int(rng.integers(lo, hi + 1))
        p = rng.dirichlet([3.0, 2.0, 2.5, 1.5])
        seqs.append("".join(rng.choice(list("ACGT"), size=length, p=p)))
    return seqs


def synth_2000():
    codes = np.random.default_rng(0).integers(0, 4, (2000, 2000), dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [r.tobytes().decode() for r in letters[codes]]


def big_count_matrix(seed=5, n=50_000, k_cols=4096, w=1995):
    """Per-kb float32 matrix generated directly from integer counts (skips slow counting)."""
    rng = np.random.default_rng(seed)
    n_int = rng.binomial(w, 1.0 / k_cols, size=(n, k_cols)).astype(np.float64)
    return (n_int * (1000.0 / w)).astype(np.float32)


def write_fasta(path, seqs, width=None, crlf=False, lower=False):
    nl = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as fh:
        for i, s in enumerate(seqs):
            if lower:
                s = s.lower()
            fh.write(">s{}{}".format(i, nl))
            if width:
                for c in range(0, len(s), width):
                    fh.write(s[c:c + width] + nl)
            else:
                fh.write(s + nl)
