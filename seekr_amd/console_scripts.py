"""`seekr_kmer_counts`, `seekr_pearson`, `seekr_norm_vectors` with the reference's flags
(console_scripts.py:564-681).  Only the three commands on the hot path are provided."""
import argparse
import sys

import numpy as np

from seekr_amd import _lib
from seekr_amd import pearson as pearson_mod
from seekr_amd.kmer_counts import BasicCounter

KMER_COUNTS_DOC = """
Description
-----------
Counts the k-mers of every sequence of a FASTA file on an MI355X and writes the per-kb,
normalised count matrix: one row per sequence, 4^k columns.  Flags and output files are
those of the reference command of the same name.

Examples
--------
    labelled CSV, 6-mers:            seekr_kmer_counts transcripts.fa -o counts.csv
    binary, unlabelled, 5-mers:      seekr_kmer_counts transcripts.fa -o counts.npy -k 5 -b -rl
    normalise with stored vectors:   seekr_kmer_counts transcripts.fa -o counts.csv -mv mean.npy -sv std.npy
"""

PEARSON_DOC = """
Description
-----------
All-pairs Pearson correlation between the rows of two k-mer count files, computed on an
MI355X.  Flags and output files are those of the reference command of the same name.

Examples
--------
    labelled CSV in and out:   seekr_pearson counts.csv counts.csv -o r.csv
    .npy in and out:           seekr_pearson counts.npy counts.npy -o r.npy -bi -bo
"""

NORM_VECTORS_DOC = """
Description
-----------
Column mean and standard deviation of the k-mer counts of a (large) FASTA file, saved as two .npy
vectors that later runs can normalise against (-mv / -sv of seekr_kmer_counts).

Examples
--------
    defaults (6-mers, mean.npy, std.npy):   seekr_norm_vectors gencode.fa
    5-mers, named outputs:                  seekr_norm_vectors gencode.fa -k 5 -mv mean5.npy -sv std5.npy
"""

_LOG2 = ["Log2.post", "Log2.pre", "Log2.none"]


def _parse_args_or_exit(parser):
    if len(sys.argv) == 1:  # console_scripts.py:520-525
        parser.print_help()
        sys.exit(0)
    return parser.parse_args()


def _run_kmer_counts(fasta, outfile, kmer, binary, centered, standardized, log2, remove_labels, mean_vector,
                     std_vector, alphabet):
    mean = mean_vector or centered  # console_scripts.py:568-569
    std = std_vector or standardized
    counter = BasicCounter(fasta, outfile, kmer, binary, mean, std, log2, label=not remove_labels, alphabet=alphabet)
    counter.make_count_file()


def console_kmer_counts():
    parser = argparse.ArgumentParser(usage=KMER_COUNTS_DOC, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("fasta", help="FASTA file with the sequences to count.")
    parser.add_argument("-o", "--outfile", default="counts.seekr", help="Where the count matrix goes.")
    parser.add_argument("-k", "--kmer", default=6, help="k, the word length (4^k columns).")
    parser.add_argument("-b", "--binary", action="store_true", help="Write .npy instead of CSV.")
    parser.add_argument("-uc", "--uncentered", action="store_false",
                        help="Leave the column means in (no centring).")
    parser.add_argument("-us", "--unstandardized", action="store_false",
                        help="Do not divide by the column standard deviations.")
    parser.add_argument("-l", "--log2", default="Log2.post", choices=_LOG2,
                        help="log2 before the column statistics (pre), after them (post), or not at all (none).")
    parser.add_argument("-rl", "--remove_labels", action="store_true",
                        help="Plain CSV without the header row and the name column (required with -b).")
    parser.add_argument("-mv", "--mean_vector", default=None, help="Centre with this stored mean vector (.npy) instead of the set's own.")
    parser.add_argument("-sv", "--std_vector", default=None, help="Scale with this stored std vector (.npy) instead of the set's own.")
    parser.add_argument("-a", "--alphabet", default="AGTC", help="The four letters, in column-index order.")
    args = _parse_args_or_exit(parser)
    _run_kmer_counts(args.fasta, args.outfile, int(args.kmer), args.binary, args.uncentered, args.unstandardized,
                     args.log2, args.remove_labels, args.mean_vector, args.std_vector, args.alphabet)


def _read_labelled_csv(path):
    """pd.read_csv(path, index_col=0) -> (float64 values, index labels): natively for the files the
    count command writes, through pandas for anything the native reader declines (csv_read.hip)."""
    native = _lib.load_csv_labelled(path)
    if native is not None:
        return native[0], np.array(native[1], dtype=object)
    import pandas as pd
    frame = pd.read_csv(path, index_col=0)
    return frame, frame.index.values


def _run_pearson(counts1, counts2, outfile, binary_input, binary_output):
    names1 = names2 = None
    same_file = counts1 == counts2  # one file twice: read once, and the contraction computes one triangle
    if binary_input:
        counts1 = np.load(counts1)
        counts2 = counts1 if same_file else np.load(counts2)
    else:  # labelled CSVs; float64 path (console_scripts.py:628-631)
        counts1, names1 = _read_labelled_csv(counts1)
        counts2, names2 = (counts1, names1) if same_file else _read_labelled_csv(counts2)
    if binary_output:
        # np.save(outfile, dist) without dist ever standing in host memory: stripes of r go from the GPU(s) to the file
        pearson_mod.pearson_to_file(counts1, counts2, outfile)
    else:
        dist = pearson_mod.pearson(counts1, counts2)
        # pd.DataFrame(dist, names1, names2).to_csv(outfile); names None -> RangeIndex labels 0..n-1
        _lib.save_csv_labelled(outfile, dist, range(dist.shape[0]) if names1 is None else names1,
                               range(dist.shape[1]) if names2 is None else names2)


def console_pearson():
    parser = argparse.ArgumentParser(usage=PEARSON_DOC, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("counts1", help="First count file (rows of the result).")
    parser.add_argument("counts2", help="Second count file (columns of the result); may be the first one again.")
    parser.add_argument("-o", "--outfile", default="pearson.seekr", help="Where the correlation matrix goes.")
    parser.add_argument("-bi", "--binary_input", action="store_true", help="The count files are .npy, not labelled CSV.")
    parser.add_argument("-bo", "--binary_output", action="store_true", help="Write .npy instead of CSV.")
    args = _parse_args_or_exit(parser)
    _run_pearson(args.counts1, args.counts2, args.outfile, args.binary_input, args.binary_output)


def _run_norm_vectors(fasta, mean_vector, std_vector, log2, kmer):
    counter = BasicCounter(fasta, k=int(kmer), log2=log2)
    counter.get_counts()
    _lib.save_npy(mean_vector, counter.mean)
    _lib.save_npy(std_vector, counter.std)


def console_norm_vectors():
    parser = argparse.ArgumentParser(usage=NORM_VECTORS_DOC, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("fasta", help="FASTA file to take the statistics of.")
    parser.add_argument("-mv", "--mean_vector", default="mean.npy", help="Output: column means (.npy).")
    parser.add_argument("-sv", "--std_vector", default="std.npy", help="Output: column standard deviations (.npy).")
    parser.add_argument("-l", "--log2", default="Log2.post", choices=_LOG2,
                        help="log2 before the column statistics (pre), after them (post), or not at all (none).")
    parser.add_argument("-k", "--kmer", default=6, help="k, the word length.")
    args = _parse_args_or_exit(parser)
    _run_norm_vectors(args.fasta, args.mean_vector, args.std_vector, args.log2, int(args.kmer))
