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
Generates a kmer count matrix of m rows by n columns,
where m is the number of transcripts in a fasta file and n is 4^kmer.
(MI355X implementation; same flags and outputs as `seekr_kmer_counts`.)

Examples
--------
    $ seekr_kmer_counts rnas.fa -o out.csv
    $ seekr_kmer_counts rnas.fa -o out.npy -k 5 -b -rl
    $ seekr_kmer_counts rnas.fa -o out.csv -mv mean.npy -sv std.npy
"""

PEARSON_DOC = """
Description
-----------
Generate a matrix of Pearson similarities from two kmer count files.
(MI355X implementation; same flags and outputs as `seekr_pearson`.)

Examples
--------
    $ seekr_pearson kc_out.csv kc_out.csv -o out.csv
    $ seekr_pearson kc_out.npy kc_out.npy -o out.npy -bi -bo
"""

NORM_VECTORS_DOC = """
Description
-----------
Generate two .npy files from a .fa file to use as normalization vectors for other .fa files.
(MI355X implementation; same flags and outputs as `seekr_norm_vectors`.)

Examples
--------
    $ seekr_norm_vectors gencode.fa
    $ seekr_norm_vectors gencode.fa -k 5 -mv mean_5mers.npy -sv std_5mers.npy
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
    parser.add_argument("fasta", help="Full path of fasta file.")
    parser.add_argument("-o", "--outfile", default="counts.seekr", help="Name of file to save counts to.")
    parser.add_argument("-k", "--kmer", default=6, help="Length of kmers you want to count.")
    parser.add_argument("-b", "--binary", action="store_true", help="Set if output should be a .npy file.")
    parser.add_argument("-uc", "--uncentered", action="store_false",
                        help="Set if output should not have the mean subtracted.")
    parser.add_argument("-us", "--unstandardized", action="store_false",
                        help="Set if output should not be divided by the standard deviation.")
    parser.add_argument("-l", "--log2", default="Log2.post", choices=_LOG2,
                        help="Decided if and when to log transform counts")
    parser.add_argument("-rl", "--remove_labels", action="store_true",
                        help="Set to save without index and column labels.")
    parser.add_argument("-mv", "--mean_vector", default=None, help="Optional path to mean vector numpy file.")
    parser.add_argument("-sv", "--std_vector", default=None, help="Optional path to std vector numpy file.")
    parser.add_argument("-a", "--alphabet", default="AGTC", help="Valid letters to include in kmer.")
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
    if binary_input:
        counts1 = np.load(counts1)
        counts2 = np.load(counts2)
    else:  # labelled CSVs; float64 path (console_scripts.py:628-631)
        same_file = counts1 == counts2
        counts1, names1 = _read_labelled_csv(counts1)
        counts2, names2 = (counts1, names1) if same_file else _read_labelled_csv(counts2)
    if binary_output:
        pearson_mod.pearson(counts1, counts2, outfile=outfile)
    else:
        dist = pearson_mod.pearson(counts1, counts2)
        # pd.DataFrame(dist, names1, names2).to_csv(outfile); names None -> RangeIndex labels 0..n-1
        _lib.save_csv_labelled(outfile, dist, range(dist.shape[0]) if names1 is None else names1,
                               range(dist.shape[1]) if names2 is None else names2)


def console_pearson():
    parser = argparse.ArgumentParser(usage=PEARSON_DOC, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("counts1", help="Full path of a count file produced by kmer_counts.py.")
    parser.add_argument("counts2", help=("Full path of a second count file produced by kmer_counts.py. "
                                         "This can be the same path as the first counts file."))
    parser.add_argument("-o", "--outfile", default="pearson.seekr", help="Path of file to save similarities to.")
    parser.add_argument("-bi", "--binary_input", action="store_true", help="Set if the input will be a .npy file.")
    parser.add_argument("-bo", "--binary_output", action="store_true", help="Set if output should be a .npy file.")
    args = _parse_args_or_exit(parser)
    _run_pearson(args.counts1, args.counts2, args.outfile, args.binary_input, args.binary_output)


def _run_norm_vectors(fasta, mean_vector, std_vector, log2, kmer):
    counter = BasicCounter(fasta, k=int(kmer), log2=log2)
    counter.get_counts()
    _lib.save_npy(mean_vector, counter.mean)
    _lib.save_npy(std_vector, counter.std)


def console_norm_vectors():
    parser = argparse.ArgumentParser(usage=NORM_VECTORS_DOC, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("fasta", help="path to .fa file")
    parser.add_argument("-mv", "--mean_vector", default="mean.npy", help="path to output mean vector")
    parser.add_argument("-sv", "--std_vector", default="std.npy", help="path to output standard deviation vector")
    parser.add_argument("-l", "--log2", default="Log2.post", choices=_LOG2,
                        help="Decided if and when to log transform counts")
    parser.add_argument("-k", "--kmer", default=6, help="length of kmers you want to count")
    args = _parse_args_or_exit(parser)
    _run_norm_vectors(args.fasta, args.mean_vector, args.std_vector, args.log2, int(args.kmer))
