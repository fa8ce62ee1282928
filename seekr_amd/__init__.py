"""seekr_amd — MI355X-native k-mer counting + Pearson hot path with SEEKR's Python API.

    from seekr_amd.kmer_counts import BasicCounter
    from seekr_amd.pearson import pearson
    from seekr_amd.fasta_reader import Reader

Host code is Python + ctypes over `libseekr_hip.so` (hand-written HIP for gfx950, see
include/seekr_hip.h).  Build with `python -m seekr_amd.build`.
"""
__version__ = "0.1.0"
