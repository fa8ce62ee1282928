"""seekr_amd — MI355X-native k-mer counting + Pearson hot path with SEEKR's Python API.

    from seekr_amd.kmer_counts import BasicCounter
    from seekr_amd.pearson import pearson
    from seekr_amd.fasta_reader import Reader

Host code is Python + ctypes over `libseekr_hip.so` (hand-written HIP for gfx950, see
include/seekr_hip.h).  Build with `python -m seekr_amd.build`.
"""
__version__ = "0.1.0"

# SEEKR_DEVICES naming several GPUs: the one environment variable RCCL's set-up between them needs must be in place before
# the first HIP call of the process (seekr_amd._lib.prepare_runtime_env says why) — importing the package is the earliest
# moment the package has.
from seekr_amd._lib import prepare_runtime_env as _prepare_runtime_env  # noqa: E402

_prepare_runtime_env()
