"""Packaging of seekr_amd.  The HIP library is built in-tree first: `python -m seekr_amd.build`."""
from setuptools import setup

setup(
    name="seekr_amd",
    version="0.1.0",
    description="MI355X-native k-mer counting + Pearson hot path with SEEKR's API",
    packages=["seekr_amd"],
    package_data={"seekr_amd": ["libseekr_hip.so", "csrc/*"]},
    install_requires=["numpy"],
    extras_require={"csv": ["pandas"], "progress": ["tqdm"]},
    entry_points={
        "console_scripts": [
            # same command names as the reference (setup.py:63-65)
            "seekr_kmer_counts = seekr_amd.console_scripts:console_kmer_counts",
            "seekr_pearson = seekr_amd.console_scripts:console_pearson",
            "seekr_norm_vectors = seekr_amd.console_scripts:console_norm_vectors",
        ]
    },
)
