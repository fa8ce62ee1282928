"""BASELINE configs 4 and 5 at their real per-rank sizes, in the driver-run suite (VERDICT r2 #1a).  The pool has one
GPU per box, so ONE rank of the 8-rank job runs at its full shard size through the production routines
(seekr_amd.distributed with the HIP engine) and what its peers would send is generated locally and delivered by RCCL
send/recv-to-self (tools/rehearsal.py).  Checked inside the tool against the oracle:

  * config 4 (200 000 x 2 kb, k = 6, ranks 0 and 7 of 8): the column mean of all 200 000 rows bit-equal to the C
    oracle's row-sequential float32 sums; the half ring's CROSS-mode blocks (25 000-row shard: own triangle + 3 1/2
    cross blocks, block and mirror) — sampled rows of every owned block inside |dr| <= 2e-6 + 1e-5 |r|, mirrors
    bit-equal to the transposes;
  * config 5 (1 000 000 x 5 kb, k = 7, rank 0 of 8): the 65.5 GB count matrix and the 65.5 GB gathered operand
    resident on the one GPU (131 of 288 GB), raw counts of the last rows bit-exact (byte offsets past 2^35), the
    rank's 15 stripes of 8 192 rows through the fused/striped edge extraction, edges of sampled rows equal to the
    oracle's.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rehearse(*argv, timeout):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearsal.py"), *argv], capture_output=True, text=True,
                         timeout=timeout)
    return out.returncode, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.parametrize("rank", [0, 7])
def test_config4_rank_at_full_shard_size(rank):
    code, text = _rehearse("cfg4", "--rank", str(rank), timeout=900)
    assert code == 0 and "rehearsal cfg4 ok rank=%d" % rank in text, text


def test_config5_rank0_at_full_shard_size():
    code, text = _rehearse("cfg5", "--rank", "0", timeout=1500)
    assert code == 0 and "rehearsal cfg5 ok rank=0" in text, text
