"""The N > 1 path over the REAL librccl, one GPU per rank.  The pool's boxes have one GPU, so these
tests skip there; on any box with two or more GPUs they are the first thing to run: launch.init's rendezvous and
self-test, the rank-chained float32 column sums, the NaN-min all-reduce, both Pearson layouts (symmetric half ring with
its split first shift, row blocks), the striped edge lists — same worker and same assertions as the mock-transport test
(tests/test_gpu_multirank_mock.py::check_ranks): statistics and normalised counts bit-identical to the single-GPU run, r
and edges equal to it — and bench.py starting its own rank processes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _gpus():
    try:
        from seekr_amd import _lib
        return _lib.device_count()
    except Exception:  # noqa: BLE001 - library not built yet: nothing to run here
        return 0


needs2 = pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (the pool's boxes have one)")
needs4 = pytest.mark.skipif(_gpus() < 4, reason="needs four GPUs")


@needs2
def test_two_ranks_over_real_rccl_equal_single_gpu(tmp_path):
    from test_gpu_multirank_mock import check_ranks, single_gpu_reference
    check_ranks(2, 0, None, single_gpu_reference(1101, 600, 6), tmp_path)


@needs4
@pytest.mark.parametrize("size", [3, 4])
def test_more_ranks_over_real_rccl_equal_single_gpu(size, tmp_path):
    from test_gpu_multirank_mock import check_ranks, single_gpu_reference
    check_ranks(size, 0, None, single_gpu_reference(1101, 600, 6), tmp_path)


@needs2
def test_mailbox_chain_between_two_real_gpus(tmp_path):
    """the peer-mailbox column-sum chain WITHOUT the host-side wait — rank 1's kernel is resident on its
    own GPU and polls, inside the kernel, the epoch words rank 0's kernel stores over xGMI into rank 1's uncached mailbox.
    This is the first place that wait runs against a live peer; statistics must equal the single-GPU run bit for bit
    and both ranks must report the mailbox transport (a failed set-up or self-test would say why instead)."""
    import numpy as np
    from test_gpu_multirank_mock import check_ranks, single_gpu_reference
    check_ranks(2, 0, None, single_gpu_reference(1101, 600, 6), tmp_path, extra_env={"SEEKR_CHAIN": "mailbox"})
    notes = [str(np.load(str(tmp_path / ("rank%d.npz" % rank)))["chain_note"]) for rank in range(2)]
    assert all("peer mailboxes" in n for n in notes), notes


@needs4
def test_mailbox_chain_between_four_real_gpus(tmp_path):
    import numpy as np
    from test_gpu_multirank_mock import check_ranks, single_gpu_reference
    check_ranks(4, 0, None, single_gpu_reference(1101, 600, 6), tmp_path, extra_env={"SEEKR_CHAIN": "mailbox"})
    notes = [str(np.load(str(tmp_path / ("rank%d.npz" % rank)))["chain_note"]) for rank in range(4)]
    assert all("peer mailboxes" in n for n in notes), notes


@needs2
def test_bench_reports_the_chain_ab_between_two_real_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR",
                                                               "SEEKR_TEST_HOOKS", "SEEKR_RCCL_LIB", "SEEKR_FORCE_DEVICE", "SEEKR_CHAIN")}
    env["SEEKR_BENCH_CHAIN_AB"] = "1"  # opt-in since round 5: a hang in the A/B must not cost the headline line
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "12000",
           "--length", "500"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    ab = out["chain_ab"]  # the launcher's first attempt measures both transports after the timed region
    assert ab["bit_identical"] is True and ab["rccl"]["ms_per_pass"] > 0 and ab["mailbox"]["ms_per_pass"] > 0
    assert "peer mailboxes" in ab["mailbox"]["transport"] and not ab["mailbox"]["a_link_gave_up_waiting"], ab


@needs2
def test_k7_over_real_rccl(tmp_path):
    from test_gpu_multirank_mock import check_ranks, single_gpu_reference
    check_ranks(2, 0, None, single_gpu_reference(520, 900, 7), tmp_path)


@needs2
@pytest.mark.parametrize("extra", [[], ["--grouped-shifts"], ["--layout", "rowblock"], ["--layout", "allgather"]])
def test_bench_starts_two_real_ranks(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR",
                                                               "SEEKR_TEST_HOOKS", "SEEKR_RCCL_LIB", "SEEKR_FORCE_DEVICE")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "12000",
           "--length", "500"] + extra
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["value"] > 0 and "layout_fallback" not in out
    assert len(out["per_rank"]["comm_ms"]) == 2


@needs2
@pytest.mark.parametrize("devices,stripe,transport", [("all", None, None), ("0,1", 256, None), ("all", None, "peer"), ("0,1", 256, "rccl")])
def test_seekr_devices_behind_the_api_on_real_gpus(devices, stripe, transport, tmp_path):
    """SEEKR_DEVICES over the REAL librccl — one host thread per GPU, ncclCommInitRank per thread — through BasicCounter,
    pearson() and the three commands: every output byte-identical to the run with SEEKR_DEVICES unset (the comparison of
    tests/test_gpu_multi_devices.py, which has only mock ranks on one GPU to offer on the pool's boxes).  With
    SEEKR_TRANSPORT unset the data MUST have travelled over RCCL (the 'auto' arm falls back to peer
    copies with one stderr line, which this test would otherwise never notice): the transport recorded is the one used,
    its set-up all-reduce counted every rank, and HSA_ENABLE_IPC_MODE_LEGACY=0 stood in the environment when the HIP
    runtime started although this test removes it from the child's (the package sets it at import).  The
    SEEKR_TRANSPORT=peer twin runs the same comparison over peer copies."""
    from test_gpu_multi_devices import run_worker, same_outputs
    baseline = run_worker(tmp_path / "one_gpu")
    env = {"SEEKR_DEVICES": devices}
    if stripe:
        env["SEEKR_PEARSON_STRIPE_ROWS"] = stripe
    if transport:
        env["SEEKR_TRANSPORT"] = transport
    saved = os.environ.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # the child's environment is copied from this one
    try:
        got = run_worker(tmp_path / "devices", **env)
    finally:
        if saved is not None:
            os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = saved
    with open(os.path.join(got, "info.json")) as fh:
        info = json.load(fh)
    size = _gpus() if devices == "all" else 2
    assert info["group_size"] == size and info["group_broken"] is False
    assert info["transport"] == (transport or "rccl"), info
    assert info["n_ranks_seen"] == size and info["ipc_env_at_load"] == "0", info
    same_outputs(baseline, got)
