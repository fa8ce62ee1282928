"""SEEKR_DEVICES behind the reference's own API and commands (VERDICT r4, row 8b'): BasicCounter.get_counts(), pearson(),
seekr_kmer_counts / seekr_norm_vectors / seekr_pearson with 2, 3 and 8 'devices' — ranks on the ONE GPU of the test box over
tests/mock_rccl (RCCL refuses two ranks on one GPU); tests/test_gpu_multirank_real.py runs the same comparison on real GPUs
wherever two are visible.  tests/multi_devices_worker.py is run once per setting, in its own process, and every file it
leaves must equal, BYTE FOR BYTE, what it leaves with SEEKR_DEVICES unset: counts, column mean / std, r (float32 split-fp16,
fp32, f16f8, float64, mixed, unstandardised, NaN rows), the CSV and .npy files of the commands, the warning text, and the
reference's exceptions (kmer_counts.py:144, fasta_reader.py:53) — raised once, the group alive afterwards.  The same holds
for one GPU with the stripe height of r forced small (r larger than the HBM takes that path: pearson.py:41 has no limit but
host RAM)."""
import filecmp
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def mock_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mock") / "libmock_rccl.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", out,
                    os.path.join(HERE, "mock_rccl", "mock_rccl.cpp")], check=True, capture_output=True)
    return out


def run_worker(out_dir, scale="full", **env):
    os.makedirs(str(out_dir), exist_ok=True)
    full = {k: v for k, v in os.environ.items() if k not in ("SEEKR_DEVICES", "SEEKR_PEARSON_STRIPE_ROWS", "SEEKR_PRECISION",
                                                             "SEEKR_TEST_HOOKS", "SEEKR_RCCL_LIB", "SEEKR_DEVICE")}
    full.update({k: str(v) for k, v in env.items()})
    proc = subprocess.run([sys.executable, os.path.join(HERE, "multi_devices_worker.py"), str(out_dir), scale], env=full,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert proc.returncode == 0, proc.stdout.decode()[-4000:]
    return str(out_dir)


@pytest.fixture(scope="module")
def baseline(tmp_path_factory):
    """SEEKR_DEVICES unset: the one-GPU path of rounds 1-4."""
    return run_worker(tmp_path_factory.mktemp("one_gpu"))


def same_outputs(a_dir, b_dir):
    a, b = np.load(os.path.join(a_dir, "results.npz")), np.load(os.path.join(b_dir, "results.npz"))
    assert sorted(a.files) == sorted(b.files)
    for name in a.files:
        x, y = a[name], b[name]
        assert x.dtype == y.dtype and x.shape == y.shape, name
        assert x.tobytes() == y.tobytes(), "{}: {} of {} cells differ".format(
            name, int((x.view(np.uint8) != y.view(np.uint8)).sum()) if x.dtype.kind == "f" else "?", x.size)
    files = sorted(f for f in os.listdir(a_dir) if f not in ("results.npz", "info.json"))
    assert files == sorted(f for f in os.listdir(b_dir) if f not in ("results.npz", "info.json")) and len(files) >= 9
    match, mismatch, errors = filecmp.cmpfiles(a_dir, b_dir, files, shallow=False)
    assert not mismatch and not errors, (mismatch, errors)


def test_the_baseline_is_what_the_reference_does(baseline):
    """The one-GPU run itself: exceptions as the reference raises them, the warning once per all-NaN pipeline, NaN where the
    reference has NaN (the parity of its numbers with the oracle is tests/test_gpu_parity.py's business)."""
    a = np.load(os.path.join(baseline, "results.npz"))
    assert int(a["zerodiv"][0]) == 1 and "division by zero" in str(a["zerodiv_text"])
    assert str(a["blank_text"]) == "string index out of range"
    # once per pipeline that ends with NaN: the 111 sequences at k = 6, and the 40 sequences over ACGTN (k-mers with N that never occur)
    assert str(a["stdout"]).count("WARNING: You have `np.nan` values") == 2 and np.isnan(a["acgtn_k3"]).any()
    assert np.isnan(a["g3_k6_nan"]).all() and np.isnan(a["g3_k6_r"]).all()
    assert np.isnan(a["g4_f64"][7]).all() and np.isnan(a["g4_f64"][:, 7]).all() and np.isfinite(a["g4_f64"][0, 1])
    assert a["big_r"].shape == (6000, 6000) and a["g3_mixed"].dtype == np.float64 and a["g3_r_k4"].dtype == np.float32
    assert np.array_equal(a["big_r"], a["big_r"].T) and np.abs(np.diag(a["big_r"]) - 1).max() < 1e-6
    r_file = np.load(os.path.join(baseline, "big_r_cross_file.npy"))
    assert r_file.shape == (6000, 777) and r_file.dtype == np.float32
    r64 = np.load(os.path.join(baseline, "cli_r64.npy"))
    assert r64.dtype == np.float64 and r64.shape == (111, 111) and np.array_equal(r64, r64.T) and np.allclose(np.diag(r64), 1.0, atol=1e-12)
    mixed = np.load(os.path.join(baseline, "mixed_r_file.npy"))
    assert mixed.dtype == np.float64 and mixed.shape == (900, 333)
    assert np.allclose(mixed, a["big_counts"][:900].astype(np.float64) @ a["big_counts"][:333].astype(np.float64).T / 4096, rtol=1e-12, atol=1e-12)
    assert np.allclose(r_file, a["big_r"][:, :777], rtol=1e-5, atol=2e-6)  # a cross comparison: same values, its own bits


def test_g10_files_that_are_not_ascii(baseline, golden_dir):
    """Golden set G10 through BasicCounter(infasta): the native parser declines a file with a byte >= 0x80 and the text-mode
    Reader gives what the reference's text-mode open gives — the reference's sequences, raw k = 2 counts bit for bit, and its
    exceptions (UnicodeDecodeError included) with their text.  The same cases run under every SEEKR_DEVICES setting of this
    file (same_outputs compares them byte for byte with this run)."""
    import locale
    with open(os.path.join(golden_dir, "g10_non_ascii.json")) as fh:
        g = json.load(fh)
    if locale.getpreferredencoding(False).lower().replace("-", "") != g["text_encoding"].lower().replace("-", ""):
        pytest.skip("text-mode open() decodes with another codec here")
    a = np.load(os.path.join(baseline, "results.npz"))
    n_counts = 0
    for case in g["cases"]:
        got = a["g10_" + case["name"]]
        if "exception" in case:
            assert str(got) == case["exception"] + ": " + case["message"], case["name"]
            continue
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), np.array(case["raw_k2_bits"], dtype=np.uint32)), case["name"]
        assert str(a["g10_seqs_" + case["name"]]).split("\n") == case["seqs"], case["name"]
        n_counts += 1
    assert n_counts >= 13
    # the same through seekr_kmer_counts: the .npy holds the reference's raw bits of the mixed file, the labelled csv its
    # headers (decoded text, as the reference's DataFrame index would hold them) in the first column
    mixed = [c for c in g["cases"] if c["name"] == "mixed_records"][0]
    raw = np.load(os.path.join(baseline, "cli_g10_raw.npy"))
    assert raw.dtype == np.float32 and np.array_equal(raw.view(np.uint32), np.array(mixed["raw_k2_bits"], dtype=np.uint32))
    named = [c for c in g["cases"] if c["name"] == "non_ascii_only_in_headers"][0]
    with open(os.path.join(baseline, "cli_g10_labelled.csv"), encoding="utf-8") as fh:
        rows = fh.read().splitlines()
    assert [r.split(",")[0] for r in rows[1:]] == named["headers"] and any(ord(ch) > 127 for ch in "".join(named["headers"]))


@pytest.mark.parametrize("stripe", [300, 1])
def test_one_gpu_with_r_in_stripes(baseline, tmp_path, stripe):
    """pearson() by row stripes on ONE GPU (what a result larger than the HBM gets), stripe height forced: same bytes.
    Height 1 only at the small scale (6 000 launches per call otherwise)."""
    if stripe == 1:
        small = run_worker(tmp_path / "small_ref", scale="small")
        same_outputs(small, run_worker(tmp_path / "small_striped", scale="small", SEEKR_PEARSON_STRIPE_ROWS=37))
        return
    same_outputs(baseline, run_worker(tmp_path / "striped", SEEKR_PEARSON_STRIPE_ROWS=stripe))


@pytest.mark.parametrize("size,stripe", [(2, None), (3, 256), (8, None)])
def test_ranks_on_one_gpu_behind_the_api(baseline, mock_lib, tmp_path, size, stripe):
    env = dict(SEEKR_DEVICES=",".join(["0"] * size), SEEKR_TEST_HOOKS=1, SEEKR_RCCL_LIB=mock_lib, MOCK_RCCL_ASYNC=0)
    if stripe:
        env["SEEKR_PEARSON_STRIPE_ROWS"] = stripe
    got = run_worker(tmp_path / ("ranks%d" % size), **env)
    with open(os.path.join(got, "info.json")) as fh:
        info = json.load(fh)
    # the run really was `size` GPU threads that lived through every call (the ZeroDivisionError included)
    assert info["devices"] == [0] * size and info["group_size"] == size and info["group_broken"] is False
    assert len(info["threads"]) == size and info["stripe_rows"] == stripe
    # the transport recorded is the one that carried the data, and its set-up all-reduce (ncclAllReduce on the devices) counted
    # every rank; SEEKR_DEVICES stood in the environment at import, so the runtime started with dmabuf IPC
    assert info["transport"] == "rccl" and info["n_ranks_seen"] == size and info["ipc_env_at_load"] == "0", info
    with open(os.path.join(baseline, "info.json")) as fh:
        assert json.load(fh)["group_size"] == 0
    same_outputs(baseline, got)


@pytest.mark.parametrize("size,stripe", [(2, 256), (3, None), (8, None)])
def test_gpu_threads_over_peer_copies(baseline, tmp_path, size, stripe):
    """SEEKR_TRANSPORT=peer: the in-process group WITHOUT RCCL — every transfer a peer copy pulled by the receiver on its own
    communication stream, ordered by events (skr_event_*, skr_peer_copy_rows).  No mock here: this is the production
    transport, its 'peers' all being GPU 0 (a same-device hipMemcpyPeerAsync); what a multi-GPU box adds is xGMI under
    the same calls.  Every output byte-identical to the one-GPU run."""
    env = dict(SEEKR_DEVICES=",".join(["0"] * size), SEEKR_TEST_HOOKS=1, SEEKR_TRANSPORT="peer")
    if stripe:
        env["SEEKR_PEARSON_STRIPE_ROWS"] = stripe
    got = run_worker(tmp_path / ("peer%d" % size), **env)
    with open(os.path.join(got, "info.json")) as fh:
        info = json.load(fh)
    assert info["group_size"] == size and info["group_broken"] is False and info["transport"] == "peer"
    assert info["n_ranks_seen"] == size
    same_outputs(baseline, got)


def test_rccl_that_cannot_be_set_up_falls_back_to_peer_copies(baseline, tmp_path):
    """SEEKR_TRANSPORT unset (auto): RCCL first — here its library cannot be loaded — then peer copies, with one line on
    stderr; the results are the same bytes."""
    got = run_worker(tmp_path / "fallback", scale="small", SEEKR_DEVICES="0,0", SEEKR_TEST_HOOKS=1,
                     SEEKR_RCCL_LIB=str(tmp_path / "no_such_librccl.so"))
    with open(os.path.join(got, "info.json")) as fh:
        info = json.load(fh)
    assert info["transport"] == "peer" and info["group_size"] == 2 and info["group_broken"] is False
    same_outputs(run_worker(tmp_path / "small_ref", scale="small"), got)


def test_bench_reports_the_host_to_host_figures_with_seekr_devices(tmp_path):
    """bench.py's `e2e.seekr_devices_all`: the FASTA -> host counts and host -> host pearson() figures
    again with every visible GPU behind the API, from a child process.  On a one-GPU box the sub-record's code path runs
    through a test hook that names GPU 0 twice (peer copies)."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, SEEKR_TEST_HOOKS="1", SEEKR_BENCH_E2E_DEVICES="0,0", SEEKR_TRANSPORT="peer")
    env.pop("SEEKR_DEVICES", None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", "6000", "--length", "600", "--steps", "2", "--warmup", "1",
                          "--no-f16f8-arm"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    dev = out["e2e"]["seekr_devices_all"]
    assert dev.get("devices") == 2 and "error" not in dev, dev
    assert dev["transport"] == "peer" and dev["n_ranks_seen"] == 2 and dev["group_size"] == 2, dev  # what was used, not what was asked
    assert dev["host_to_host_pearson_mpairs_per_s"] > 0 and dev["fasta_to_host_counts_mbases_per_s"] > 0 and dev["first_call_s"] > 0
    assert "target_200k" not in out  # only the default workload carries it
