"""The N > 1 path for real on ONE GPU: 2, 3, 4 and 8 rank processes share device 0 through
tests/mock_rccl (RCCL itself refuses two ranks on one GPU), running launch.init, the C-ABI comm
layer with its streams and tickets, and the HIP engine under the sharded orchestration.  Results
must equal the single-GPU pipeline: statistics and normalised counts bit for bit (the float32 sum
chain crosses ranks in row order), r in both layouts, the striped edge list."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def mock_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mock") / "libmock_rccl.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", out,
                    os.path.join(HERE, "mock_rccl", "mock_rccl.cpp")], check=True, capture_output=True)
    return out


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def single_gpu_reference(n_total, length, k):
    """The same workload on one GPU through the same entry points."""
    from seekr_amd import _lib
    from seekr_amd.distributed import HipEngine, SingleComm, sharded_normalize_prepare
    from seekr_amd.synthetic import synthetic_ascii
    ctx = _lib.default_context()
    blob, offsets = synthetic_ascii(11, n_total, length)
    x = _lib.count_per_kb(ctx, _lib.PackedSeqs.from_buffer(ctx, blob, offsets, "AGTC"), k)
    engine = HipEngine(ctx)
    mean, std, has_nan, z = sharded_normalize_prepare(engine, SingleComm(), x, n_total, "Log2.post", True, True)
    r = ctx.empty(n_total, n_total)
    _lib.pearson_gemm_op(ctx, z, z, r, symmetric=True)
    return dict(n_total=n_total, length=length, k=k, mean=mean.vector(), std=std.vector(), x=x.to_numpy(), r=r.to_numpy())


@pytest.fixture(scope="module")
def single():
    return single_gpu_reference(1101, 600, 6)


@pytest.fixture(scope="module")
def single_k7():
    return single_gpu_reference(520, 900, 7)


def test_k7_ranks_on_one_gpu(mock_lib, single_k7, tmp_path):
    """k = 7: 16 384 columns, i.e. the block-per-row operand fill and four accumulator restarts per
    contraction (later chunks add into C), in SELF, CROSS and PLAIN mode."""
    check_ranks(3, 1, mock_lib, single_k7, tmp_path)


@pytest.mark.parametrize("size,asynchronous", [(2, 0), (3, 0), (4, 0), (8, 0), (2, 1), (3, 1), (4, 1)])
def test_ranks_on_one_gpu_equal_single_gpu(size, asynchronous, mock_lib, single, tmp_path):
    check_ranks(size, asynchronous, mock_lib, single, tmp_path)


@pytest.mark.parametrize("size", [4, 5, 8])
def test_all_shifts_posted_as_one_group(size, mock_lib, single, tmp_path):
    """sharded_pearson_symmetric(grouped=True): every shift of the half ring in ONE ncclGroup (skr_comm_exchange), one
    receive buffer per shift — all xGMI links at once on real hardware; here the asynchronous mock checks the data."""
    check_ranks(size, 1, mock_lib, single, tmp_path, extra_env={"MOCK_GROUPED_SHIFTS": "1"})


def check_ranks(size, asynchronous, mock_lib, single, tmp_path, extra_env=None):
    """asynchronous = 1: the mock enqueues its copies and waits on the communication stream like RCCL's
    kernels, so the product's event / ticket waits between the two streams are what keeps the data right.
    mock_lib = None: the real librccl, one GPU per rank (tests/test_gpu_multirank_real.py)."""
    env = dict(os.environ, WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    if mock_lib is not None:
        env.update(LOCAL_RANK="0", SEEKR_TEST_HOOKS="1", SEEKR_RCCL_LIB=mock_lib, MOCK_RCCL_ASYNC=str(asynchronous))
    else:
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for key in ("SEEKR_TEST_HOOKS", "SEEKR_RCCL_LIB", "SEEKR_FORCE_DEVICE"):
            env.pop(key, None)
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mock_rank_worker.py"), str(tmp_path),
                               str(single["n_total"]), str(single["length"]), str(single["k"])],
                              env=dict(env, RANK=str(rank), **({} if mock_lib is not None else {"LOCAL_RANK": str(rank)})),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for rank in range(size)]
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out.decode()[-3000:])
    parts = [np.load(str(tmp_path / ("rank%d.npz" % rank))) for rank in range(size)]
    n = single["n_total"]
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)  # noqa: E731
    for p in parts:
        lo, hi = int(p["lo"]), int(p["hi"])
        assert np.array_equal(bits(p["mean"]), bits(single["mean"]))   # the chain crossed ranks in row order
        assert np.array_equal(bits(p["std"]), bits(single["std"]))
        assert np.array_equal(bits(p["x"]), bits(single["x"][lo:hi]))
        assert not bool(p["has_nan"])
        # the row block is the one-GPU result BIT FOR BIT: shards that come before the rank's own are multiplied with the
        # cross products in their mirror's order (skr_pearson_gemm_op, symmetric = 2; round 5)
        assert np.array_equal(bits(p["r"]), bits(single["r"][lo:hi]))
        assert np.array_equal(bits(p["r_ag"]), bits(p["r"]))  # the all-gather schedule: the row-block result bit for bit
    # symmetric layout: every cell on exactly one rank, exactly symmetric, same values as one GPU
    full, hits = np.zeros((n, n), np.float32), np.zeros((n, n), np.int32)
    for p in parts:
        for which, br, bc, nr, nc, gr, gc in p["blocks"]:
            buf = p["r_row"] if which == 0 else p["r_col"]
            full[gr:gr + nr, gc:gc + nc] = buf[br:br + nr, bc:bc + nc]
            hits[gr:gr + nr, gc:gc + nc] += 1
    assert (hits == 1).all()
    assert np.array_equal(bits(full), bits(full.T.copy()))
    # ... and the one-GPU result bit for bit: in every cross block the rows that come first in the matrix take the A side
    assert np.array_equal(bits(full), bits(single["r"]))
    # striped edge lists: the union over the ranks = upper triangle of the thresholded single-GPU matrix
    want = np.triu(np.where(single["r"] < 0.05, 0, single["r"]), 1)
    got = np.zeros_like(want)
    for p in parts:
        assert not got[p["e_i"], p["e_j"]].any()
        got[p["e_i"], p["e_j"]] = p["e_v"]
    clear = np.abs(single["r"] - 0.05) > 1e-5
    assert np.array_equal(got[clear] != 0, want[clear] != 0) and np.allclose(got[clear], want[clear], rtol=1e-6, atol=1e-6)
    assert np.count_nonzero(got[clear]) == np.count_nonzero(want[clear])
    if single["k"] == 6:
        assert np.count_nonzero(got) > 100


def test_bench_under_torchrun_with_two_ranks(mock_lib):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per
    rank), both ranks on device 0 over the mock transport: one JSON line from rank 0, whole-job value."""
    import json
    env = dict(os.environ, SEEKR_TEST_HOOKS="1", SEEKR_RCCL_LIB=mock_lib, SEEKR_FORCE_DEVICE="0", MOCK_RCCL_ASYNC="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--rows", "6000", "--length", "500"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong" and out["value"] > 0  # --rows fixes the set: strong (the default sizing says weak)
    assert out["config"]["rows_total"] == 6000 and out["config"]["rows_per_gpu"] == 3000
    assert "symmetric" in out["config"]["layout"] and "cpu_baseline" not in out
    # half the ordered pairs are multiplied: own triangle + half of the one cross block
    assert abs(out["roofline"]["pairs_multiplied_per_step"] / out["roofline"]["pairs_delivered_per_step"] - 0.5) < 0.1


def test_column_sum_chain_through_peer_mailboxes(mock_lib, single, tmp_path):
    """SEEKR_CHAIN=mailbox with two rank processes on one GPU: each rank's mailbox (uncached device memory) is exported as a
    HIP IPC handle and opened by the other process for real; the column-sum kernel of rank 1 waits inside the kernel for
    rank 0's running sums and the finished sums come back through the result box — no send/recv between two kernels.
    Statistics and normalised counts must equal the single-GPU run bit for bit, as over send/recv.
    (SEEKR_CHAIN_HOST_WAIT=1: on a shared GPU the wait for the mailbox is done by the host before the launch — a waiting
    kernel fills every CU and the kernel it waits for would never start; between GPUs the wait is inside the kernel.)"""
    check_ranks(2, 1, mock_lib, single, tmp_path, extra_env={"SEEKR_CHAIN": "mailbox", "SEEKR_CHAIN_HOST_WAIT": "1"})
    notes = [str(np.load(str(tmp_path / ("rank%d.npz" % rank)))["chain_note"]) for rank in range(2)]
    assert all("peer mailboxes" in n for n in notes), notes


@pytest.mark.parametrize("size", [3, 4])
def test_column_sum_chain_through_peer_mailboxes_more_ranks(size, mock_lib, single, tmp_path):
    check_ranks(size, 1, mock_lib, single, tmp_path, extra_env={"SEEKR_CHAIN": "mailbox", "SEEKR_CHAIN_HOST_WAIT": "1"})
    notes = [str(np.load(str(tmp_path / ("rank%d.npz" % rank)))["chain_note"]) for rank in range(size)]
    assert all("peer mailboxes" in n for n in notes), notes


def test_column_sum_chain_with_empty_shards(mock_lib, tmp_path):
    """Fewer rows than ranks: a rank without rows still passes the running sums on (chain_forward_kernel)."""
    check_ranks(4, 1, mock_lib, single_gpu_reference(3, 400, 2), tmp_path, extra_env={"SEEKR_CHAIN": "mailbox", "SEEKR_CHAIN_HOST_WAIT": "1"})


def test_column_sum_chain_falls_back_to_send_recv(mock_lib, single, tmp_path):
    """Without the opt-in the ranks of a shared GPU keep the send/recv chain (a kernel waiting for another process's
    kernel on the SAME GPU may keep it from being scheduled); the note says so."""
    check_ranks(3, 1, mock_lib, single, tmp_path)
    notes = [str(np.load(str(tmp_path / ("rank%d.npz" % rank)))["chain_note"]) for rank in range(3)]
    assert all(n == "not requested" for n in notes), notes


def _launcher_env(mock_lib, **extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(SEEKR_TEST_HOOKS="1", SEEKR_RCCL_LIB=mock_lib, SEEKR_FORCE_DEVICE="0", MOCK_RCCL_ASYNC="1", **extra)
    return env


def test_bench_starts_its_own_rank_processes(mock_lib):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the shape of the driver's N = 1 command, VERDICT r2
    #2): the launcher spawns the two ranks itself, the half-ring self-test runs before the warm-up, and rank 0's line
    carries n_ranks_seen and the per-rank transfer / exposed-wait times."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "6000",
           "--length", "500"]
    res = subprocess.run(cmd, env=_launcher_env(mock_lib), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and "symmetric" in out["config"]["layout"]
    assert "layout_fallback" not in out and out["selftest"].startswith("half-ring schedule == row-block")
    assert out["verified"] is True and out["verified_detail"]["worst_error_over_bar"] <= 1.0
    pr = out["per_rank"]
    for key in ("comm_ms", "exposed_wait_ms", "chain_wait_ms", "gemm_ms"):
        assert len(pr[key]) == 2 and all(v >= 0 for v in pr[key]), (key, pr[key])
    assert min(pr["gemm_ms"]) > 0 and max(pr["comm_ms"]) > 0
    assert {"comm_xfer", "comm_wait", "comm_vec", "comm_wait_vec"} <= set(out["kernels_ms_per_step"])


def _gpu_and_shm_room():
    """(free HBM in GB, free /dev/shm in GB) — the full-size 8-rank test below needs 8 x 21 GB of HBM and room for
    the mock transport's message files."""
    from seekr_amd import _lib
    ctx = _lib.default_context()
    free_hbm = ctx.mem_info()[0] / 1e9 if hasattr(ctx, "mem_info") else 0.0
    st = os.statvfs("/dev/shm")
    return free_hbm, st.f_bavail * st.f_frsize / 1e9


@pytest.mark.parametrize("layout", ["symmetric", "rowblock", "allgather"])
def test_eight_ranks_at_the_drivers_weak_scaling_size(layout, mock_lib):
    """VERDICT r3 #3b: `bench.py --gpus 8` exactly at the size the driver's 8-GPU run uses (141 424 rows, 17 678-row
    shards, r_row + r_col = 20 GB per rank: 160 GB of the 288 on this one GPU), eight rank processes over the mock
    transport, each of the three layouts: real buffer sizes, ticket recycling and 64-bit offsets through the whole
    stack; every rank verified against the oracle, all eight seen; the column-sum chain's A/B (send/recv against the
    peer mailboxes, host-side wait because the ranks share a GPU) runs after the timed region and must agree bit for bit."""
    import json
    free_hbm, free_shm = _gpu_and_shm_room()
    if free_hbm < 200 or free_shm < 24:
        pytest.skip("needs 200 GB of free HBM and 24 GB of /dev/shm (have %.0f / %.0f)" % (free_hbm, free_shm))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--layout", layout,
           "--launch-timeout", "900"]
    env = _launcher_env(mock_lib, SEEKR_BENCH_CHAIN_AB="1", SEEKR_CHAIN_HOST_WAIT="1")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-6000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert "layout_fallback" not in out, out.get("layout_fallback")
    assert out["n_gpus"] == 8 and out["n_ranks_seen"] == 8
    assert out["config"]["rows_total"] == 141424 and out["config"]["rows_per_gpu"] == 17678
    assert out["verified"] is True and out["verified_detail"]["worst_error_over_bar"] <= 1.0
    want = {"symmetric": "symmetric half-ring", "rowblock": "row blocks", "allgather": "row blocks after one all-gather"}[layout]
    assert out["config"]["layout"].startswith(want)
    ab = out["chain_ab"]
    assert ab["bit_identical"] is True and "peer mailboxes" in ab["mailbox"]["transport"], ab
    assert not ab["mailbox"]["a_link_gave_up_waiting"] and ab["rccl"]["transport"] in ("not requested", "send/recv")
    assert len(out["per_rank"]["gemm_ms"]) == 8 and min(out["per_rank"]["gemm_ms"]) > 0
    check_scale_line(out, 8, "weak")


SCALE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_count", "n_ranks_seen", "per_rank", "verified", "verified_detail")


def check_scale_line(out, n, scaling):
    """What the driver's SCALE run (bench.py --gpus N for N in 1, 2, 4, 8) must find in the line the first time it is ever
    run on a real node: the contract's keys, the sharding and scaling named, every rank seen by an RCCL
    all-reduce, a verification all-reduced over the ranks, per-rank times, and the roofline object of the dominant kernel."""
    for key in SCALE_KEYS:
        assert key in out, key
    assert out["n_gpus"] == n and out["n_ranks_seen"] == n and out["scaling"] == scaling and out["vs_baseline"] is None
    assert out["higher_is_better"] is True and out["data"] == "synthetic" and out["value"] > 0 and out["ms_per_step"] > 0
    cfg = out["config"]
    assert cfg["sharding"] == "rows x%d" % n and "workload" in cfg and cfg["k"] == 6 and cfg["rows_total"] >= n * cfg["rows_per_gpu"] - n
    assert out["verified"] is True and "all-reduced" in out["verified_detail"]["bar"] and out["verified_detail"]["worst_error_over_bar"] <= 1.0
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert out["roofline_count"]["bound"] == "hbm" and out["roofline_count"]["unit"] == "GB/s"
    for name in ("gemm_ms", "comm_ms"):
        assert len(out["per_rank"][name]) == n
    assert "cpu_baseline" not in out and "e2e" not in out  # rank 0 at N = 1 only (the contract), never at N > 1


def test_the_strong_scaling_spelling_of_the_scale_command(mock_lib):
    """`bench.py --gpus 8 --rows R`: a fixed transcript set cut over the ranks (config 4 is R = 200 000: README) — the line
    says "strong" and carries the same contract.  R = 60 000 here (eight rank processes share this one GPU)."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rows", "60000", "--steps", "2", "--warmup", "1",
           "--launch-timeout", "600"]
    res = subprocess.run(cmd, env=_launcher_env(mock_lib), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-6000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert "layout_fallback" not in out, out.get("layout_fallback")
    check_scale_line(out, 8, "strong")
    assert out["config"]["rows_total"] == 60000 and out["config"]["rows_per_gpu"] == 7500 and out["steps"] == 2


def test_bench_rank_gives_up_when_a_peer_never_arrives(mock_lib):
    """bench.py as rank 0 of 2 (the shape torch.distributed.run gives it) while rank 1 never starts: the rank sits inside a
    collective's C call for ever — a watchdog thread says where it stood after --launch-timeout seconds and leaves with
    exit code 3, so that the launcher (or torchrun) ends the job instead of waiting for its own limit."""
    import time
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               SEEKR_TEST_HOOKS="1", SEEKR_RCCL_LIB=mock_lib, SEEKR_FORCE_DEVICE="0", MOCK_RCCL_ASYNC="1")
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--rows", "4000",
                          "--length", "400", "--launch-timeout", "6"], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 3, (res.returncode, res.stderr[-1500:])
    import re
    said = re.search(r"no progress for (\d+) s, last stage:", res.stderr)  # (the watchdog looks once a second: 6 s, or 7 on a busy box)
    assert said and 6 <= int(said.group(1)) <= 9 and time.time() - t0 < 60, res.stderr[-500:]
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_falls_back_to_row_blocks_when_the_selftest_fails(mock_lib):
    """The half-ring self-test fails on rank 1 of 3 (test hook): every rank leaves with the self-test's exit code, and the
    launcher starts a NEW set of rank processes with --layout allgather; the line says so."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "1", "--rows", "4500",
           "--length", "500"]
    res = subprocess.run(cmd, env=_launcher_env(mock_lib, SEEKR_BENCH_FAIL_SELFTEST="1"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "self-test FAILED" in res.stderr and "retrying once" in res.stderr and "failure injected" in res.stderr
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 3 and out["n_ranks_seen"] == 3 and out["config"]["layout"].startswith("row blocks after one all-gather")
    assert "self-test failed" in out["layout_fallback"] and out["value"] > 0


def test_every_rank_switches_when_one_needs_the_fp32_kernel(mock_lib, tmp_path):
    """Rank 0 holds two identical homopolymers (one-hot raw count rows): its operand falls back to the
    float32 layout, the flag is all-reduced and the other ranks follow, so shards stay compatible;
    the pair's r is 1 to float32 accuracy (the split contraction alone would give 1 - 2.4e-4)."""
    size, n_total, length, k = 3, 700, 600, 6
    env = dict(os.environ, WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_RANK="0",
               SEEKR_TEST_HOOKS="1", SEEKR_RCCL_LIB=mock_lib, MOCK_RCCL_ASYNC="1", MOCK_RAW_HOMOPOLYMERS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mock_rank_worker.py"), str(tmp_path), str(n_total), str(length),
                               str(k)], env=dict(env, RANK=str(rank)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for rank in range(size)]
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out.decode()[-3000:])
    parts = [np.load(str(tmp_path / ("rank%d.npz" % rank))) for rank in range(size)]
    r0 = parts[0]["r"]                       # rank 0's row block [n_0, N]
    assert abs(float(r0[0, 1]) - 1.0) < 2e-6 and abs(float(r0[1, 0]) - 1.0) < 2e-6
    full = np.concatenate([p["r"] for p in parts], axis=0)
    assert np.allclose(full, full.T, rtol=0, atol=1e-6) and np.allclose(np.diag(full), 1.0, atol=2e-6)
    x = np.concatenate([p["x"] for p in parts], axis=0)
    from oracle import seekr_oracle as orc
    assert np.allclose(full, orc.pearson(x, x), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("mode,want_kind", [("keep", 3), ("route", 2)])
def test_f16f8_across_ranks(mode, want_kind, mock_lib, tmp_path):
    """The opt-in two-product-unit precision with three ranks: every shard keeps the H / X layout (kind 3), or — rank 0's
    rows are few-valued, its fill routes them back to the three-product split — the all-reduced verdict makes EVERY rank
    refill as f16x3 (kind 2), so that shards stay compatible; r (row blocks and the half ring) inside the bar of float64
    either way."""
    from x8_case import x8_matrix
    from oracle import seekr_oracle as orc
    size, n_total, cols = 3, 700, 4096
    env = dict(os.environ, WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_RANK="0",
               SEEKR_TEST_HOOKS="1", SEEKR_RCCL_LIB=mock_lib, MOCK_RCCL_ASYNC="1", MOCK_X8=mode)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mock_rank_worker.py"), str(tmp_path), str(n_total), "600", "6"],
                              env=dict(env, RANK=str(rank)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for rank in range(size)]
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out.decode()[-3000:])
    parts = [np.load(str(tmp_path / ("rank%d.npz" % rank))) for rank in range(size)]
    assert [int(p["kind"]) for p in parts] == [want_kind] * size
    bounds = [int(parts[0]["lo"])] + [int(p["hi"]) for p in parts]
    x = x8_matrix(n_total, cols, mode == "route", bounds[1])
    truth = orc.pearson_f64_truth(x, x)
    bar = 2e-6 + 1e-5 * np.abs(truth)
    full = np.concatenate([p["r"] for p in parts], axis=0)
    assert (np.abs(full - truth) <= bar).all(), float((np.abs(full - truth) / bar).max())
    sym = np.zeros_like(full)
    for p in parts:
        for which, br, bc, nr, nc, gr, gc in p["blocks"]:
            buf = p["r_row"] if which == 0 else p["r_col"]
            sym[gr:gr + nr, gc:gc + nc] = buf[br:br + nr, bc:bc + nc]
    assert (np.abs(sym - truth) <= bar).all(), float((np.abs(sym - truth) / bar).max())


def test_coherent_flag_is_global(mock_lib, tmp_path):
    """only the last rank holds rows that are mostly one repeated value.  Its flag (skr_operand_coherent)
    is all-reduced, so every rank restarts the accumulators of every block every 1 024 columns; at K = 16 384 the
    uncorrected accumulation of a cross block would sit at 1.04 x the bar (tools/margin_probe.py)."""
    from coherent_case import coherent_matrix
    from oracle import seekr_oracle as orc
    size, n_total, cols = 2, 96, 16384
    env = dict(os.environ, WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_RANK="0",
               SEEKR_TEST_HOOKS="1", SEEKR_RCCL_LIB=mock_lib, MOCK_RCCL_ASYNC="1", MOCK_COHERENT_LAST_RANK="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mock_rank_worker.py"), str(tmp_path), str(n_total), "600", "7"],
                              env=dict(env, RANK=str(rank)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for rank in range(size)]
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out.decode()[-3000:])
    parts = [np.load(str(tmp_path / ("rank%d.npz" % rank))) for rank in range(size)]
    x = coherent_matrix(n_total, cols, size)
    truth = orc.pearson_f64_truth(x, x)
    bar = 2e-6 + 1e-5 * np.abs(truth)
    full = np.concatenate([p["r"] for p in parts], axis=0)          # row-block layout
    assert (np.abs(full - truth) <= bar).all(), float((np.abs(full - truth) / bar).max())
    sym = np.zeros_like(full)
    for p in parts:
        for which, br, bc, nr, nc, gr, gc in p["blocks"]:
            buf = p["r_row"] if which == 0 else p["r_col"]
            sym[gr:gr + nr, gc:gc + nc] = buf[br:br + nr, bc:bc + nc]
    assert (np.abs(sym - truth) <= bar).all(), float((np.abs(sym - truth) / bar).max())


@pytest.mark.parametrize("n_total,size", [(6, 4), (6, 8), (3, 3), (17, 5)])
def test_tiny_and_empty_shards(n_total, size, mock_lib, tmp_path):
    """Fewer rows than ranks (empty shards), one row per rank, odd splits: k = 2, i.e. 16 columns and the
    float32 contraction."""
    check_ranks(size, 1, mock_lib, single_gpu_reference(n_total, 400, 2), tmp_path)
