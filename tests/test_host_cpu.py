"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the product
fails loudly without a GPU, and the host mirrors (Reader, CLI parsers) behave like the oracle."""
import os
import subprocess
import re
import sys

import numpy as np
import pytest

from oracle import seekr_oracle as orc
from inputs import EXAMPLE_FA, skewed_set, write_fasta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from seekr_amd import _lib
    handle = _lib.lib()
    header = open(os.path.join(ROOT, "include", "seekr_hip.h")).read()
    product, _, diag = header.partition("#ifdef SEEKR_DIAG")
    declared = set(re.findall(r"\b(skr_[a-z0-9_]+)\s*\(", product))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.skr_abi_version() == 1
    # the diagnostic entry points (and the stamping instance of the contraction with its result-corrupting switches)
    # live in libseekr_hip_diag.so only: the production library must not export them
    diag_declared = set(re.findall(r"\b(skr_[a-z0-9_]+)\s*\(", diag.partition("#endif")[0]))
    assert diag_declared == set(_lib.DIAG_SIGNATURES), diag_declared ^ set(_lib.DIAG_SIGNATURES)
    for name in diag_declared:
        assert not hasattr(handle, name), name + " is exported by the production library"
    # ... nor does it hold the stamping instance of the contraction (PERSIST = true, DIAG = true) or the 4-wave A/B arm
    blob = open(_lib.LIB_PATH, "rb").read()
    assert not re.search(rb"split16_kernel\w*Lb1ELb1E", blob), "the production library holds the DIAG instance of the contraction"
    assert not re.search(rb"split16_kernel\w*Li4EEEv", blob), "the production library holds the 4-wave arm of the contraction"
    assert re.search(rb"split16_kernel\w*Lb1ELb0ELi8EEEv", blob), "the shipped instance of the contraction is missing"


def test_no_launch_path_reads_the_environment():
    """ADVICE r2 / VERDICT r2 weak 8: environment switches are read in skr_ctx_reload_knobs (ctx creation), never per
    launch; what is left are the two host-side test hooks (FASTA piece size, the RCCL library override)."""
    import glob
    allowed = {"ctx.hip": 1, "pack.hip": 1, "comm.hip": 2}
    for path in glob.glob(os.path.join(ROOT, "seekr_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "seekr_amd", "csrc", "*.hpp")):
        n = len(re.findall(r"\bgetenv\s*\(", open(path).read()))
        assert n <= allowed.get(os.path.basename(path), 0), (path, n)


def test_product_fails_loudly_without_gpu():
    from seekr_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    from seekr_amd.kmer_counts import BasicCounter
    from seekr_amd.pearson import pearson
    c = BasicCounter(k=2, silent=True)
    c.seqs = ["ACGT", "AACC"]
    with pytest.raises(_lib.SeekrHipError):
        c.get_counts()
    with pytest.raises(_lib.SeekrHipError):
        pearson(np.eye(3, dtype=np.float32), np.eye(3, dtype=np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "seekr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), (f, "mentions the oracle")


def test_reader_mirror_matches_oracle_reader(tmp_path):
    from seekr_amd.fasta_reader import Reader
    p = tmp_path / "example.fa"
    p.write_text(EXAMPLE_FA)
    headers, seqs = orc.read_fasta(str(p))
    assert Reader(str(p)).get_seqs() == seqs and Reader(str(p)).get_headers() == headers
    pairs, h2, s2 = Reader(str(p)).get_data()
    assert list(pairs) == list(zip(headers, seqs)) and h2 == headers and s2 == seqs
    rs = skewed_set(5, 9, 30, 400)
    q = str(tmp_path / "ml.fa")
    write_fasta(q, rs, width=70, crlf=True, lower=True)
    assert Reader(q).get_seqs() == rs == orc.read_fasta(q)[1]
    rd = Reader(q, outfasta=str(tmp_path / "out.fa"), names=iter(["n%d" % i for i in range(9)]))
    rd.get_lines()
    rd.data = rd.supply_basic_header()
    rd.save()
    text = open(str(tmp_path / "out.fa")).read().splitlines()
    assert text[0] == ">||||n0||{}|".format(len(rs[0])) and text[1] == rs[0]
    for bad, exc in ((">a\nAC\n\n>b\nAC\n", IndexError), (">a\nAC\n>b\n>c\nAC\n", AssertionError)):
        b = tmp_path / "bad.fa"
        b.write_text(bad)
        with pytest.raises(exc):
            Reader(str(b)).get_seqs()


def test_cli_flag_surface(monkeypatch, capsys):
    from seekr_amd import console_scripts as cs
    seen = {}
    monkeypatch.setattr(cs, "_run_kmer_counts", lambda *a: seen.setdefault("kc", a))
    monkeypatch.setattr(cs, "_run_pearson", lambda *a: seen.setdefault("p", a))
    monkeypatch.setattr(cs, "_run_norm_vectors", lambda *a: seen.setdefault("nv", a))
    monkeypatch.setattr(sys, "argv", ["seekr_kmer_counts", "x.fa"])
    cs.console_kmer_counts()
    assert seen["kc"] == ("x.fa", "counts.seekr", 6, False, True, True, "Log2.post", False, None, None, "AGTC")
    seen.clear()
    monkeypatch.setattr(sys, "argv", ["seekr_kmer_counts", "x.fa", "-o", "o.npy", "-k", "5", "-b", "-uc", "-us", "-l",
                                      "Log2.pre", "-rl", "-mv", "m.npy", "-sv", "s.npy", "-a", "ACGT"])
    cs.console_kmer_counts()
    assert seen["kc"] == ("x.fa", "o.npy", 5, True, False, False, "Log2.pre", True, "m.npy", "s.npy", "ACGT")
    monkeypatch.setattr(sys, "argv", ["seekr_pearson", "a.npy", "b.npy", "-bi", "-bo"])
    cs.console_pearson()
    assert seen["p"] == ("a.npy", "b.npy", "pearson.seekr", True, True)
    monkeypatch.setattr(sys, "argv", ["seekr_norm_vectors", "g.fa", "-k", "4"])
    cs.console_norm_vectors()
    assert seen["nv"] == ("g.fa", "mean.npy", "std.npy", "Log2.post", 4)
    monkeypatch.setattr(sys, "argv", ["seekr_kmer_counts"])
    with pytest.raises(SystemExit) as e:
        cs.console_kmer_counts()
    assert e.value.code == 0 and "usage" in capsys.readouterr().out.lower()


def test_counter_constructor_rules_without_gpu():
    from seekr_amd.kmer_counts import BasicCounter
    c = BasicCounter(k=3, silent=True)
    assert c.seqs is None and c.alpha_len == 4 and len(c.kmers) == 64
    words, col = orc.kmer_vocabulary(3)
    assert c.kmers == words and c.map == col
    with pytest.raises(ValueError):
        BasicCounter(k=3, log2="log2")
    c = BasicCounter(k=2, alphabet="ACGT", mean=False, std=np.ones(16), silent=True)
    assert c.kmers[:5] == ["AA", "AC", "AG", "AT", "CA"] and c.mean is False


def _build_c_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "host_example")
    libdir = os.path.join(ROOT, "seekr_amd")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "host_example.c"), "-o", exe, "-L" + libdir, "-lseekr_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """include/seekr_hip.h must be usable from C (no C++-isms) and the library from a program that is not
    Python; without a GPU the program fails loudly with the library's error message."""
    import subprocess
    from seekr_amd import _lib
    _lib.lib()  # the shared library is built
    exe = _build_c_example(tmp_path)
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible: the run itself is covered by the gpu tests")
    fa = tmp_path / "x.fa"
    fa.write_text(">a\nACGTACGT\n>b\nTTTTGGGA\n")
    out = subprocess.run([exe, str(fa), "2", str(tmp_path / "c.npy"), str(tmp_path / "r.npy")], capture_output=True, text=True)
    assert out.returncode == 3 and "skr_ctx_create" in out.stderr  # 1 + |SKR_ERR_HIP|


def test_test_hooks_are_off_without_the_opt_in():
    """SEEKR_RCCL_LIB / SEEKR_FORCE_DEVICE are honoured only under SEEKR_TEST_HOOKS=1."""
    code = ("import os; from seekr_amd import launch; os.environ['RANK']='1'; os.environ['LOCAL_RANK']='5';"
            "print(launch.world()[2])")
    env = dict(os.environ, SEEKR_FORCE_DEVICE="0", PYTHONPATH=ROOT)
    env.pop("SEEKR_TEST_HOOKS", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.stdout.strip() == "5", out.stdout + out.stderr
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, SEEKR_TEST_HOOKS="1"), capture_output=True, text=True, timeout=120)
    assert out.stdout.strip() == "0", out.stdout + out.stderr


# ---------------------------------------------------------------- bench.py: launcher and measurement contract ----
def _bench():
    import importlib
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_bench_launcher_reports_every_rank_and_retries_once():
    """`python bench.py --gpus 2` without WORLD_SIZE starts its own rank processes.  Here there is no
    GPU (or, on a test box, one): both attempts fail, and the launcher must say so loudly — non-zero exit, every
    rank's stderr tail, the one retry with --layout allgather — instead of dying in argument checking or hanging."""
    from seekr_amd import _lib
    if _lib.device_count() >= 2:
        pytest.skip("two GPUs are visible: the launch would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--launch-timeout", "120"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 1 and res.stdout.strip() == ""
    err = res.stderr
    assert "attempt 1" in err and "attempt 2" in err and "retrying once" in err and "--layout allgather" in err
    for rank in (0, 1):
        assert err.count("---- rank %d (exit code" % rank) == 2
    assert "SeekrHipError" in err or "out of range" in err


def test_bench_launcher_kills_a_hung_rank_set(tmp_path):
    bench = _bench()
    t0 = __import__("time").time()
    ok, kind, report = bench._run_rank_set(["-c", "import sys, time; sys.stderr.write('hung rank\\n'); sys.stderr.flush(); time.sleep(600)"],
                                           2, 2.0, {}, str(tmp_path), 1, program=[sys.executable])
    assert not ok and "timeout" in kind and report.count("hung rank") == 2
    assert __import__("time").time() - t0 < 30
    # a rank set that works: rank 0's single JSON line comes back
    ok, line, _ = bench._run_rank_set(["-c", "import os; print('{\"rank\": %s}' % os.environ['RANK']) if os.environ['RANK'] == '0' else None"],
                                      3, 30.0, {}, str(tmp_path), 2, program=[sys.executable])
    assert ok and line == '{"rank": 0}'
    # the self-test's exit code is told apart from any other failure
    ok, kind, _ = bench._run_rank_set(["-c", "import sys; sys.exit(%d)" % bench.SELFTEST_EXIT], 2, 30.0, {}, str(tmp_path), 3,
                                      program=[sys.executable])
    assert not ok and kind == "selftest"


def test_pmc_traffic_refuses_a_summary_of_other_kernels(tmp_path):
    """bench.py quotes roofline.traffic from a committed --pmc summary only when that summary was
    taken on the same workload with a library holding the same kernel symbols as the one loaded now."""
    bench = _bench()
    lib_path = os.path.join(ROOT, "seekr_amd", "libseekr_hip.so")
    h = bench.kernel_symbols_sha256(lib_path)
    assert len(h) == 64
    wl = bench.workload_key(50000, 2000, 6, "f16x3", 1)
    body = "_ZN12_GLOBAL__N_127pearson_gemm_split16_kernelIDF16_Li3ELi1ELb1ELb0EEEv\n    FETCH_SIZE      1000\n    WRITE_SIZE      500\n"
    (tmp_path / "a_pmc_summary.txt").write_text("# workload: %s\n# kernel_symbols_sha256: %s\n%s" % (wl, "0" * 64, body))
    got, why = bench.pmc_traffic("split16_kernelIDF16_Li3", wl, lib_path, profiles_dir=str(tmp_path))
    assert got is None and "refused" in why
    (tmp_path / "b_pmc_summary.txt").write_text("# workload: %s\n# kernel_symbols_sha256: %s\n%s" % (wl, h, body))
    got, why = bench.pmc_traffic("split16_kernelIDF16_Li3", wl, lib_path, profiles_dir=str(tmp_path))
    assert why is None and got["bytes"] == (2 * 1000 + 500) * 1024.0
    got, why = bench.pmc_traffic("split16_kernelIDF16_Li3", bench.workload_key(200000, 2000, 6, "f16x3", 1), lib_path,
                                 profiles_dir=str(tmp_path))
    assert got is None and "no profiles" in why
    # the committed summaries either match the library that is built from this tree or are not quoted
    got, why = bench.pmc_traffic("split16_kernelIDF16_Li3", wl, lib_path)
    assert (got is None) != (why is None)


def test_committed_round6_summaries_belong_to_the_library_built_from_this_tree():
    """The round-6 PMC summaries under profiles/ carry the kernel-symbol fingerprint of the library this tree builds, so
    bench.py quotes roofline.traffic from them for the default, the 200 000-row, the k = 7 and the f16f8 workloads (a
    kernel added or removed without re-collecting them would silently turn `traffic` into null in the driver's line)."""
    bench = _bench()
    lib_path = os.path.join(ROOT, "seekr_amd", "libseekr_hip.so")
    for rows, length, k, prec, key in ((50000, 2000, 6, "f16x3", "split16_kernelIDF16_Li3"), (200000, 2000, 6, "f16x3", "split16_kernelIDF16_Li3"),
                                       (50000, 5000, 7, "f16x3", "split16_kernelIDF16_Li3"), (50000, 2000, 6, "f16f8", "split16_kernelIDF16_Li2")):
        got, why = bench.pmc_traffic(key, bench.workload_key(rows, length, k, prec, 1), lib_path)
        assert why is None and got["bytes"] > 0 and got["source"].startswith("profiles/r6_"), (rows, k, prec, why)
        count, why = bench.pmc_traffic("count_rows_kernel<0", bench.workload_key(rows, length, k, prec, 1), lib_path)
        assert why is None and count["write_bytes"] > 0, (rows, k, prec, why)


def test_pearson_result_dtype_rule_is_numpys_promotion_of_the_reference():
    """pearson.py:35-41: every float dtype survives np.mean / np.std, integers and bool become float64, np.inner takes the
    wider — `seekr_amd.pearson._result_dtype` against the oracle's numpy statement, no device involved."""
    from seekr_amd.pearson import _result_dtype
    sys.path.insert(0, ROOT)
    from oracle import seekr_oracle as orc
    a = np.arange(12).reshape(3, 4) % 5
    names = ["float16", "float32", "float64", "int8", "int32", "int64", "uint8", "uint64", "bool"]
    for d1 in names:
        for d2 in names:
            x, y = a.astype(d1), a[:2].astype(d2)
            with np.errstate(all="ignore"):
                assert _result_dtype(x, y) == orc.pearson(x, y).dtype, (d1, d2)


def test_user_vectors_follow_numpys_broadcasting_and_its_words():
    """`counts -= mean` / `counts /= std` (kmer_counts.py:169,175) with a user operand: every shape numpy spreads over the rows
    becomes the K column values; what numpy refuses is refused with numpy's own sentence (shapes of the in-place operation);
    an operand that varies along the rows is declined explicitly.  Host logic, no device."""
    from seekr_amd.kmer_counts import _column_vector
    n, k = 5, 16
    base = np.zeros((n, k), np.float32)
    for vec in (np.float64(0.5), np.arange(k, dtype=np.float64), np.ones(1), np.ones((1, k)), np.ones((1, 1)), np.ones(5), np.ones((2, k)),
                np.ones((1, 5)), np.ones((1, 1, k)), np.ones((n, 1)), np.ones((n, k)), np.ones((0,))):
        vec = np.asarray(vec)
        try:
            want = base.copy()
            want -= vec
            want_exc = None
        except ValueError as e:
            want_exc = str(e)
        try:
            got = _column_vector(vec, n, k)
            got_exc = None
        except ValueError as e:
            got_exc = str(e)
        except NotImplementedError:
            assert want_exc is None and vec.ndim == 2 and vec.shape[0] == n  # legal numpy, varies along the rows
            continue
        assert got_exc == want_exc, (vec.shape, got_exc, want_exc)
        if want_exc is None:
            assert got.shape == (k,) and np.array_equal(base - got, want)
