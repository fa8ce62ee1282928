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
    declared = set(re.findall(r"\b(skr_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.skr_abi_version() == 1


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
