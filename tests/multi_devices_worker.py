"""One process's run of the public surface — BasicCounter, pearson(), the three console commands — whose every output is
written to `out_dir`.  tests/test_gpu_multi_devices.py runs it with SEEKR_DEVICES unset, then with several 'devices' (ranks
on one GPU over tests/mock_rccl, or real GPUs) and with forced stripe heights, and compares the directories byte for byte:
the number of GPUs behind the reference's API must not show in any result."""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def cli(entry, *argv):
    old = sys.argv
    sys.argv = [entry.__name__] + [str(a) for a in argv]
    try:
        with contextlib.redirect_stderr(io.StringIO()):
            entry()
    finally:
        sys.argv = old


def main():
    out_dir, scale = sys.argv[1], sys.argv[2]
    from inputs import EXAMPLE_FA, skewed_set, synth_2000, write_fasta
    from seekr_amd import console_scripts as cs
    from seekr_amd.kmer_counts import BasicCounter
    from seekr_amd.pearson import pearson, pearson_to_file
    from seekr_amd.synthetic import synthetic_codes
    out = {}
    log = io.StringIO()

    def save(name, a):
        out[name] = np.ascontiguousarray(a)

    def counter(seqs=None, **kw):
        c = BasicCounter(silent=True, **kw)
        if seqs is not None:
            c.seqs = list(seqs)
        with contextlib.redirect_stdout(log):
            c.get_counts()
        return c

    def path(name):
        return os.path.join(out_dir, name)

    # ---- config 1: the reference's example.fa, five sequences of 6..76 nt (fewer rows than ranks at 8 'devices')
    with open(path("example.fa"), "w") as fh:
        fh.write(EXAMPLE_FA)
    for k in (1, 2, 3):
        c = BasicCounter(path("example.fa"), k=k, mean=False, std=False, log2="Log2.none", silent=True)
        c.get_counts()
        save("example_raw_k%d" % k, c.counts)
    c = BasicCounter(path("example.fa"), k=2, silent=True)
    c.get_counts()
    save("example_k2", c.counts), save("example_k2_mean", c.mean), save("example_k2_std", c.std)
    # ---- golden set g3: 111 / 151 skewed sequences, the norm-vector flow, every log2 mode, k = 6 all-NaN + warning
    s1, s2 = skewed_set(101, 111), skewed_set(202, 151)
    write_fasta(path("s1.fa"), s1, width=70)
    write_fasta(path("s2.fa"), s2, crlf=True)
    for k in (4, 5):
        nv = BasicCounter(path("s1.fa"), k=k, silent=True)
        nv.get_counts()
        save("g3_mean_k%d" % k, nv.mean), save("g3_std_k%d" % k, nv.std), save("g3_self_k%d" % k, nv.counts)
        c1 = counter(s1, k=k, mean=nv.mean, std=nv.std)
        c2 = BasicCounter(path("s2.fa"), k=k, mean=nv.mean, std=nv.std.astype(np.float64), silent=True)
        c2.get_counts()
        save("g3_c1_k%d" % k, c1.counts), save("g3_c2_k%d" % k, c2.counts)
        save("g3_r_k%d" % k, pearson(c1.counts, c2.counts))
    for tag in ("pre", "none"):
        c = counter(s1, k=4, log2="Log2." + tag)
        save("g3_%s" % tag, c.counts), save("g3_%s_mean" % tag, c.mean), save("g3_%s_std" % tag, c.std)
    x6 = counter(s1, k=6)
    save("g3_k6_nan", x6.counts), save("g3_k6_r", pearson(x6.counts, x6.counts))
    a, b = out["g3_c1_k4"][:7], out["g3_c1_k4"][7:12]
    save("g3_nostd", pearson(a, b, row_standardize=False))
    save("g3_f64", pearson(a.astype(np.float64), b.astype(np.float64)))
    save("g3_mixed", pearson(a, b.astype(np.float64)))
    save("g3_mixed2", pearson(a.astype(np.float64), b))
    save("g3_int", pearson(np.arange(40).reshape(5, 8) % 7, np.arange(48).reshape(6, 8) % 5))
    # ---- any alphabet (the general counting kernel), sequences assigned by hand
    c = counter([s.replace("A", "N", 3) for s in s1[:40]], k=3, alphabet="ACGTN")
    save("acgtn_k3", c.counts), save("acgtn_k3_mean", c.mean)
    # ---- golden set g4: 2 000 x 2 kb at k = 6; self, cross, float64, a constant row (NaN row and column)
    seqs = synth_2000() if scale == "full" else synth_2000()[:600]
    c = counter(seqs, k=6)
    save("g4_counts", c.counts), save("g4_mean", c.mean), save("g4_std", c.std)
    save("g4_r", pearson(c.counts, c.counts))
    n = len(seqs)
    save("g4_cross", pearson(c.counts[:n // 3], c.counts[n // 5:]))
    f64 = c.counts[:n // 2].astype(np.float64)
    f64[7] = 1.5
    save("g4_f64", pearson(f64, f64)), save("g4_f64_cross", pearson(f64[:100], f64[50:]))
    # ---- 6 000 synthetic transcripts of 600 nt, k = 6, from a FASTA file
    n_big = 6000 if scale == "full" else 1500
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    big = [r.tobytes().decode() for r in letters[synthetic_codes(3, n_big, 600)]]
    write_fasta(path("big.fa"), big)
    cb = BasicCounter(path("big.fa"), k=6, silent=True)
    cb.get_counts()
    save("big_counts", cb.counts), save("big_mean", cb.mean), save("big_std", cb.std)
    save("big_r", pearson(cb.counts, cb.counts))
    os.environ["SEEKR_PRECISION"] = "fp32"
    save("big_r_fp32", pearson(cb.counts[:1000], cb.counts[:1000]))
    os.environ["SEEKR_PRECISION"] = "f16f8"
    save("big_r_f16f8", pearson(cb.counts[:1200], cb.counts[:1200]))
    del os.environ["SEEKR_PRECISION"]
    # ---- the three console commands, file to file
    cli(cs.console_kmer_counts, path("s1.fa"), "-o", path("cli_counts.csv"), "-k", 4)
    cli(cs.console_kmer_counts, path("s1.fa"), "-o", path("cli_counts.npy"), "-k", 4, "-b", "-rl")
    cli(cs.console_kmer_counts, path("s2.fa"), "-o", path("cli_counts_plain.csv"), "-k", 3, "-rl", "-l", "Log2.pre")
    cli(cs.console_norm_vectors, path("big.fa"), "-k", 5, "-mv", path("cli_mean.npy"), "-sv", path("cli_std.npy"))
    cli(cs.console_kmer_counts, path("s2.fa"), "-o", path("cli_counts_vec.npy"), "-k", 5, "-b", "-rl", "-mv", path("cli_mean.npy"),
        "-sv", path("cli_std.npy"))
    cli(cs.console_pearson, path("cli_counts.csv"), path("cli_counts.csv"), "-o", path("cli_r.csv"))
    cli(cs.console_pearson, path("cli_counts.npy"), path("cli_counts.npy"), "-o", path("cli_r.npy"), "-bi", "-bo")
    cli(cs.console_kmer_counts, path("s2.fa"), "-o", path("cli_counts2.npy"), "-k", 4, "-b", "-rl", "-uc", "-us", "-l", "Log2.none")
    cli(cs.console_pearson, path("cli_counts.npy"), path("cli_counts2.npy"), "-o", path("cli_r_from_npy.csv"), "-bi")
    cli(cs.console_pearson, path("cli_counts.csv"), path("cli_counts.csv"), "-o", path("cli_r64.npy"), "-bo")  # float64, straight to the file
    pearson_to_file(cb.counts, cb.counts[:777], path("big_r_cross_file"))  # np.save appends .npy
    pearson_to_file(cb.counts[:900].astype(np.float64), cb.counts[:333], path("mixed_r_file.npy"), row_standardize=False)
    # ---- an exception raised for one range: once, itself; the next call works
    bad = list(s1[:60])
    bad[41] = "ACG"  # k = 4: W = 0 (kmer_counts.py:144)
    try:
        counter(bad, k=4)
        save("zerodiv", np.array([0]))
    except ZeroDivisionError as e:
        save("zerodiv", np.array([1]))
        out["zerodiv_text"] = np.array(str(e))
    write_fasta(path("blank.fa"), s1[:5])
    with open(path("blank.fa"), "a") as fh:
        fh.write(">x\nACGT\n\n>y\nAC\n")
    try:
        BasicCounter(path("blank.fa"), k=2, silent=True)
        out["blank_text"] = np.array("no error")
    except IndexError as e:
        out["blank_text"] = np.array(str(e))
    c = counter(s1[:60], k=4)
    save("after_error", c.counts)
    # ---- golden set G10: files with bytes >= 0x80 (read in text mode, fasta_reader.py:44): counts or the reference's exception
    import json
    with open(os.path.join(ROOT, "tests", "golden", "g10_non_ascii.json")) as fh:
        g10 = json.load(fh)["cases"]
    for case in g10:
        with open(path("g10.fa"), "wb") as fh:
            fh.write(bytes.fromhex(case["hex"]))
        try:
            c = BasicCounter(path("g10.fa"), k=2, mean=False, std=False, log2="Log2.none", silent=True)
            c.get_counts()
            save("g10_" + case["name"], c.counts)
            out["g10_seqs_" + case["name"]] = np.array("\n".join(c.seqs))
        except Exception as e:  # noqa: BLE001 - compared with the reference's own
            out["g10_" + case["name"]] = np.array(type(e).__name__ + ": " + str(e))
    with open(path("g10.fa"), "wb") as fh:
        fh.write(bytes.fromhex([c for c in g10 if c["name"] == "mixed_records"][0]["hex"]))
    c = BasicCounter(path("g10.fa"), k=2, silent=True)
    c.get_counts()
    save("g10_mixed_normalised", c.counts), save("g10_mixed_mean", c.mean)
    # ... and through the command (console_scripts.py:564-572): raw counts of the same file to .npy and to the labelled csv
    cli(cs.console_kmer_counts, path("g10.fa"), "-o", path("cli_g10_raw.npy"), "-k", 2, "-b", "-rl", "-uc", "-us", "-l", "Log2.none")
    with open(path("g10.fa"), "wb") as fh:
        fh.write(bytes.fromhex([c for c in g10 if c["name"] == "non_ascii_only_in_headers"][0]["hex"]))
    cli(cs.console_kmer_counts, path("g10.fa"), "-o", path("cli_g10_labelled.csv"), "-k", 2)
    os.remove(path("g10.fa"))
    out["stdout"] = np.array(log.getvalue())
    np.savez(path("results.npz"), **out)
    # what actually ran (not compared between settings): the device list, the live group's size, the stripe height
    from seekr_amd import multi
    with open(path("info.json"), "w") as fh:
        json.dump({"devices": multi.requested_devices(), "group_size": multi._group.size if multi._group else 0,
                   "group_broken": bool(multi._group.broken) if multi._group else None,
                   "transport": multi._group.transport if multi._group else None,
                   "n_ranks_seen": multi.group_info()["n_ranks_seen"], "ipc_env_at_load": multi.group_info()["ipc_env_at_load"],
                   "threads": sorted(t.name for t in multi._group._threads) if multi._group else [],
                   "stripe_rows": multi.forced_stripe_rows()}, fh)
    for name in ("example.fa", "s1.fa", "s2.fa", "big.fa", "blank.fa"):
        os.remove(path(name))


if __name__ == "__main__":
    main()
