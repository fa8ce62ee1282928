"""bench.end_to_end's own statements with a copy-rate probe between them (a new 190 MB upload + a 549 MB download into a
registered, touched array), in the order bench.py runs them: FIRST GPU work of the process = the FASTA path."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
if "--import-bench" in sys.argv:
    import bench  # noqa: E402,F401
from seekr_amd import _lib  # noqa: E402
from seekr_amd.kmer_counts import BasicCounter  # noqa: E402
from seekr_amd.pearson import pearson  # noqa: E402
from seekr_amd.synthetic import synthetic_ascii  # noqa: E402

ctx = _lib.default_context()
state = {}


def probe(tag):
    if "r" not in state:
        state["head"] = np.random.default_rng(0).random((12000, 4096), dtype=np.float32)
        d = ctx.from_numpy(state["head"])
        state["r"] = _lib.pearson(ctx, d, d, precision=_lib.PREC_F16X3)
        state["keep"] = np.zeros((12000, 12000), np.float32)
    tu, td = [], []
    for _ in range(3):
        t0 = time.perf_counter(); x = ctx.from_numpy(state["head"]); ctx.sync(); tu.append((time.perf_counter() - t0) * 1e3); x.free()
        t0 = time.perf_counter(); state["r"].to_numpy(out=state["keep"]); td.append((time.perf_counter() - t0) * 1e3)
    print("%-52s upload %.1f ms  download %.1f ms" % (tag, min(tu), min(td)), flush=True)


warm = [a for a in sys.argv if a.startswith("--warm=")]
if warm:
    mb = int(warm[0].split("=")[1])
    w = np.ones(mb << 18, np.float32)
    t0 = time.perf_counter()
    dm = ctx.from_numpy(w.reshape(-1, 1024))
    dm.to_numpy(out=w.reshape(-1, 1024))
    dm.free()
    print("warm-up with %d MB up and down: %.1f ms" % (mb, (time.perf_counter() - t0) * 1e3))
early = "--probe-first" in sys.argv
if early:
    probe("start (transfers before the FASTA path)")
blob, _ = synthetic_ascii(2, 50000, 2000)
rows = blob.reshape(50000, 2000)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "cfg.fa")
    with open(path, "wb") as fh:
        fh.write(b"".join(b">s%d\n" % i + rows[i].tobytes() + b"\n" for i in range(50000)))
    for i in range(3):
        c = BasicCounter(path, k=6, mean=False, std=False, log2="Log2.none", silent=True)
        c.get_counts()
        probe("after BasicCounter + get_counts #%d" % i)
head = np.ascontiguousarray(c.counts[:12000])
pearson(head[:256], head[:256])
for i in range(3):
    t0 = time.perf_counter()
    pearson(head, head)
    print("pearson(head, head): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
probe("end")
