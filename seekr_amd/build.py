"""Build libseekr_hip.so (hipcc, gfx950 only) in-tree.  `python -m seekr_amd.build [--force] [--diag]`.

`--diag` additionally links libseekr_hip_diag.so: the same objects except pearson_bf16.hip, which is compiled a second
time with -DSEEKR_DIAG (the stamping instance of the contraction and its timing experiments, tools/gemm_diag.py).  The
production library holds none of that."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libseekr_hip.so")
DIAG_LIB = os.path.join(HERE, "libseekr_hip_diag.so")
DIAG_SOURCES = ["pearson_bf16.hip"]  # compiled a second time with -DSEEKR_DIAG for the diagnostic library
SOURCES = ["ctx.hip", "pack.hip", "count.hip", "normalize.hip", "normalize_any.hip", "pearson.hip", "pearson_bf16.hip", "operand.hip",
           "consumers.hip", "fused_edges.hip", "comm.hip", "io.hip", "csv_read.hip", "host_api.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(HERE, "..", "include")]
PER_FILE = {
    # numpy-order parity: no FMA contraction anywhere in the normalisation / scaling arithmetic
    "normalize.hip": ["-ffp-contract=off"],
    "normalize_any.hip": ["-ffp-contract=off"],
    "count.hip": ["-ffp-contract=off"],
    # the 4-wave geometry of the contraction has 64 accumulator tiles per lane: its epilogue loops are only unrolled (and
    # the accumulator array only kept in registers) above clang's default 16 384-instruction limit for `#pragma unroll`
    "pearson_bf16.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
}


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src, *extra))


def build(force=False, verbose=True, diag=False):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.hpp"), os.path.join(HERE, "..", "include", "seekr_hip.h"),
               os.path.abspath(__file__)]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    jobs = []
    objs = []
    for name in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJDIR, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj, headers):
            jobs.append([HIPCC, *COMMON, *PER_FILE.get(name, []), "-c", src, "-o", obj])
    diag_objs, diag_jobs = list(objs), 0
    if diag:
        for name in DIAG_SOURCES:
            src = os.path.join(CSRC, name)
            obj = os.path.join(OBJDIR, name.replace(".hip", ".diag.o"))
            diag_objs[SOURCES.index(name)] = obj
            if force or _newer(src, obj, headers):
                jobs.append([HIPCC, *COMMON, *PER_FILE.get(name, []), "-DSEEKR_DIAG", "-c", src, "-o", obj])
                diag_jobs += 1

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
        if verbose and proc.stderr.strip():
            print(proc.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(run, jobs))
    if len(jobs) > diag_jobs or force or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"])
    if diag and (jobs or force or not os.path.exists(DIAG_LIB)):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", DIAG_LIB, *diag_objs, "-ldl", "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv))
