"""End-to-end CLI-shaped timing on the GPU box: FASTA -> counts -> CSV / .npy on disk, native
writers vs numpy's (config-2 shape scaled by --rows)."""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seekr_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=20000)
args = ap.parse_args()
ctx = L.default_context()
rng = np.random.default_rng(0)
a = np.log2(rng.binomial(30, 0.1, size=(args.rows, 4096)).astype(np.float32) * np.float32(0.5) + 1) - np.float32(0.7)
dev = ctx.from_numpy(a)
d = tempfile.mkdtemp(dir="/tmp")
n = a.size
t = time.time(); L.save_csv(os.path.join(d, "dev.csv"), dev); t_dev = time.time() - t
t = time.time(); L.save_csv(os.path.join(d, "host.csv"), a); t_host = time.time() - t
sub = a[:500]
t = time.time(); np.savetxt(os.path.join(d, "np.csv"), sub, delimiter=",", fmt="%1.6f"); t_np = (time.time() - t) * n / sub.size
t = time.time(); L.save_npy(os.path.join(d, "dev.npy"), dev); t_npy = time.time() - t
t = time.time(); np.save(os.path.join(d, "np.npy"), a); t_npnpy = time.time() - t
size = os.path.getsize(os.path.join(d, "dev.csv"))
print("matrix %d x 4096 (%.2f G numbers, csv %.2f GB)" % (args.rows, n / 1e9, size / 1e9))
print("csv  device->file %.2f s (%.0f M numbers/s, %.2f GB/s)   host->file %.2f s   numpy.savetxt (extrapolated) %.0f s"
      % (t_dev, n / t_dev / 1e6, size / t_dev / 1e9, t_host, t_np))
print("npy  device->file %.2f s (%.2f GB/s)   numpy.save of a host array %.2f s" % (t_npy, a.nbytes / t_npy / 1e9, t_npnpy))
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
