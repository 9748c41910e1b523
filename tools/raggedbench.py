"""Times a ragged short-read batch (query 100-150, target 250-300) in Global and Semiglobal mode and checks
a sample against nothing but its own CIGARs (re-scoring); prints the kernel family chosen."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sequencealigner_amd import align, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
q, t = synth.pairs_short_read_window(n, seed=31)
rng = np.random.default_rng(7)
ql = rng.integers(100, 151, n); tl = rng.integers(250, 301, n)
qs = [bytes(q[i, :ql[i]]) for i in range(n)]
ts = [bytes(t[i, :tl[i]]) for i in range(n)]
for typ, name in ((0, "global"), (2, "semiglobal"), (1, "local")):
    b = align.DeviceAlignBatch(qs, ts, typ, 1, -1, -1)
    b.run(); torch.cuda.synchronize()
    t0 = time.time(); b.run(); torch.cuda.synchronize(); dt = time.time() - t0
    print(name, "kernel", b.kernel_name(), "%.2f ms" % (dt * 1e3), "%.0f GCUPS" % (b.cells / dt / 1e9), flush=True)
