"""Times Global batches of several shapes with the kernel family chosen by the dispatcher and with each
family forced (sa_set_option), to check the dispatcher's rate estimates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sequencealigner_amd import align, capi, synth

def run(q, t, typ):
    b = align.DeviceAlignBatch(q, t, typ, 1, -1, -1)
    name = b.kernel_name()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        b.run(); torch.cuda.synchronize()
        best = min(best, time.time() - t0)
    return name, best, b.cells

for n, L in ((400_000, 200), (200_000, 300), (60_000, 600), (20_000, 1000), (100_000, 1000), (8_000, 2000), (2_000, 1000),
             (500, 1000), (3_000, 300)):
    q, t = synth.pairs_global_1k(n, seed=1, length=L) if n * L <= 2e7 else (None, None)
    if q is None:
        rng = np.random.default_rng(1)
        t = synth.random_bases(rng, (n, L))
        q = synth.substitute(rng, t.copy(), 0.05)
    for typ in (0, 1):
        line = "n=%d %dx%d type %d:" % (n, L, L, typ)
        for fam in (None, "lane", "wave"):
            capi.set_option("force_family", {None: 0, "lane": 1, "wave": 2}[fam])
            name, dt, cells = run(q, t, typ)
            line += "  %s %s %.2f ms %.0f GCUPS" % (fam or "auto", name, dt * 1e3, cells / dt / 1e9)
        capi.set_option("force_family", 0)
        print(line, flush=True)
