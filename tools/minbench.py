"""Times the minimizer path on the benchmark shapes (genome + reads); used under rocprofv3."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sequencealigner_amd import capi, minimize, synth
if os.environ.get("SA_LIB"):   # a library built with other options (kernel experiments)
    capi.LIB_PATH = os.path.abspath(os.environ["SA_LIB"])
torch.cuda.set_device(0)
if os.environ.get("SA_MIN_CHUNK"):   # windows per chunk (test hook of the library)
    capi.set_option("min_chunk", int(os.environ["SA_MIN_CHUNK"]))
which = sys.argv[1] if len(sys.argv) > 1 else "both"
shapes = []
if which in ("both", "genome"):
    shapes.append(("genome4.6M", [synth.genome(4_600_000, seed=4)]))
if which in ("both", "reads"):
    shapes.append(("reads100kx10k", synth.reads(100_000, 10_000, seed=6)))
for name, seqs in shapes:
    b = minimize.DeviceMinimizeBatch(seqs, 15, 5)
    for it in range(int(os.environ.get('SA_ITERS', '3'))):
        torch.cuda.synchronize(); t0 = time.time()
        b.run(); torch.cuda.synchronize(); dt = time.time() - t0
        print(name, "run %.3f ms" % (dt * 1e3), "%.2f Gwindows/s" % (b.windows / dt / 1e9), "out", int(b.d_out_off[-1].item()), flush=True)
    del b
    torch.cuda.empty_cache()
