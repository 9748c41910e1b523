"""Times the reads shape of the minimizer benchmark with several chunk lengths (option "min_chunk")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequencealigner_amd import capi, minimize, synth
seqs = synth.reads(100_000, 10_000, seed=6)
for chunk in (0, 512, 1024, 1536, 2048, 2560, 3328, 4096):
    capi.set_option("min_chunk", chunk)
    b = minimize.DeviceMinimizeBatch(seqs, 15, 5)
    b.run(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): b.run()
    torch.cuda.synchronize()
    print("chunk", chunk, "%.3f ms" % ((time.time() - t0) / 3 * 1e3), flush=True)
    del b
capi.set_option("min_chunk", 0)
