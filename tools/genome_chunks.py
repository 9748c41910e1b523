"""Runs the genome-sized minimizer call with several chunk lengths (for a kernel trace)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequencealigner_amd import capi, minimize, synth
torch.cuda.set_device(0)
seqs = [synth.genome(4_600_000, seed=4)]
for c in [int(x) for x in sys.argv[1:]] or [32, 64, 128, 256]:
    capi.set_option("min_chunk", c)
    b = minimize.DeviceMinimizeBatch(seqs, 15, 5)
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.time()
        b.run(); torch.cuda.synchronize(); dt = time.time() - t0
    print("chunk", c, "call %.3f ms" % (dt * 1e3), flush=True)
    del b
