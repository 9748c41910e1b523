"""Times the cooperative Global fill on single pairs of several shapes and strip widths, to separate the two
terms of its run time: (target rows + strips * lag) * time per step.  Prints fill / traceback milliseconds."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequencealigner_amd import align, capi, synth

if len(sys.argv) > 1:   # another build of the library (timing experiments)
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
lib = capi.lib()
shapes = ((100_000, 100_000), (100_000, 20_000), (20_000, 100_000), (20_000, 20_000))
for (Q, T) in shapes[:2] if len(sys.argv) > 1 else shapes:
    q, t = synth.pair_global_long(max(Q, T), seed=5)
    q, t = q[:Q], t[:T]
    for c, no_tables in ((2, 0), (2, 1), (4, 0), (4, 1), (8, 0)) if len(sys.argv) <= 1 else ((2, 0),):
        capi.set_option("coop_c", c)
        capi.set_option("no_tables", no_tables)
        b = align.DeviceAlignBatch([q], [t], 0, 1, -1, -1)
        b.run()
        torch.cuda.synchronize()
        lib.sa_align_profile(1)
        for _ in range(3):
            b.run()
        torch.cuda.synchronize()
        ms = (C.c_double * 3)()
        cnt = (C.c_uint64 * 3)()
        lib.sa_align_profile_read(ms, cnt)
        lib.sa_align_profile(0)
        print("Q %6d T %6d C %d tables %d %s: fill %.2f ms, traceback %.2f ms" % (Q, T, c, 1 - no_tables, b.kernel_name(), ms[1] / 3, ms[2] / 3),
              flush=True)
        del b
capi.set_option("coop_c", 0)
capi.set_option("no_tables", 0)
