"""Times the 100 kbp x 100 kbp pair's traceback with one wavefront and with one wavefront per segment of target
rows (option seg_rows), and checks that both give the same runs."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequencealigner_amd import align, capi, synth

lib = capi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
qq, tt = synth.pair_global_long(n, seed=5)
ref = None
for typ in (0, 2):
    for rows, noseg in ((0, 1), (512, 0), (1024, 0), (2048, 0), (4096, 0)):
        capi.set_option("no_segments", noseg)
        capi.set_option("seg_rows", rows)
        b = align.DeviceAlignBatch([qq], [tt], typ, 1, -1, -1)
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
        cg = b.cigars([0])[0]
        tb = int(b.target_begins()[0])
        if noseg:
            ref = (cg, tb, int(b.scores()[0]))
        same = (cg, tb, int(b.scores()[0])) == ref
        print(f"type {typ} rows {rows or 'one wavefront':>13}: fill {ms[1] / 3:7.3f} ms  traceback {ms[2] / 3:7.3f} ms  "
              f"runs {cg.count('M') + cg.count('X') + cg.count('I') + cg.count('D')}  same as one wavefront: {same}",
              flush=True)
capi.set_option("no_segments", 0)
capi.set_option("seg_rows", 0)
