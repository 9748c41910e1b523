"""Times the traceback of single long pairs with one wavefront and with one wavefront per segment of target
rows (option seg_rows), and checks that both give the same runs.  Shapes: a related pair, and the same pair with
the query or the target cut to a fifth (the path then holds a gap of 80 000 columns or rows)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequencealigner_amd import align, capi, synth

lib = capi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
qq, tt = synth.pair_global_long(n, seed=5)
for Q, T in ((n, n), (n, n // 5), (n // 5, n)):
    q, t = qq[:Q], tt[:T]
    for typ in (0, 2):
        ref = None
        for rows, noseg, spec in ((0, 1, 2), (1024, 0, 0), (1024, 0, 1), (1024, 0, 2), (1024, 0, 3), (2048, 0, 2)):
            capi.set_option("no_segments", noseg)
            capi.set_option("seg_spec", spec)
            capi.set_option("seg_rows", rows)
            b = align.DeviceAlignBatch([q], [t], typ, 1, -1, -1)
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
            got = (b.cigars([0])[0], int(b.target_begins()[0]), int(b.scores()[0]))
            if noseg:
                ref = got
            print(f"Q {Q:6d} T {T:6d} type {typ} rows {rows or 'one wavefront':>13} spec {spec}: fill {ms[1] / 3:7.3f} ms  "
                  f"traceback {ms[2] / 3:7.3f} ms  same as one wavefront: {got == ref}", flush=True)
            del b
capi.set_option("no_segments", 0)
capi.set_option("seg_rows", 0)
capi.set_option("seg_spec", 2)
