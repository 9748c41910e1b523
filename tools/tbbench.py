import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from sequencealigner_amd import capi
if len(sys.argv) > 1: capi.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from sequencealigner_amd import align, synth
lib = capi.lib()
q, t = synth.pairs_global_1k(10000, seed=1001)
for typ in (0, 2):
    b = align.DeviceAlignBatch(q, t, typ, 1, -1, -1)
    b.run(); torch.cuda.synchronize()
    lib.sa_align_profile(1)
    for _ in range(5): b.run()
    torch.cuda.synchronize()
    ms = (C.c_double * 3)(); cnt = (C.c_uint64 * 3)()
    lib.sa_align_profile_read(ms, cnt); lib.sa_align_profile(0)
    print(sys.argv[1:] , typ, b.kernel_name(), "fill %.3f tb %.3f" % (ms[1] / 5, ms[2] / 5), flush=True)
