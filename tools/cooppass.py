"""Hundreds of long pairs: the one-wavefront-per-pair kernels against the cooperative kernel run in passes
(options force_coop, coop_c, coop_waves)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequencealigner_amd import align, capi, synth

lib = capi.lib()


def run(q, t, typ, label):
    b = align.DeviceAlignBatch(q, t, typ, 1, -1, -1)
    b.run()
    torch.cuda.synchronize()
    lib.sa_align_profile(1)
    b.run()
    torch.cuda.synchronize()
    ms = (C.c_double * 3)()
    cnt = (C.c_uint64 * 3)()
    lib.sa_align_profile_read(ms, cnt)
    lib.sa_align_profile(0)
    print(f"  {label:36s} {b.kernel_name():14s} fill {ms[1]:8.2f} ms ({cnt[1]} launches)  traceback {ms[2]:7.2f} ms",
          flush=True)
    s = b.scores().copy()
    del b
    torch.cuda.empty_cache()
    return s


shapes = [(10_000, 11_000, (128, 256, 512, 1024, 2048)), (3_000, 3_300, (256, 1024, 4096)), (30_000, 33_000, (64, 256)),
          (800, 50_000, (16, 100, 400)), (1000, 1100, (16, 100, 400, 1000)), (500, 5000, (100, 400))]
for read_len, window_len, counts in shapes:
    for n in counts:
        q, t = synth.pairs_semiglobal_long(n, seed=3, read_len=read_len, window_len=window_len, lead=200)
        print(f"{n} pairs {read_len} x {window_len} semiglobal", flush=True)
        capi.set_option("force_family", 2)
        capi.set_option("force_coop", 0)
        ref = run(q, t, 2, "one wavefront per pair")
        capi.set_option("force_coop", 1)
        for c in (2, 4, 8):
            for waves in ((1024, 2048, 4096, 8192) if read_len >= 3000 else (8192,)):
                capi.set_option("coop_c", c)
                capi.set_option("coop_waves", waves)
                s = run(q, t, 2, f"cooperative C {c} waves/pass {waves}")
                assert (s == ref).all()
capi.set_option("force_family", 0)
capi.set_option("force_coop", -1)
capi.set_option("coop_c", 0)
capi.set_option("coop_waves", 8192)
