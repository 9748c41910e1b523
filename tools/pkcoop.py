"""Long pairs, Semiglobal or Global: the packed wavefront kernel with one wavefront per two pairs against its
cooperative form (options coop_pk, coop_c, coop_waves) and the 32-bit cooperative kernel.
usage: python tools/pkcoop.py [shape ...]   shape = pairs:read_len:window_len[:type[:workspace GiB]]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequencealigner_amd import align, capi, synth

lib = capi.lib()


def run(q, t, typ, label, ws):
    b = align.DeviceAlignBatch(q, t, typ, 1, -1, -1, **({"workspace_bytes": ws} if ws else {}))
    b.run()
    torch.cuda.synchronize()
    lib.sa_align_profile(1)
    t0 = time.time()
    b.run()
    torch.cuda.synchronize()
    wall = (time.time() - t0) * 1e3
    ms = (C.c_double * 3)()
    cnt = (C.c_uint64 * 3)()
    lib.sa_align_profile_read(ms, cnt)
    lib.sa_align_profile(0)
    cells = float(sum(len(a) * len(c) for a, c in zip(q, t)))
    print(f"  {label:34s} {b.kernel_name():15s} fill {ms[1]:8.2f} ms ({cnt[1]} launches) traceback {ms[2]:7.2f} ms "
          f"call {wall:8.2f} ms  {cells / wall / 1e9:6.2f} TCUPS end to end, {cells / ms[1] / 1e9:6.2f} fill", flush=True)
    s = b.scores().copy()
    del b
    torch.cuda.empty_cache()
    return s


shapes = sys.argv[1:] or ["256:10000:11000", "1024:10000:11000"]
for shape in shapes:
    f = shape.split(":")
    n, read_len, window_len = int(f[0]), int(f[1]), int(f[2])
    typ = int(f[3]) if len(f) > 3 else 2
    ws = int(float(f[4]) * (1 << 30)) if len(f) > 4 else 0
    if typ == 2:
        q, t = synth.pairs_semiglobal_long(n, seed=3, read_len=read_len, window_len=window_len, lead=200)
    else:
        q, t = synth.pairs_global_1k(n, seed=3, length=read_len)
        t = t[:, :window_len]
    print(f"{n} pairs {read_len} x {window_len} type {typ}" + (f" workspace {ws >> 30} GiB" if ws else ""), flush=True)
    capi.set_option("force_family", 2)
    modes = os.environ.get("PKCOOP_MODES", "auto,wave,pk8,pk10,pk12,pk16,i32").split(",")
    ref = None
    for mode in modes:
        capi.set_option("force_coop", -1)
        capi.set_option("coop_pk", -1)
        capi.set_option("coop_c", 0)
        if mode == "wave":
            capi.set_option("force_coop", 0)
        elif mode.startswith("pk"):
            capi.set_option("force_coop", 1)
            capi.set_option("coop_pk", 1)
            capi.set_option("coop_c", int(mode[2:]))
        elif mode == "i32":
            capi.set_option("force_coop", 1)
            capi.set_option("coop_pk", 0)
        s = run(q, t, typ, mode, ws)
        if ref is None:
            ref = s
        assert (s == ref).all(), mode
capi.set_option("force_family", 0)
capi.set_option("force_coop", -1)
capi.set_option("coop_pk", -1)
capi.set_option("coop_c", 0)
